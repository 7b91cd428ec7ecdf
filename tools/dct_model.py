"""Executable model of the fast engine's 1-D DCT dataflow (design aid + test oracle
for the index maps; nothing here runs in the product).

A length-N orthonormal DCT-II is computed as
   Makhoul reordering -> half-length (M=N/2) complex FFT of z[n] = v[2n] + i v[2n+1]
   -> real-FFT recombination of (Z[k], Z[M-k]) -> quarter-wave twiddle,
with the work of one transform spread over G lanes (E = M/G complex per lane):
   * pass 0 takes its operands straight from 32-byte "quads" x[4q..4q+3]; each lane owns
     a mirror pair of butterflies (m', L1-1-m') so that no exchange is needed to pack z;
   * middle passes exchange through LDS;
   * the last pass owns mirror pairs (kappa, S-kappa) so the recombination needs no exchange.
The inverse (DCT-III) is the exact transpose of that network run backwards.
"""
import numpy as np


def plan(M, G, radices):
    P = len(radices)
    S = [1]
    for r in radices:
        S.append(S[-1] * r)
    assert S[-1] == M, (S, M)
    L = [M // s for s in S]
    E = M // G
    assert E * G == M
    assert 2 * radices[0] <= E and 2 * radices[-1] <= E
    for r in radices[1:-1]:
        assert r <= E
    return dict(M=M, G=G, E=E, P=P, r=list(radices), S=S, L=L)


def tables(N):
    M = N // 2
    k = np.arange(M + 1, dtype=np.longdouble)
    PI = np.pi.__class__(np.longdouble(3.14159265358979323846264338327950288419716939937510))
    PI = np.longdouble(3.14159265358979323846264338327950288419716939937510)
    # w'_k = -i * exp(-2 pi i k / N)
    ang = -2 * PI * k / N
    wp = (np.sin(ang) + 0j) + 1j * (-np.cos(ang))  # -i*(cos+ i sin) = sin - i cos
    # T_k = s_k/2 * exp(-i pi k/(2N)),  k = 0..N-1 (only 0..M needed for T1; T2 uses M-k)
    kk = np.arange(N, dtype=np.longdouble)
    s = np.full(N, np.sqrt(np.longdouble(2) / N))
    s[0] = np.sqrt(np.longdouble(1) / N)
    a2 = -PI * kk / (2 * N)
    T = (s / 2) * (np.cos(a2) + 1j * np.sin(a2))
    return wp.astype(np.complex128), T.astype(np.complex128)


def fft_twiddle(Lp, e):
    """omega_{Lp}^{e} accurately."""
    e = np.asarray(e) % Lp
    a = -2 * np.longdouble(np.pi) * e.astype(np.longdouble) / Lp
    PI = np.longdouble(3.14159265358979323846264338327950288419716939937510)
    a = -2 * PI * e.astype(np.longdouble) / Lp
    return (np.cos(a) + 1j * np.sin(a)).astype(np.complex128)


def dft_mat(r, conj=False):
    j = np.arange(r)
    W = fft_twiddle(r, np.outer(j, j))
    return W.conj() if conj else W


class Model:
    def __init__(self, N, G, radices):
        self.N = N
        self.M = N // 2
        self.pl = plan(self.M, G, radices)
        self.wp, self.T = tables(N)

    # ---- ownership maps -------------------------------------------------------
    def pass0_pairs(self, lane):
        """mirror pairs (m', m'') of pass-0 butterflies owned by `lane`"""
        pl = self.pl
        L1 = pl['L'][1]
        npairs = pl['E'] // (2 * pl['r'][0])
        out = []
        for q in range(npairs):
            m1 = lane + pl['G'] * q
            out.append((m1, L1 - 1 - m1))
        return out

    def last_pairs(self, lane):
        """pairs (kappa, kappa') of last-pass butterflies; (0, S/2) is the special pair"""
        pl = self.pl
        S = pl['S'][-2]
        npairs = pl['E'] // (2 * pl['r'][-1])
        out = []
        for q in range(npairs):
            k1 = lane + pl['G'] * q
            out.append((0, S // 2) if k1 == 0 else (k1, S - k1))
        return out

    def mid_butterflies(self, p, lane):
        """(kappa, m') of middle pass p owned by lane: butterfly id b = lane + G*i, m' fastest"""
        pl = self.pl
        Lp1 = pl['L'][p + 1]
        nb = pl['E'] // pl['r'][p]
        return [((lane + pl['G'] * i) // Lp1, (lane + pl['G'] * i) % Lp1) for i in range(nb)]

    # ---- forward --------------------------------------------------------------
    def forward(self, x):
        pl, N, M = self.pl, self.N, self.M
        r, S, L, G = pl['r'], pl['S'], pl['L'], pl['G']
        P = pl['P']
        x = np.asarray(x, dtype=np.float64)
        quad = x.reshape(M // 2, 4)
        # pass 0 straight from quads
        A = np.zeros((S[1], L[1]), dtype=np.complex128)  # A_1[k0][m']
        r0, L1 = r[0], L[1]
        W0 = dft_mat(r0)
        for lane in range(G):
            for (m1, m2) in self.pass0_pairs(lane):
                a = np.zeros(r0, complex)
                b = np.zeros(r0, complex)
                for j in range(r0 // 2):
                    q1 = quad[m1 + L1 * j]
                    q2 = quad[m2 + L1 * j]
                    a[j] = q1[0] + 1j * q1[2]
                    b[j] = q2[0] + 1j * q2[2]
                    b[r0 - 1 - j] = q1[3] + 1j * q1[1]
                    a[r0 - 1 - j] = q2[3] + 1j * q2[1]
                for (m, v) in ((m1, a), (m2, b)):
                    y = W0 @ v
                    if P > 1:
                        y = y * fft_twiddle(L[0], m * np.arange(r0))
                    A[:, m] = y
        # middle passes
        for p in range(1, P - 1):
            rp = r[p]
            Wp = dft_mat(rp)
            Lp, Lp1 = L[p], L[p + 1]
            B = np.zeros((S[p + 1], Lp1), dtype=np.complex128)
            for lane in range(G):
                for (kap, m) in self.mid_butterflies(p, lane):
                    v = A[kap, m + Lp1 * np.arange(rp)]
                    y = (Wp @ v) * fft_twiddle(Lp, m * np.arange(rp))
                    B[kap + S[p] * np.arange(rp), m] = y
            A = B
        # last pass + recombination; output dict k -> X[k]
        X = np.full(N + 1, np.nan)
        rl = r[-1]
        Sl = S[-2]
        Wl = dft_mat(rl)
        for lane in range(G):
            for (k1, k2) in self.last_pairs(lane):
                o1 = Wl @ A[k1, :] if P > 1 else None
                o2 = Wl @ A[k2, :]
                o1 = Wl @ A[k1, :]
                for (Aidx, Bidx, kk) in self.slots(k1, k2):
                    Av = (o1 if Aidx[0] == 0 else o2)[Aidx[1]]
                    Z2 = (o1 if Bidx[0] == 0 else o2)[Bidx[1]]
                    y = self.slot_fwd(Av, Z2, kk)
                    for idx, val in zip((kk, N - kk, M - kk, M + kk), y):
                        if idx < N:
                            assert np.isnan(X[idx]) or abs(X[idx] - val) < 1e-12 * (1 + abs(val)), (idx, X[idx], val)
                            X[idx] = val
        return X[:N]

    def slots(self, k1, k2):
        """(A source, Z[M-kk] source, kk) for every recombination slot of a butterfly pair.
        source = (which butterfly 0/1, output index)"""
        pl = self.pl
        rl, Sl = pl['r'][-1], pl['S'][-2]
        out = []
        if k1 != 0:
            for k in range(rl):
                out.append(((0, k), (1, rl - 1 - k), k1 + Sl * k))
        else:
            for k in range(rl // 2 + 1):          # butterfly 0 with itself (k=0 and rl/2 self-paired)
                out.append(((0, k), (0, (rl - k) % rl), Sl * k))
            for k in range(rl // 2):              # butterfly S/2 with itself
                out.append(((1, k), (1, rl - 1 - k), Sl // 2 + Sl * k))
        return out

    def slot_fwd(self, A, Z2, kk):
        M = self.M
        Bc = np.conj(Z2)
        Pp = A + Bc
        D = A - Bc
        Q = self.wp[kk] * D
        S1 = Pp + Q
        S2 = Pp - Q
        T1 = self.T[kk]
        T2 = np.conj(self.T[M - kk])
        u = T1 * S1
        v = T2 * S2
        return (u.real, -u.imag, v.real, v.imag)

    def slot_adj(self, y, kk):
        """adjoint of slot_fwd: returns (gA, gZ2)"""
        M = self.M
        T1 = self.T[kk]
        T2 = np.conj(self.T[M - kk])
        Y1 = y[0] + 1j * y[1]
        Y2 = y[2] + 1j * y[3]
        gS1 = np.conj(T1 * Y1)
        gS2 = np.conj(T2) * Y2
        gP = gS1 + gS2
        gQ = gS1 - gS2
        gD = np.conj(self.wp[kk]) * gQ
        gA = gP + gD
        gB = gP - gD
        return gA, np.conj(gB)

    # ---- inverse (exact transpose) ----------------------------------------------
    def inverse(self, X):
        pl, N, M = self.pl, self.N, self.M
        r, S, L, G = pl['r'], pl['S'], pl['L'], pl['G']
        P = pl['P']
        X = np.asarray(X, dtype=np.float64)
        rl, Sl = r[-1], S[-2]
        Wl = dft_mat(rl, conj=True)
        A = np.zeros((Sl, rl), dtype=np.complex128)
        for lane in range(G):
            for (k1, k2) in self.last_pairs(lane):
                g = [np.zeros(rl, complex), np.zeros(rl, complex)]
                for (Aidx, Bidx, kk) in self.slots(k1, k2):
                    y = [0.0, 0.0, 0.0, 0.0]
                    idxs = (kk, N - kk, M - kk, M + kk)
                    seen = set()
                    for t, idx in enumerate(idxs):
                        if idx < N and idx not in seen:   # duplicates of self-paired slots count once
                            y[t] = X[idx]
                            seen.add(idx)
                    gA, gZ2 = self.slot_adj(y, kk)
                    g[Aidx[0]][Aidx[1]] += gA
                    g[Bidx[0]][Bidx[1]] += gZ2
                A[k1, :] = Wl @ g[0]
                A[k2, :] = Wl @ g[1]
        for p in range(P - 2, 0, -1):
            rp = r[p]
            Wp = dft_mat(rp, conj=True)
            Lp, Lp1 = L[p], L[p + 1]
            B = np.zeros((S[p], Lp), dtype=np.complex128)
            for lane in range(G):
                for (kap, m) in self.mid_butterflies(p, lane):
                    gy = A[kap + S[p] * np.arange(rp), m] * np.conj(fft_twiddle(Lp, m * np.arange(rp)))
                    B[kap, m + Lp1 * np.arange(rp)] = Wp @ gy
            A = B
        r0, L1 = r[0], L[1]
        W0 = dft_mat(r0, conj=True)
        quad = np.zeros((M // 2, 4))
        for lane in range(G):
            for (m1, m2) in self.pass0_pairs(lane):
                ga = {}
                for m in (m1, m2):
                    gy = A[:, m]
                    if P > 1:
                        gy = gy * np.conj(fft_twiddle(L[0], m * np.arange(r0)))
                    ga[m] = W0 @ gy
                a, b = ga[m1], ga[m2]
                for j in range(r0 // 2):
                    quad[m1 + L1 * j, 0] = a[j].real
                    quad[m1 + L1 * j, 2] = a[j].imag
                    quad[m2 + L1 * j, 0] = b[j].real
                    quad[m2 + L1 * j, 2] = b[j].imag
                    quad[m1 + L1 * j, 3] = b[r0 - 1 - j].real
                    quad[m1 + L1 * j, 1] = b[r0 - 1 - j].imag
                    quad[m2 + L1 * j, 3] = a[r0 - 1 - j].real
                    quad[m2 + L1 * j, 1] = a[r0 - 1 - j].imag
        return quad.reshape(N)


if __name__ == '__main__':
    import scipy.fftpack as sf
    rng = np.random.default_rng(0)
    for (N, G, rad) in [(128, 4, (8, 8)), (256, 8, (8, 2, 8)), (512, 16, (8, 4, 8)), (1024, 32, (8, 8, 8)),
                        (2048, 64, (8, 16, 8)), (4096, 64, (16, 8, 16)), (8192, 64, (16, 16, 16)),
                        (128, 8, (4, 4, 4)), (256, 16, (4, 8, 4)), (512, 32, (4, 4, 4, 4)), (1024, 64, (4, 8, 4, 4)),
                        (8192, 256, (8, 8, 8, 8)), (4096, 128, (8, 4, 8, 8)), (4096, 128, (8, 8, 4, 8)), (2048, 128, (4, 8, 8, 4)), (8192, 128, (16, 4, 4, 16))]:
        m = Model(N, G, rad)
        x = rng.standard_normal(N)
        X = m.forward(x)
        ref = sf.dct(x, type=2, norm='ortho')
        e1 = np.max(np.abs(X - ref))
        xr = m.inverse(ref)
        e2 = np.max(np.abs(xr - x))
        print(N, G, rad, 'fwd err %.2e  inv err %.2e' % (e1, e2))
