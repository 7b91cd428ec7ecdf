// chs_fast_core.h -- device core of the fast transform engine (gfx950).
//
// One length-N orthonormal DCT-II (scipy.fftpack.dct(type=2, norm='ortho'), the 1-D
// factor of the dctn calls at chsimpy/solver.py:159,201) is computed by a GROUP of G
// lanes (G <= 64, so a group never spans wavefronts) as
//     Makhoul reordering + packing  z[n] = v[2n] + i v[2n+1]        (implicit in the loads)
//  -> M = N/2 point complex FFT, decimation in frequency, 2 or 3 register-resident
//     radix passes with the operands exchanged through a group-private LDS scratch
//  -> real-FFT recombination of (Z[k], Z[M-k]) and the quarter-wave twiddle.
// Each lane keeps E = M/G complex values in registers.  Two ownership tricks remove
// every exchange except the ones between radix passes:
//   * pass 0 owns MIRROR PAIRS of butterflies (m', L1-1-m'): the 32-byte quads
//     x[4q..4q+3] a lane loads contain exactly the operands of its two butterflies;
//   * the last pass owns MIRROR PAIRS (kappa, S2-kappa): Z[k] and Z[M-k] meet in one lane.
// The DCT-III (inverse, solver.py:208) is the exact transpose of this network run
// backwards (the orthonormal DCT matrix is orthogonal).
// tools/dct_model.py is the executable model of these index maps.
#pragma once
#include <hip/hip_runtime.h>

// ---------------------------------------------------------------------------
// compile-time configuration
// ---------------------------------------------------------------------------
template <typename T_, int N_, int G_, int R0_, int R1_, int R2_, int PAD1_, int PAD2_>
struct FCfg {
  using T = T_;
  static constexpr int N = N_;
  static constexpr int M = N_ / 2;
  static constexpr int G = G_;          // lanes per transform
  static constexpr int E = M / G_;      // complex values per lane
  static constexpr int R0 = R0_, R1 = R1_, R2 = R2_;  // R1 == 1: two passes only
  static constexpr int L1 = M / R0_;    // sub-transform length after pass 0
  static constexpr int L2 = L1 / R1_;   // == R2
  static constexpr int S1 = R0_;
  static constexpr int S2 = R0_ * R1_;  // number of last-pass butterflies
  static constexpr int NP0 = E / (2 * R0_);  // mirror pairs per lane, pass 0
  static constexpr int NB1 = E / R1_;        // butterflies per lane, pass 1
  static constexpr int NP2 = E / (2 * R2_);  // mirror pairs per lane, last pass
  static constexpr int THREADS = 256;
  static constexpr int C = THREADS / G_;     // transforms per workgroup
  static constexpr int P1 = L1 + PAD1_;      // pitch of A1[k0][m']
  static constexpr int P2 = S2 + PAD2_;      // pitch of A2[m''][kappa]
  static constexpr int SCR1 = (R1_ > 1) ? R0_ * P1 : 0;
  static constexpr int SCR2 = R2_ * P2;
  static constexpr int SCR = (SCR1 > SCR2 ? SCR1 : SCR2);  // scratch elements per transform
  static_assert(L2 == R2_, "radices must multiply to M");
  static_assert(NP0 >= 1 && NP2 >= 1, "E must be >= 2*R0 and >= 2*R2");
  static_assert(R1_ == 1 || NB1 >= 1, "E must be >= R1");
  static_assert(G_ <= 64 && (64 % G_) == 0, "a group must not span wavefronts");
};

// twiddle tables in global memory (complex interleaved: re, im)
template <typename T>
struct FTables {
  const T* tw0;  // [(k0-1)*L1 + m'] = omega_M^(m' k0),   k0 = 1..R0-1
  const T* tw1;  // [(k1-1)*L2 + m''] = omega_L1^(m'' k1), k1 = 1..R1-1
  const T* wp;   // [kk] = -i exp(-2 pi i kk / N),          kk = 0..M
  const T* t1;   // [kk] = s_kk/2 exp(-i pi kk/(2N))
  const T* t2;   // [kk] = conj(s_(M-kk)/2 exp(-i pi (M-kk)/(2N)))
};

// ---------------------------------------------------------------------------
// small complex helpers and the constant roots of unity (16th roots)
// ---------------------------------------------------------------------------
#define FC_SQRT1_2 0.70710678118654752440
#define FC_COS_PI_8 0.92387953251128675613
#define FC_SIN_PI_8 0.38268343236508977173

// (r, i) *= exp(-2 pi i * IDX / 16)   (IDX taken mod 16); CONJ flips the sign of the angle
template <typename T, int IDX, bool CONJ>
__device__ __forceinline__ void mul_w16(T& r, T& i) {
  constexpr int idx = ((CONJ ? -IDX : IDX) % 16 + 16) % 16;
  if constexpr (idx == 0) {
  } else if constexpr (idx == 4) {   // -i
    T t = r; r = i; i = -t;
  } else if constexpr (idx == 8) {
    r = -r; i = -i;
  } else if constexpr (idx == 12) {  // +i
    T t = r; r = -i; i = t;
  } else {
    // exp(-i a) = c - i s with a = 2 pi idx/16
    constexpr double cs[16] = {1.0, FC_COS_PI_8, FC_SQRT1_2, FC_SIN_PI_8, 0.0, -FC_SIN_PI_8, -FC_SQRT1_2, -FC_COS_PI_8,
                               -1.0, -FC_COS_PI_8, -FC_SQRT1_2, -FC_SIN_PI_8, 0.0, FC_SIN_PI_8, FC_SQRT1_2, FC_COS_PI_8};
    constexpr double sn[16] = {0.0, FC_SIN_PI_8, FC_SQRT1_2, FC_COS_PI_8, 1.0, FC_COS_PI_8, FC_SQRT1_2, FC_SIN_PI_8,
                               0.0, -FC_SIN_PI_8, -FC_SQRT1_2, -FC_COS_PI_8, -1.0, -FC_COS_PI_8, -FC_SQRT1_2, -FC_SIN_PI_8};
    const T c = (T)cs[idx], s = (T)sn[idx];
    const T nr = r * c + i * s;
    const T ni = i * c - r * s;
    r = nr; i = ni;
  }
}

// In-place DFT of size R on re[0..R), im[0..R): y[k] = sum_j a[j] exp(-+2 pi i jk/R)
// (INV: conjugate roots, unnormalised).  Natural order in and out.
template <typename T, int R, bool INV>
struct Dft;

template <typename T, bool INV>
struct Dft<T, 1, INV> {
  static __device__ __forceinline__ void run(T*, T*) {}
};

template <typename T, bool INV>
struct Dft<T, 2, INV> {
  static __device__ __forceinline__ void run(T* re, T* im) {
    const T r = re[0] - re[1], i = im[0] - im[1];
    re[0] += re[1]; im[0] += im[1];
    re[1] = r; im[1] = i;
  }
};

template <typename T, bool INV>
struct Dft<T, 4, INV> {
  static __device__ __forceinline__ void run(T* re, T* im) {
    const T t0r = re[0] + re[2], t0i = im[0] + im[2];
    const T t1r = re[0] - re[2], t1i = im[0] - im[2];
    const T t2r = re[1] + re[3], t2i = im[1] + im[3];
    T t3r = re[1] - re[3], t3i = im[1] - im[3];
    mul_w16<T, 4, INV>(t3r, t3i);  // * (-i) forward, * (+i) inverse
    re[0] = t0r + t2r; im[0] = t0i + t2i;
    re[2] = t0r - t2r; im[2] = t0i - t2i;
    re[1] = t1r + t3r; im[1] = t1i + t3i;
    re[3] = t1r - t3r; im[3] = t1i - t3i;
  }
};

// R = A*B Cooley-Tukey with A = 4:  n = B n1 + n2,  k = k1 + A k2
template <typename T, int R, bool INV>
struct Dft {
  static constexpr int A = 4, B = R / 4;
  static_assert(R == 8 || R == 16, "radix must be 2, 4, 8 or 16");
  template <int N2, int K1>
  static __device__ __forceinline__ void tw(T& r, T& i) {
    mul_w16<T, (N2 * K1 * 16) / R, INV>(r, i);
  }
  template <int N2>
  static __device__ __forceinline__ void col(const T* re, const T* im, T (*cr)[A], T (*ci)[A]) {
    T xr[A], xi[A];
#pragma unroll
    for (int n1 = 0; n1 < A; ++n1) { xr[n1] = re[B * n1 + N2]; xi[n1] = im[B * n1 + N2]; }
    Dft<T, A, INV>::run(xr, xi);
    tw<N2, 1>(xr[1], xi[1]);
    tw<N2, 2>(xr[2], xi[2]);
    tw<N2, 3>(xr[3], xi[3]);
#pragma unroll
    for (int k1 = 0; k1 < A; ++k1) { cr[N2][k1] = xr[k1]; ci[N2][k1] = xi[k1]; }
  }
  static __device__ __forceinline__ void run(T* re, T* im) {
    T cr[B][A], ci[B][A];
    col<0>(re, im, cr, ci);
    col<1>(re, im, cr, ci);
    if constexpr (B == 4) {
      col<2>(re, im, cr, ci);
      col<3>(re, im, cr, ci);
    }
#pragma unroll
    for (int k1 = 0; k1 < A; ++k1) {
      T yr[B], yi[B];
#pragma unroll
      for (int n2 = 0; n2 < B; ++n2) { yr[n2] = cr[n2][k1]; yi[n2] = ci[n2][k1]; }
      Dft<T, B, INV>::run(yr, yi);
#pragma unroll
      for (int k2 = 0; k2 < B; ++k2) { re[k1 + A * k2] = yr[k2]; im[k1 + A * k2] = yi[k2]; }
    }
  }
};

// (r, i) *= (wr, wi)  or  *= conj(wr, wi)
template <typename T, bool CONJ>
__device__ __forceinline__ void cmul(T& r, T& i, T wr, T wi) {
  if constexpr (!CONJ) {
    const T nr = r * wr - i * wi;
    const T ni = r * wi + i * wr;
    r = nr; i = ni;
  } else {
    const T nr = r * wr + i * wi;
    const T ni = i * wr - r * wi;
    r = nr; i = ni;
  }
}

template <typename T>
__device__ __forceinline__ void ldc(const T* __restrict__ tab, int idx, T& r, T& i) {
  if constexpr (sizeof(T) == 8) {
    const double2 v = *reinterpret_cast<const double2*>(tab + 2 * (size_t)idx);
    r = v.x; i = v.y;
  } else {
    const float2 v = *reinterpret_cast<const float2*>(tab + 2 * (size_t)idx);
    r = v.x; i = v.y;
  }
}

// Group-level synchronisation of the LDS exchange.  A group never spans wavefronts and
// the LDS executes one wave's accesses in issue order, so no s_barrier is needed: the
// wavefront-scope fences only keep the COMPILER from moving LDS accesses across the point
// where other lanes of the same wave take over the data.
__device__ __forceinline__ void xsync() {
#ifdef CHS_XSYNC_BLOCK
  __syncthreads();
#else
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

// ---------------------------------------------------------------------------
// ownership helpers (runtime, per lane)
// ---------------------------------------------------------------------------
template <class C>
struct Own {
  // last-pass pair q of lane l: kappa1, kappa2 and whether it is the special pair (0, S2/2)
  static __device__ __forceinline__ void last_pair(int l, int q, int& k1, int& k2, bool& special) {
    const int kap = l + C::G * q;
    special = (kap == 0);
    k1 = special ? 0 : kap;
    k2 = special ? C::S2 / 2 : C::S2 - kap;
  }
  // frequency kk of recombination slot k of pair q
  static __device__ __forceinline__ int slot_kk(int l, int q, int k) {
    const int kap = l + C::G * q;
    if (kap == 0) return (k < C::R2 / 2) ? C::S2 * k : C::S2 / 2 + C::S2 * (k - C::R2 / 2);
    return kap + C::S2 * k;
  }
  // output index of position (q, k, t), t = 0..3  <->  (kk, N-kk, M-kk, M+kk);
  // the special lane's position k=0 holds (X[0], X[M/2], X[M], X[3M/2]).
  static __device__ __forceinline__ int out_index(int l, int q, int k, int t) {
    const int kap = l + C::G * q;
    if (kap == 0 && k == 0) return t * (C::M / 2);
    const int kk = slot_kk(l, q, k);
    return t == 0 ? kk : (t == 1 ? C::N - kk : (t == 2 ? C::M - kk : C::M + kk));
  }
};

// ---------------------------------------------------------------------------
// recombination slot and its adjoint (tools/dct_model.py: slot_fwd / slot_adj)
// ---------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void slot_fwd(T Ar, T Ai, T Zr, T Zi, const FTables<T>& tb, int kk, T& y0, T& y1, T& y2,
                                         T& y3) {
  // B = conj(Z2); P = A + B; D = A - B
  const T Pr = Ar + Zr, Pi = Ai - Zi;
  T Dr = Ar - Zr, Di = Ai + Zi;
  T wr, wi, ar, ai, br, bi;
  ldc(tb.wp, kk, wr, wi);
  ldc(tb.t1, kk, ar, ai);
  ldc(tb.t2, kk, br, bi);
  cmul<T, false>(Dr, Di, wr, wi);  // Q = w' D
  const T S1r = Pr + Dr, S1i = Pi + Di;
  const T S2r = Pr - Dr, S2i = Pi - Di;
  y0 = ar * S1r - ai * S1i;      // Re(T1 S1)
  y1 = -(ar * S1i + ai * S1r);   // -Im(T1 S1)
  y2 = br * S2r - bi * S2i;      // Re(T2 S2)
  y3 = br * S2i + bi * S2r;      // Im(T2 S2)
}

template <typename T>
__device__ __forceinline__ void slot_adj(T y0, T y1, T y2, T y3, const FTables<T>& tb, int kk, T& gAr, T& gAi, T& gZr,
                                         T& gZi) {
  T wr, wi, ar, ai, br, bi;
  ldc(tb.wp, kk, wr, wi);
  ldc(tb.t1, kk, ar, ai);
  ldc(tb.t2, kk, br, bi);
  // gS1 = conj(T1 * (y0 + i y1));  gS2 = conj(T2) * (y2 + i y3)
  const T g1r = ar * y0 - ai * y1, g1i = -(ar * y1 + ai * y0);
  const T g2r = br * y2 + bi * y3, g2i = br * y3 - bi * y2;
  const T gPr = g1r + g2r, gPi = g1i + g2i;
  T gDr = g1r - g2r, gDi = g1i - g2i;
  cmul<T, true>(gDr, gDi, wr, wi);  // gD = conj(w') gQ
  gAr = gPr + gDr; gAi = gPi + gDi;
  // gB = gP - gD ; gZ2 = conj(gB)
  gZr = gPr - gDr; gZi = -(gPi - gDi);
}

// ---------------------------------------------------------------------------
// LDS exchange helpers.  `scr` points at the group's scratch (C::SCR elements);
// real and imaginary parts travel one after the other through the same scratch.
// ---------------------------------------------------------------------------

// ===========================================================================
// Forward radix passes.  In: re/im[E] = pass-0 operands, index ((q*2+b)*R0 + j).
// Out: re/im[E] = last-pass outputs Z, index ((q*2+b)*R2 + k)  (b = 0: kappa1, 1: kappa2).
// ===========================================================================
template <class C>
__device__ __forceinline__ void fwd_passes(typename C::T* re, typename C::T* im, typename C::T* scr,
                                           const FTables<typename C::T>& tb, int l) {
  using T = typename C::T;
  constexpr int R0 = C::R0, R1 = C::R1, R2 = C::R2;
  // ---- pass 0: radix R0 on every owned butterfly, then twiddle by omega_M^(m k0)
#pragma unroll
  for (int q = 0; q < C::NP0; ++q) {
    const int m1 = l + C::G * q;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int m = b ? (C::L1 - 1 - m1) : m1;
      T* r = re + (q * 2 + b) * R0;
      T* i = im + (q * 2 + b) * R0;
      Dft<T, R0, false>::run(r, i);
#pragma unroll
      for (int k = 1; k < R0; ++k) {
        T wr, wi;
        ldc(tb.tw0, (k - 1) * C::L1 + m, wr, wi);
        cmul<T, false>(r[k], i[k], wr, wi);
      }
    }
  }
  T xr[C::E], xi[C::E];
  if constexpr (R1 > 1) {
    // ---- exchange A1[k0][m'] -> pass-1 operands (kappa = id % S1, m'' = id / S1)
#pragma unroll
    for (int part = 0; part < 2; ++part) {
      T* src = part ? im : re;
      T* dst = part ? xi : xr;
      xsync();
#pragma unroll
      for (int q = 0; q < C::NP0; ++q) {
        const int m1 = l + C::G * q;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int m = b ? (C::L1 - 1 - m1) : m1;
#pragma unroll
          for (int k = 0; k < R0; ++k) scr[k * C::P1 + m] = src[(q * 2 + b) * R0 + k];
        }
      }
      xsync();
#pragma unroll
      for (int ib = 0; ib < C::NB1; ++ib) {
        const int id = l + C::G * ib;
        const int kap = id % C::S1, mm = id / C::S1;
#pragma unroll
        for (int j = 0; j < R1; ++j) dst[ib * R1 + j] = scr[kap * C::P1 + mm + C::L2 * j];
      }
    }
    // ---- pass 1
#pragma unroll
    for (int ib = 0; ib < C::NB1; ++ib) {
      const int id = l + C::G * ib;
      const int mm = id / C::S1;
      T* r = xr + ib * R1;
      T* i = xi + ib * R1;
      Dft<T, R1, false>::run(r, i);
#pragma unroll
      for (int k = 1; k < R1; ++k) {
        T wr, wi;
        ldc(tb.tw1, (k - 1) * C::L2 + mm, wr, wi);
        cmul<T, false>(r[k], i[k], wr, wi);
      }
    }
  }
  // ---- exchange into A2[m''][kappa] -> last-pass operands
#pragma unroll
  for (int part = 0; part < 2; ++part) {
    xsync();
    if constexpr (R1 > 1) {
      T* src = part ? xi : xr;
#pragma unroll
      for (int ib = 0; ib < C::NB1; ++ib) {
        const int id = l + C::G * ib;
        const int kap = id % C::S1, mm = id / C::S1;
#pragma unroll
        for (int k = 0; k < R1; ++k) scr[mm * C::P2 + kap + C::S1 * k] = src[ib * R1 + k];
      }
    } else {
      T* src = part ? im : re;
#pragma unroll
      for (int q = 0; q < C::NP0; ++q) {
        const int m1 = l + C::G * q;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int m = b ? (C::L1 - 1 - m1) : m1;
#pragma unroll
          for (int k = 0; k < R0; ++k) scr[m * C::P2 + k] = src[(q * 2 + b) * R0 + k];
        }
      }
    }
    xsync();
    T* dst = part ? im : re;
#pragma unroll
    for (int q = 0; q < C::NP2; ++q) {
      int k1, k2; bool sp;
      Own<C>::last_pair(l, q, k1, k2, sp);
#pragma unroll
      for (int mm = 0; mm < R2; ++mm) {
        dst[(q * 2 + 0) * R2 + mm] = scr[mm * C::P2 + k1];
        dst[(q * 2 + 1) * R2 + mm] = scr[mm * C::P2 + k2];
      }
    }
  }
  // ---- last pass
#pragma unroll
  for (int q = 0; q < 2 * C::NP2; ++q) Dft<T, R2, false>::run(re + q * R2, im + q * R2);
}

// ===========================================================================
// Recombination stage, in place on the last-pass registers.
//   FWD: (A, Z) -> y[4] = the four real coefficients of the slot (slot_fwd)
//   f(pbase, idx, y): the caller consumes / replaces y.  pbase = (q*R2 + k)*4 is the
//        compile-time position of the slot, idx[t] the coefficient index of y[t]
//        (Own::out_index).
//   ADJ: y[4] -> (gA, gZ) written back over (A, Z) (slot_adj)
// With FWD only the registers are left untouched; with ADJ only y comes from f.
// ===========================================================================
template <class C, bool FWD, bool ADJ, class F>
__device__ __forceinline__ void recombine(typename C::T* re, typename C::T* im, const FTables<typename C::T>& tb,
                                          int l, F&& f) {
  using T = typename C::T;
  constexpr int R2 = C::R2, N = C::N, M = C::M;
#pragma unroll
  for (int q = 0; q < C::NP2; ++q) {
    int k1, k2; bool sp;
    Own<C>::last_pair(l, q, k1, k2, sp);
    T* r1 = re + (q * 2 + 0) * R2; T* i1 = im + (q * 2 + 0) * R2;
    T* r2 = re + (q * 2 + 1) * R2; T* i2 = im + (q * 2 + 1) * R2;
    if (!sp) {
#pragma unroll
      for (int k = 0; k < R2; ++k) {
        const int kk = k1 + C::S2 * k;
        T y[4] = {T(0), T(0), T(0), T(0)};
        if constexpr (FWD) slot_fwd<T>(r1[k], i1[k], r2[R2 - 1 - k], i2[R2 - 1 - k], tb, kk, y[0], y[1], y[2], y[3]);
        const int idx[4] = {kk, N - kk, M - kk, M + kk};
        f((q * R2 + k) * 4, idx, y);
        if constexpr (ADJ) slot_adj<T>(y[0], y[1], y[2], y[3], tb, kk, r1[k], i1[k], r2[R2 - 1 - k], i2[R2 - 1 - k]);
      }
    } else {
      // butterfly 0 pairs with itself (k <-> R2-k), butterfly S2/2 with itself (k <-> R2-1-k);
      // position 0 holds (X[0], X[M/2], X[M], X[3M/2]) from the two self-paired outputs
      {
        T y[4] = {T(0), T(0), T(0), T(0)};
        if constexpr (FWD) {
          T a0, a1, a2, a3, b0, b1, b2, b3;
          slot_fwd<T>(r1[0], i1[0], r1[0], i1[0], tb, 0, a0, a1, a2, a3);
          slot_fwd<T>(r1[R2 / 2], i1[R2 / 2], r1[R2 / 2], i1[R2 / 2], tb, M / 2, b0, b1, b2, b3);
          y[0] = a0; y[1] = b0; y[2] = a2; y[3] = b1;
        }
        const int idx[4] = {0, M / 2, M, 3 * (M / 2)};
        f(q * R2 * 4, idx, y);
        if constexpr (ADJ) {
          T gar, gai, gzr, gzi;
          slot_adj<T>(y[0], T(0), y[2], T(0), tb, 0, gar, gai, gzr, gzi);
          r1[0] = gar + gzr; i1[0] = gai + gzi;
          slot_adj<T>(y[1], y[3], T(0), T(0), tb, M / 2, gar, gai, gzr, gzi);
          r1[R2 / 2] = gar + gzr; i1[R2 / 2] = gai + gzi;
        }
      }
#pragma unroll
      for (int k = 1; k < R2 / 2; ++k) {
        const int kk = C::S2 * k;
        T y[4] = {T(0), T(0), T(0), T(0)};
        if constexpr (FWD) slot_fwd<T>(r1[k], i1[k], r1[R2 - k], i1[R2 - k], tb, kk, y[0], y[1], y[2], y[3]);
        const int idx[4] = {kk, N - kk, M - kk, M + kk};
        f((q * R2 + k) * 4, idx, y);
        if constexpr (ADJ) slot_adj<T>(y[0], y[1], y[2], y[3], tb, kk, r1[k], i1[k], r1[R2 - k], i1[R2 - k]);
      }
#pragma unroll
      for (int k = 0; k < R2 / 2; ++k) {
        const int kk = C::S2 / 2 + C::S2 * k;
        T y[4] = {T(0), T(0), T(0), T(0)};
        if constexpr (FWD) slot_fwd<T>(r2[k], i2[k], r2[R2 - 1 - k], i2[R2 - 1 - k], tb, kk, y[0], y[1], y[2], y[3]);
        const int idx[4] = {kk, N - kk, M - kk, M + kk};
        f((q * R2 + R2 / 2 + k) * 4, idx, y);
        if constexpr (ADJ) slot_adj<T>(y[0], y[1], y[2], y[3], tb, kk, r2[k], i2[k], r2[R2 - 1 - k], i2[R2 - 1 - k]);
      }
    }
  }
}

// ===========================================================================
// Inverse radix passes (exact transpose of fwd_passes).  In: re/im[E] = last-pass
// output gradients, index ((q*2+b)*R2 + k).  Out: pass-0 operands ((q*2+b)*R0 + j).
// ===========================================================================
template <class C>
__device__ __forceinline__ void inv_passes(typename C::T* re, typename C::T* im, typename C::T* scr,
                                           const FTables<typename C::T>& tb, int l) {
  using T = typename C::T;
  constexpr int R0 = C::R0, R1 = C::R1, R2 = C::R2;
#pragma unroll
  for (int q = 0; q < 2 * C::NP2; ++q) Dft<T, R2, true>::run(re + q * R2, im + q * R2);
  // ---- exchange A2[m''][kappa] back
  T xr[C::E], xi[C::E];
#pragma unroll
  for (int part = 0; part < 2; ++part) {
    const T* src = part ? im : re;
    xsync();
#pragma unroll
    for (int q = 0; q < C::NP2; ++q) {
      int k1, k2; bool sp;
      Own<C>::last_pair(l, q, k1, k2, sp);
#pragma unroll
      for (int mm = 0; mm < R2; ++mm) {
        scr[mm * C::P2 + k1] = src[(q * 2 + 0) * R2 + mm];
        scr[mm * C::P2 + k2] = src[(q * 2 + 1) * R2 + mm];
      }
    }
    xsync();
    if constexpr (R1 > 1) {
      T* dst = part ? xi : xr;
#pragma unroll
      for (int ib = 0; ib < C::NB1; ++ib) {
        const int id = l + C::G * ib;
        const int kap = id % C::S1, mm = id / C::S1;
#pragma unroll
        for (int k = 0; k < R1; ++k) dst[ib * R1 + k] = scr[mm * C::P2 + kap + C::S1 * k];
      }
    } else {
      T* dst = part ? xi : xr;
#pragma unroll
      for (int q = 0; q < C::NP0; ++q) {
        const int m1 = l + C::G * q;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int m = b ? (C::L1 - 1 - m1) : m1;
#pragma unroll
          for (int k = 0; k < R0; ++k) dst[(q * 2 + b) * R0 + k] = scr[m * C::P2 + k];
        }
      }
    }
  }
  if constexpr (R1 > 1) {
    // ---- pass 1 transposed: conj twiddle, conjugate DFT
#pragma unroll
    for (int ib = 0; ib < C::NB1; ++ib) {
      const int id = l + C::G * ib;
      const int mm = id / C::S1;
      T* r = xr + ib * R1;
      T* i = xi + ib * R1;
#pragma unroll
      for (int k = 1; k < R1; ++k) {
        T wr, wi;
        ldc(tb.tw1, (k - 1) * C::L2 + mm, wr, wi);
        cmul<T, true>(r[k], i[k], wr, wi);
      }
      Dft<T, R1, true>::run(r, i);
    }
    // ---- exchange A1[k0][m'] back
#pragma unroll
    for (int part = 0; part < 2; ++part) {
      const T* src = part ? xi : xr;
      T* dst = part ? im : re;
      xsync();
#pragma unroll
      for (int ib = 0; ib < C::NB1; ++ib) {
        const int id = l + C::G * ib;
        const int kap = id % C::S1, mm = id / C::S1;
#pragma unroll
        for (int j = 0; j < R1; ++j) scr[kap * C::P1 + mm + C::L2 * j] = src[ib * R1 + j];
      }
      xsync();
#pragma unroll
      for (int q = 0; q < C::NP0; ++q) {
        const int m1 = l + C::G * q;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int m = b ? (C::L1 - 1 - m1) : m1;
#pragma unroll
          for (int k = 0; k < R0; ++k) dst[(q * 2 + b) * R0 + k] = scr[k * C::P1 + m];
        }
      }
    }
  } else {
#pragma unroll
    for (int e = 0; e < C::E; ++e) { re[e] = xr[e]; im[e] = xi[e]; }
  }
  // ---- pass 0 transposed
#pragma unroll
  for (int q = 0; q < C::NP0; ++q) {
    const int m1 = l + C::G * q;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int m = b ? (C::L1 - 1 - m1) : m1;
      T* r = re + (q * 2 + b) * R0;
      T* i = im + (q * 2 + b) * R0;
#pragma unroll
      for (int k = 1; k < R0; ++k) {
        T wr, wi;
        ldc(tb.tw0, (k - 1) * C::L1 + m, wr, wi);
        cmul<T, true>(r[k], i[k], wr, wi);
      }
      Dft<T, R0, true>::run(r, i);
    }
  }
}

// ---------------------------------------------------------------------------
// quad <-> pass-0 operand packing (tools/dct_model.py: forward()/inverse()).
// Lane l, pair q, half-index j < R0/2:
//   quad Q1 = x[4*(m1 + L1 j) ..+3],  Q2 = x[4*(m2 + L1 j) ..+3],  m2 = L1-1-m1
//   a[j] = (Q1.0, Q1.2)   a[R0-1-j] = (Q2.3, Q2.1)      (butterfly m1)
//   b[j] = (Q2.0, Q2.2)   b[R0-1-j] = (Q1.3, Q1.1)      (butterfly m2)
// ---------------------------------------------------------------------------
template <class C>
__device__ __forceinline__ void pack_quads(const typename C::T q1[4], const typename C::T q2[4], int q, int j,
                                           typename C::T* re, typename C::T* im) {
  constexpr int R0 = C::R0;
  re[(q * 2 + 0) * R0 + j] = q1[0]; im[(q * 2 + 0) * R0 + j] = q1[2];
  re[(q * 2 + 1) * R0 + j] = q2[0]; im[(q * 2 + 1) * R0 + j] = q2[2];
  re[(q * 2 + 1) * R0 + (R0 - 1 - j)] = q1[3]; im[(q * 2 + 1) * R0 + (R0 - 1 - j)] = q1[1];
  re[(q * 2 + 0) * R0 + (R0 - 1 - j)] = q2[3]; im[(q * 2 + 0) * R0 + (R0 - 1 - j)] = q2[1];
}
template <class C>
__device__ __forceinline__ void unpack_quads(const typename C::T* re, const typename C::T* im, int q, int j,
                                             typename C::T q1[4], typename C::T q2[4]) {
  constexpr int R0 = C::R0;
  q1[0] = re[(q * 2 + 0) * R0 + j]; q1[2] = im[(q * 2 + 0) * R0 + j];
  q2[0] = re[(q * 2 + 1) * R0 + j]; q2[2] = im[(q * 2 + 1) * R0 + j];
  q1[3] = re[(q * 2 + 1) * R0 + (R0 - 1 - j)]; q1[1] = im[(q * 2 + 1) * R0 + (R0 - 1 - j)];
  q2[3] = re[(q * 2 + 0) * R0 + (R0 - 1 - j)]; q2[1] = im[(q * 2 + 0) * R0 + (R0 - 1 - j)];
}
