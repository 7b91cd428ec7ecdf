"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY, never the product path.

A from-scratch numpy/scipy restatement of the reference's per-timestep
Cahn-Hilliard solver loop, written from the *text* of the reference sources
(the reference package is neither imported nor executed: SURVEY.md section 8c
records a binding permission denial for that).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; ``chsimpy_amd`` (the product) never does.

Pinning status ("parity unpinned" for the solver loop proper): the reference's
own tests hold exactly one numeric known-answer vector on this path -- the
5x4 LCG matrix of ``tests/test.py:25-37`` -- which this oracle reproduces
(``tests/test_oracle.py``).  The per-step solver results are not pinned by any
reference test; they are cross-checked against (a) the figures observed from
reference runs before the denial and recorded in SURVEY.md section 8(c)
(E[0], E2[0], E[-1], E2[-1], min/max U for N=128, seed 2023, ntmax=200) and
(b) analytic known answers (fixed point, mass conservation, single-mode
growth factor).  The transforms are ``scipy.fftpack.dctn/idctn`` -- the very
third-party routines the reference calls (``chsimpy/solver.py:159,201,208``).

Each function cites the reference file:line it follows.
"""

from __future__ import annotations

import numpy as np
import scipy.fftpack as scifft

# --------------------------------------------------------------------------
# a17  chsimpy/mport.py:8-32  -- "BSD rand" LCG carried out in float64
# --------------------------------------------------------------------------


def lcg_sample(n1: int, n2: int, seed) -> np.ndarray:
    """n1 x n2 sample on [0,1); follows chsimpy/mport.py:15-32 literally.

    The recurrence x <- (a*x + c) mod m is evaluated in IEEE float64 (a*x
    reaches ~2^61 > 2^53, so the product is a *rounded* double and ``%`` is the
    float64 fmod), values are stored column-major and divided by (m - 1).
    """
    a = np.float64(1103515245)
    c = np.float64(12345)
    m = np.float64(2 ** 31)
    x = seed
    sample = np.zeros((n1, n2))
    for i in range(n1 * n2):
        x = (a * x + c) % m
        sample[int(i % n1), int(i / n1)] = x
    sample /= (m - 1)
    return sample


# --------------------------------------------------------------------------
# chsimpy/utils.py:26-49  -- Redlich-Kister coefficients, DCT eigenvalue grid
# --------------------------------------------------------------------------


def A0(T):
    """chsimpy/utils.py:26-27"""
    return 186.0575 - 0.3654 * T


def A1(T):
    """chsimpy/utils.py:30-31"""
    return 43.7207 - 0.1401 * T


def eigenvalues(N: int) -> np.ndarray:
    """chsimpy/utils.py:34-36: leig_ij = lam_i + lam_j, lam_i = 2cos(pi i/(N-1)) - 2.

    (The N-1 in the cosine argument is the reference's, kept on purpose.)
    """
    lam = 2 * np.cos(np.pi * np.arange(0, N) / (N - 1)) - 2
    return lam.reshape(N, 1) @ np.ones((1, N)) + np.ones((N, 1)) @ lam.reshape(1, N)


def get_coefficients(N, kappa_tilde, delt, delx2):
    """chsimpy/utils.py:39-49"""
    lam1 = delt / delx2
    lam2 = kappa_tilde * lam1 / delx2
    leig = eigenvalues(N)
    CHeig = np.ones((N, N)) + lam2 * leig * leig
    Seig = lam1 * leig
    return CHeig, Seig


# --------------------------------------------------------------------------
# a16  chsimpy/parameters.py:24-61 -- the fields the path reads, with defaults
# --------------------------------------------------------------------------


class OracleParams:
    def __init__(self, **kw):
        self.seed = 2023
        self.N = 512
        self.L = 2
        self.XXX = 0.875
        self.temp = 650 + 273.15
        self.B = 12.86
        self.R = 0.0083144626181532
        self.N_A = 6.02214076e+23
        self.delt = 3e-8
        self.delt_max = 9e-8
        self.M_tilde = 1.71e-8
        self.kappa_tilde = 0.0002989112919661156  # reference default (sympy 1.14)
        self.threshold = self.XXX
        self.ntmax = int(1e6)
        self.full_sim = False
        self.time_max = None
        self.generator = 'uniform'
        self.adaptive_time = False
        self.jitter = None
        self.func_A0 = A0
        self.func_A1 = A1
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError(k)
            setattr(self, k, v)
        if 'threshold' not in kw:
            self.threshold = self.XXX


class OracleTimeData:
    """chsimpy/timedata.py:4-63; columns [it, E, E2, SA, domtime, Ra, L2, PS, delt]."""

    def __init__(self):
        self._rows = []

    def insert(self, it, delt, E, E2, SA, domtime, Ra, L2, PS):
        row = [it, E, E2, SA, domtime, Ra, L2, PS, delt]
        self._rows.append(row)
        assert not np.any(np.isnan(np.asarray(row, dtype=np.float64)))  # timedata.py:10

    def data(self):
        return np.asarray(self._rows, dtype=np.float64).reshape(-1, 9)

    def col(self, j):
        return self.data()[:, j]

    def energy_falls(self, it):
        """timedata.py:51-63"""
        E2 = self._rows
        return E2[it - 1][2] > E2[it][2] > E2[0][2]


class OracleSolver:
    """Restates chsimpy/solver.py:45-252 (+ solution.py:25-61 constants)."""

    def __init__(self, params: OracleParams, U_init=None):
        p = self.params = params
        N = p.N
        # solution.py:25-55
        self.Am = (25.13 * 1e6 / p.N_A) ** (2 / 3) * p.N_A
        self.delx = p.L / (N - 1)
        self.delx2 = self.delx ** 2
        self.RT = p.R * p.temp
        self.BRT = p.B * p.R * p.temp
        self.Amr = 1 / self.Am
        self.A0 = p.func_A0(p.temp)
        self.A1 = p.func_A1(p.temp)
        self.kappa_tilde = p.kappa_tilde
        self.CHeig, self.Seig = get_coefficients(N, self.kappa_tilde, p.delt, self.delx2)
        # solution.py:57-61
        self.tau0 = 0
        self.t0 = 0
        self.computed_steps = 0
        self.stop_reason = 'None'
        self.U = None
        self.timedata = None
        # solver.py:50-54
        self.skip_check = False
        self.time_delta_sum = 0.0
        self.time_passed = 0.0
        self._prepared = False
        self.delt = p.delt
        self.create_rand = None
        # solver.py:59-82
        if U_init is not None:
            if U_init.shape != (N, N):
                raise SystemExit(1)  # solver.py:63-64
            self.U_init = U_init
        elif p.generator == 'lcg':
            self.U_init = p.XXX + (p.XXX * 0.01 * lcg_sample(N, N, p.seed))  # solver.py:66 (no -0.5)
        elif p.generator == 'sobol':
            from scipy.stats import qmc
            qrng = qmc.Sobol(d=N, seed=p.seed)
            self.create_rand = lambda n: qrng.random(n)
            self.U_init = p.XXX + (p.XXX * 0.01 * (self.create_rand(N) - 0.5))
        elif p.generator == 'simplex':
            # solver.py:72-75 (third-party `opensimplex`, requirements.txt: opensimplex~=0.4; not in this
            # image: the import error surfaces here exactly as a missing dependency does in the reference)
            import opensimplex
            self.create_rand = lambda n: opensimplex.noise2array(np.linspace(0, 48, n), np.linspace(0, 48, n))
            self.U_init = p.XXX + (p.XXX * 0.01 * (self.create_rand(N) - 0.5))
        else:
            rng = np.random.Generator(np.random.PCG64(p.seed))
            self.create_rand = lambda n: rng.random((n, n))
            self.U_init = p.XXX + (p.XXX * 0.01 * (self.create_rand(N) - 0.5))

    # -- shared by prepare() and the loop: solver.py:100-116 / 213-228 -------
    def _energies(self, U):
        p = self.params
        DUx, DUy = np.gradient(U, self.delx, axis=[0, 1], edge_order=1)
        Du2 = DUx ** 2 + DUy ** 2
        Uinv = 1 - U
        E2 = 0.5 * self.Amr * self.kappa_tilde * p.L ** 2 * np.mean(Du2)
        E = self.Amr * p.L ** 2 * np.mean(
            self.RT * (U * (np.log(U) - p.B) + Uinv * np.log(Uinv))
            + (self.A0 + self.A1 * (Uinv - U)) * U * Uinv) + E2
        return E, E2

    def _stats(self, U):
        N = self.params.N
        Um = U - np.mean(U)
        PS = np.sum(np.abs(Um)) / (N ** 2)
        r = int(N / 2) + 1
        Ra = np.mean(np.abs(U[r, :] - np.mean(U[r, :])))
        return PS, Ra

    def prepare(self):
        """solver.py:84-135"""
        U = self.U_init.copy()
        E, E2 = self._energies(U)
        PS, Ra = self._stats(U)
        self.timedata = OracleTimeData()
        self.timedata.insert(it=0, delt=self.delt, E=E, E2=E2, SA=0, domtime=0, Ra=Ra, L2=0, PS=PS)
        self.U = U
        self.tau0 = 0.0
        self.t0 = 0.0
        self.stop_reason = 'None'
        self.computed_steps = 1
        self._prepared = True

    def mu(self, U):
        """EnergieEut, solver.py:166-175"""
        Uinv = 1 - U
        U1Uinv = U / Uinv
        U2inv = Uinv - U
        return (self.RT * np.log(U1Uinv) - self.BRT
                + (self.A0 + self.A1 * U2inv) * U2inv
                - 2 * self.A1 * U * Uinv)

    def solve_or_resume(self, nsteps=None, record=None):
        """solver.py:137-252.  ``record(step_index, U)`` is an optional test hook."""
        assert self._prepared is True
        p = self.params
        N = p.N
        if nsteps is None:
            nsteps = max(p.ntmax, 0)
        time_limit = None
        if p.time_max is not None and p.time_max > 0:
            time_limit = p.time_max * 60
        Seig, CHeig = self.Seig, self.CHeig
        U = self.U
        hat_U = scifft.dctn(U, norm='ortho')  # solver.py:159
        itbegin = 1 if self.computed_steps == 1 else 0  # solver.py:160-163
        for it in range(itbegin, nsteps):
            EnergieEut = self.mu(U)
            if p.adaptive_time and self.computed_steps > 500 and np.remainder(self.computed_steps, 2) == 0:
                delt_alpha = 500 / (2) ** 3
                delt_dyn = np.linalg.norm(p.delt_max / np.sqrt(1 + delt_alpha * np.abs(EnergieEut) ** 2), ord=-1)
                delt_new = max(p.delt, delt_dyn)
                if delt_new / self.delt > 1.15:
                    self.delt = 0.75 * self.delt + 0.25 * delt_new
                else:
                    self.delt = delt_new
                CHeig, Seig = get_coefficients(N, self.kappa_tilde, self.delt, self.delx2)
            self.time_delta_sum += self.delt
            self.time_passed = self.time_delta_sum / p.M_tilde
            if time_limit is not None and self.time_passed > time_limit:
                self.stop_reason = 'time-limit'
                break
            hat_rhs = hat_U + Seig * scifft.dctn(EnergieEut, norm='ortho')
            hat_U = hat_rhs / CHeig
            U = scifft.idctn(hat_U, norm='ortho')
            if p.jitter is not None and 0.0 < p.jitter < 0.1:
                U += p.jitter * (2 * self.create_rand(N) - 1)
            E, E2 = self._energies(U)
            PS, Ra = self._stats(U)
            L2 = np.linalg.norm(EnergieEut) / N ** 2
            SA = np.sum(U < p.threshold) / (N ** 2)
            domtime = self.time_passed ** (1 / 3)
            self.timedata.insert(it=self.computed_steps, delt=self.delt, E=E, E2=E2, SA=SA,
                                 domtime=domtime, Ra=Ra, L2=L2, PS=PS)
            self.computed_steps += 1
            if record is not None:
                record(self.computed_steps - 1, U)
            if not self.skip_check and self.timedata.energy_falls(self.computed_steps - 1):
                self.tau0 = self.computed_steps
                self.t0 = self.time_passed
                if not p.full_sim:
                    self.stop_reason = 'energy'
                    break
                else:
                    self.skip_check = True
        self.U = U
        return self


# --------------------------------------------------------------------------
# a18: direct-summation DCT-II / DCT-III ('ortho'), extended precision.
# Used to pin the transform convention independently of scipy.
# --------------------------------------------------------------------------


def dct2_ortho_direct(x: np.ndarray) -> np.ndarray:
    """X_k = f_k * 2 * sum_n x_n cos(pi k (2n+1) / (2N)), f_0=sqrt(1/4N), f_k=sqrt(1/2N)."""
    x = np.asarray(x, dtype=np.longdouble)
    N = x.shape[-1]
    n = np.arange(N, dtype=np.longdouble)
    k = n.reshape(-1, 1)
    C = np.cos(np.pi * k * (2 * n + 1) / (2 * N))
    f = np.full(N, np.sqrt(np.longdouble(1) / (2 * N)))
    f[0] = np.sqrt(np.longdouble(1) / (4 * N))
    return ((2 * f).reshape(-1, 1) * C) @ x


def dct3_ortho_direct(X: np.ndarray) -> np.ndarray:
    X = np.asarray(X, dtype=np.longdouble)
    N = X.shape[-1]
    n = np.arange(N, dtype=np.longdouble)
    k = n.reshape(-1, 1)
    C = np.cos(np.pi * k * (2 * n + 1) / (2 * N))
    f = np.full(N, np.sqrt(np.longdouble(1) / (2 * N)))
    f[0] = np.sqrt(np.longdouble(1) / (4 * N))
    return ((2 * f).reshape(-1, 1) * C).T @ X


def make_params(N, ntmax, **kw):
    """The synthetic configuration of SURVEY.md section 8(d)."""
    base = dict(N=N, ntmax=ntmax, full_sim=True, kappa_tilde=0.0002989112919661156)
    base.update(kw)
    return OracleParams(**base)
