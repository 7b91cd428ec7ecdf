"""ctypes binding of ``libchs_hip.so`` (C ABI: ``include/chs_hip.h``).

The HIP library *is* the product path: there is no CPU fallback.  A missing
library, a missing symbol or a missing GPU is an error raised here, loudly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', 'libchs_hip.so')

CHS_OK, CHS_EINVAL, CHS_EHIP, CHS_ENAN, CHS_ESTATE = 0, -1, -2, -3, -4
CHS_F64, CHS_F32 = 0, 1
CHS_ENGINE_AUTO, CHS_ENGINE_DIRECT, CHS_ENGINE_FAST = 0, 1, 2
CHS_STOP_NONE, CHS_STOP_ENERGY, CHS_STOP_TIME_LIMIT = 0, 1, 2
CHS_STEP_CARRY_HAT = 1
CHS_STEP_REDERIVE_HAT = 2
CHS_STEP_LAST_CALL = 4
CHS_STEP_KEEP_T1 = 8
CHS_NKERNELS = 8

STOP_NAMES = {CHS_STOP_NONE: 'None', CHS_STOP_ENERGY: 'energy', CHS_STOP_TIME_LIMIT: 'time-limit'}
STOP_CODES = {v: k for k, v in STOP_NAMES.items()}
ENGINES = {'auto': CHS_ENGINE_AUTO, 'direct': CHS_ENGINE_DIRECT, 'fast': CHS_ENGINE_FAST}
DTYPES = {'float64': CHS_F64, 'f64': CHS_F64, 'float32': CHS_F32, 'f32': CHS_F32}

# every symbol include/chs_hip.h declares
SYMBOLS = (
    'chs_create', 'chs_destroy', 'chs_pool_clear', 'chs_pool_count', 'chs_set_U', 'chs_init_U_pcg64', 'chs_get_U', 'chs_prepare', 'chs_step_n',
    'chs_get_state', 'chs_set_state', 'chs_set_jitter_noise', 'chs_set_jitter_pcg64', 'chs_dctn', 'chs_get_mu', 'chs_test_math',
    'chs_engine', 'chs_kernel_name', 'chs_profile_steps', 'chs_last_step_ms',
    'chs_last_error', 'chs_version',
)


class chs_consts(C.Structure):
    _fields_ = [('N', C.c_int32), ('dtype', C.c_int32), ('device', C.c_int32), ('engine', C.c_int32),
                ('adaptive_time', C.c_int32), ('full_sim', C.c_int32),
                ('RT', C.c_double), ('BRT', C.c_double), ('B', C.c_double), ('A0', C.c_double),
                ('A1', C.c_double), ('Amr', C.c_double), ('kappa_tilde', C.c_double), ('L', C.c_double),
                ('delx', C.c_double), ('delt', C.c_double), ('delt_max', C.c_double),
                ('M_tilde', C.c_double), ('threshold', C.c_double), ('time_limit_s', C.c_double)]


class chs_state(C.Structure):
    _fields_ = [('delt', C.c_double), ('time_delta_sum', C.c_double), ('time_passed', C.c_double),
                ('tau0', C.c_double), ('t0', C.c_double), ('computed_steps', C.c_int64),
                ('skip_check', C.c_int32), ('stop_reason', C.c_int32)]


class EngineError(RuntimeError):
    pass


_lib = None


def resolve_library():
    """The library to load and whether it is the product.  CHS_LIB_PATH selects another build (an experiment
    variant under lib/variants/, tools/ab.sh) -- said aloud on stderr, never silently.  The product library
    (lib/libchs_hip.so) must carry the sha256 of the sources in the tree (chsimpy_amd/_build.py): one that does
    not -- a variant copied over it, a library older than an edit -- is rebuilt when hipcc is at hand and refused
    otherwise (CHS_NO_REBUILD=1: always refused)."""
    import sys
    from . import _build
    override = os.environ.get('CHS_LIB_PATH')
    if override:
        have, flags = _build.embedded_provenance(override)
        sys.stderr.write(f"chsimpy_amd: CHS_LIB_PATH -> {override} (sources {str(have)[:12]}, extra flags '{flags}'): "
                         "NOT the product library\n")
        return override, False
    path = LIB_PATH
    want = _build.source_hash()
    if want is None or not os.path.exists(path):
        return path, True        # no sources to check against (an installed copy) / missing: load() reports it
    have, flags = _build.embedded_provenance(path)
    if have == want and not flags:
        return path, True
    what = (f"{path} was built from other sources than the tree holds (library {str(have)[:12]}, tree {want[:12]}"
            + (f", extra flags '{flags}'" if flags else '') + ")")
    if os.environ.get('CHS_NO_REBUILD') == '1':
        raise EngineError(what + "; rebuild it with `python -c 'import __graft_entry__ as g; g.build()'`")
    try:
        _build.build_hip()
    except Exception as e:
        raise EngineError(what + f"; rebuilding it failed: {e}") from e
    return path, True


def load():
    """Load the HIP library once; raise if it (or any declared symbol) is missing or it is not built from the
    sources in the tree."""
    global _lib
    if _lib is not None:
        return _lib
    path, _product = resolve_library()
    if not os.path.exists(path):
        raise EngineError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). chsimpy_amd has no CPU fallback.")
    lib = C.CDLL(path)
    for s in SYMBOLS:
        if not hasattr(lib, s):
            raise EngineError(f"{path} does not export `{s}` (declared in include/chs_hip.h)")
    dp = C.POINTER(C.c_double)
    lib.chs_create.argtypes = [C.POINTER(chs_consts), dp, C.POINTER(C.c_void_p)]
    lib.chs_destroy.argtypes = [C.c_void_p]
    lib.chs_set_U.argtypes = [C.c_void_p, dp]
    lib.chs_get_U.argtypes = [C.c_void_p, dp]
    lib.chs_prepare.argtypes = [C.c_void_p, dp]
    lib.chs_step_n.argtypes = [C.c_void_p, C.c_int64, C.c_int32, dp, C.POINTER(C.c_int64)]
    lib.chs_get_state.argtypes = [C.c_void_p, C.POINTER(chs_state)]
    lib.chs_set_state.argtypes = [C.c_void_p, C.POINTER(chs_state)]
    lib.chs_set_jitter_noise.argtypes = [C.c_void_p, C.c_double, dp]
    lib.chs_set_jitter_pcg64.argtypes = [C.c_void_p, C.c_double, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.chs_init_U_pcg64.argtypes = [C.c_void_p, C.c_double, C.c_double, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.chs_dctn.argtypes = [C.c_void_p, dp, dp, C.c_int]
    lib.chs_get_mu.argtypes = [C.c_void_p, dp]
    lib.chs_engine.argtypes = [C.c_void_p]
    lib.chs_test_math.argtypes = [C.c_int, C.c_int, dp, dp, dp, C.c_int64]
    lib.chs_kernel_name.argtypes = [C.c_void_p, C.c_int]
    lib.chs_kernel_name.restype = C.c_char_p
    lib.chs_profile_steps.argtypes = [C.c_void_p, C.c_int64, dp, C.POINTER(C.c_int64)]
    lib.chs_last_step_ms.argtypes = [C.c_void_p]
    lib.chs_last_step_ms.restype = C.c_double
    lib.chs_last_error.restype = C.c_char_p
    lib.chs_version.restype = C.c_char_p
    lib.chs_pool_clear.argtypes = []
    # engines parked by chs_destroy are device memory of this process: hand them back at interpreter exit
    import atexit
    atexit.register(lib.chs_pool_clear)
    _lib = lib
    return lib


def pool_clear():
    """Free the engines `chs_destroy` has parked for reuse (include/chs_hip.h: chs_pool_clear)."""
    load().chs_pool_clear()


def pool_count():
    """Number of engines parked for reuse right now."""
    return int(load().chs_pool_count())


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _as_f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and a.shape != shape:
        raise ValueError(f"expected shape {shape}, got {a.shape}")
    return a


def test_math(which, a, b=None, device=0):
    """Device math primitives (accuracy tests): see chs_test_math in include/chs_hip.h."""
    lib = load()
    a = np.ascontiguousarray(a, dtype=np.float64).ravel()
    out = np.empty_like(a)
    if b is None:
        b = np.zeros(4)
    b = np.ascontiguousarray(b, dtype=np.float64).ravel()
    rc = lib.chs_test_math(device, which, _dptr(a), _dptr(b), _dptr(out), a.size)
    if rc != CHS_OK:
        raise EngineError(f"chs_test_math failed ({rc}): {lib.chs_last_error().decode()}")
    return out


class Engine:
    """One device-resident simulation (an opaque ``chs_handle``)."""

    def __init__(self, consts: chs_consts, lam):
        self.lib = load()
        self.N = int(consts.N)
        self._h = C.c_void_p()
        lam = _as_f64(lam, (self.N,))
        rc = self.lib.chs_create(C.byref(consts), _dptr(lam), C.byref(self._h))
        self._check(rc, 'chs_create')

    # -- error mapping --------------------------------------------------------
    def _check(self, rc, what):
        if rc == CHS_OK:
            return
        msg = self.lib.chs_last_error().decode(errors='replace')
        if rc == CHS_ENAN:
            raise AssertionError(f"{what}: {msg}")  # the reference asserts (timedata.py:10)
        if rc == CHS_ESTATE:
            raise AssertionError(f"{what}: {msg}")  # solver.py:139
        raise EngineError(f"{what} failed ({rc}): {msg}")

    def close(self):
        if getattr(self, '_h', None) is not None and self._h.value:
            self.lib.chs_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- field ----------------------------------------------------------------
    def set_U(self, U):
        U = _as_f64(U, (self.N, self.N))
        self._check(self.lib.chs_set_U(self._h, _dptr(U)), 'chs_set_U')

    def get_U(self):
        U = np.empty((self.N, self.N), dtype=np.float64)
        self._check(self.lib.chs_get_U(self._h, _dptr(U)), 'chs_get_U')
        return U

    # -- loop -----------------------------------------------------------------
    def prepare(self):
        row = np.empty(9, dtype=np.float64)
        self._check(self.lib.chs_prepare(self._h, _dptr(row)), 'chs_prepare')
        return row

    def step_n(self, nsteps, carry_hat=False, rederive_hat=False, last_call=False, keep_t1=False):
        """Returns (rows[k,9], rc) -- rc is CHS_OK or CHS_ENAN (rows then end with the NaN row)."""
        nsteps = int(max(nsteps, 0))
        rows = np.empty((max(nsteps, 1), 9), dtype=np.float64)
        done = C.c_int64(0)
        rc = self.lib.chs_step_n(self._h, nsteps, (CHS_STEP_CARRY_HAT if carry_hat else 0) | (CHS_STEP_REDERIVE_HAT if rederive_hat else 0)
                                 | (CHS_STEP_LAST_CALL if last_call else 0) | (CHS_STEP_KEEP_T1 if keep_t1 else 0),
                                 _dptr(rows), C.byref(done))
        if rc not in (CHS_OK, CHS_ENAN):
            self._check(rc, 'chs_step_n')
        return rows[:done.value].copy(), rc

    def get_state(self):
        s = chs_state()
        self._check(self.lib.chs_get_state(self._h, C.byref(s)), 'chs_get_state')
        return s

    def set_state(self, s):
        self._check(self.lib.chs_set_state(self._h, C.byref(s)), 'chs_set_state')

    def set_jitter_noise(self, jitter, noise):
        if noise is None:
            self._check(self.lib.chs_set_jitter_noise(self._h, float(jitter or 0.0), None), 'chs_set_jitter_noise')
        else:
            noise = _as_f64(noise, (self.N, self.N))
            self._check(self.lib.chs_set_jitter_noise(self._h, float(jitter), _dptr(noise)), 'chs_set_jitter_noise')

    def init_U_pcg64(self, base, scale, state, inc):
        """U = base + scale*(rand - 0.5) drawn on the device from numpy's PCG64 stream (chs_init_U_pcg64)."""
        m64 = (1 << 64) - 1
        st = (C.c_uint64 * 2)((int(state) >> 64) & m64, int(state) & m64)
        ic = (C.c_uint64 * 2)((int(inc) >> 64) & m64, int(inc) & m64)
        self._check(self.lib.chs_init_U_pcg64(self._h, float(base), float(scale), st, ic), 'chs_init_U_pcg64')

    def set_jitter_pcg64(self, jitter, state, inc):
        """Jitter noise drawn on the device from numpy's PCG64 stream: `state`, `inc` = the two 128-bit
        integers of ``Generator.bit_generator.state['state']``."""
        m64 = (1 << 64) - 1
        st = (C.c_uint64 * 2)((int(state) >> 64) & m64, int(state) & m64)
        ic = (C.c_uint64 * 2)((int(inc) >> 64) & m64, int(inc) & m64)
        self._check(self.lib.chs_set_jitter_pcg64(self._h, float(jitter), st, ic), 'chs_set_jitter_pcg64')

    # -- hooks ----------------------------------------------------------------
    def dctn(self, X, inverse=False):
        X = _as_f64(X, (self.N, self.N))
        Y = np.empty_like(X)
        self._check(self.lib.chs_dctn(self._h, _dptr(X), _dptr(Y), 1 if inverse else 0), 'chs_dctn')
        return Y

    def get_mu(self):
        M = np.empty((self.N, self.N), dtype=np.float64)
        self._check(self.lib.chs_get_mu(self._h, _dptr(M)), 'chs_get_mu')
        return M

    @property
    def engine(self):
        return {CHS_ENGINE_DIRECT: 'direct', CHS_ENGINE_FAST: 'fast'}[self.lib.chs_engine(self._h)]

    def kernel_names(self):
        out = []
        for i in range(CHS_NKERNELS):
            n = self.lib.chs_kernel_name(self._h, i)
            out.append(n.decode() if n else None)
        return out

    def profile_steps(self, nsteps):
        ms = np.zeros(CHS_NKERNELS, dtype=np.float64)
        calls = np.zeros(CHS_NKERNELS, dtype=np.int64)
        rc = self.lib.chs_profile_steps(self._h, int(nsteps), _dptr(ms), calls.ctypes.data_as(C.POINTER(C.c_int64)))
        self._check(rc, 'chs_profile_steps')
        return ms, calls

    def last_step_ms(self):
        return float(self.lib.chs_last_step_ms(self._h))
