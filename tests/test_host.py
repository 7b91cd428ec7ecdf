"""CPU tests of the host-side mirror (no compute calls: there is no GPU here)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import chsimpy_amd
from chsimpy_amd import _lib, mport, utils
from oracle import chs_oracle as orc
from test_oracle import LCG_KAT

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lcg_known_answer_product():
    # same vector as the reference's tests/test.py:25-37, same assertion
    assert np.allclose(mport.matlab_lcg_sample(5, 4, 2023), LCG_KAT)
    assert np.array_equal(mport.matlab_lcg_sample(7, 3, 11), orc.lcg_sample(7, 3, 11))


def test_library_loads_and_exports_every_declared_symbol(hip_lib):
    hdr = open(os.path.join(ROOT, 'include', 'chs_hip.h')).read()
    declared = set(re.findall(r'\b(chs_[a-z_0-9A-Z]+)\s*\(', hdr)) - {'chs_handle_s'}
    assert declared, "no prototypes found"
    for name in sorted(declared):
        assert hasattr(hip_lib, name), f"{name} declared in include/chs_hip.h but not exported"
    assert declared == set(_lib.SYMBOLS)
    assert b'gfx950' in hip_lib.chs_version()


def test_header_constants_match_the_binding():
    """Every `#define CHS_<NAME> <integer>` of include/chs_hip.h that the ctypes binding mirrors has the same value there."""
    hdr = open(os.path.join(ROOT, 'include', 'chs_hip.h')).read()
    defs = {k: int(v, 0) for k, v in re.findall(r'#define\s+(CHS_[A-Z_0-9]+)\s+(-?(?:0x[0-9a-fA-F]+|\d+))\b', hdr)}
    for name in ('CHS_STEP_CARRY_HAT', 'CHS_STEP_REDERIVE_HAT', 'CHS_STEP_LAST_CALL', 'CHS_STEP_KEEP_T1'):
        assert name in defs, name
    mirrored = [k for k in defs if hasattr(_lib, k)]
    assert len(mirrored) >= 6
    for k in mirrored:
        assert getattr(_lib, k) == defs[k], k


def test_struct_layouts_match_header():
    assert ctypes.sizeof(_lib.chs_consts) == 6 * 4 + 14 * 8
    assert ctypes.sizeof(_lib.chs_state) == 5 * 8 + 8 + 2 * 4


def test_solution_constants_match_oracle():
    p = chsimpy_amd.Parameters()
    p.N = 128
    p.kappa_tilde = 0.0002989112919661156
    sol = chsimpy_amd.Solution(p)
    o = orc.OracleSolver(orc.make_params(128, 2))
    for k in ('Am', 'delx', 'delx2', 'RT', 'BRT', 'Amr', 'A0', 'A1', 'kappa_tilde'):
        assert getattr(sol, k) == getattr(o, k), k
    assert np.array_equal(sol.CHeig, o.CHeig) and np.array_equal(sol.Seig, o.Seig)
    assert np.array_equal(utils.eigenvalues(16), orc.eigenvalues(16))


def test_default_kappa_matches_reference_observation():
    sympy = pytest.importorskip('sympy')
    p = chsimpy_amd.Parameters()
    p.N = 16
    sol = chsimpy_amd.Solution(p)
    # observed from the reference with sympy 1.14
    if sympy.__version__.startswith('1.14'):
        assert sol.kappa_base == pytest.approx(0.03144365587960252, rel=1e-12)
        assert sol.kappa_tilde == pytest.approx(0.0002989112919661156, rel=1e-12)
    else:
        assert sol.kappa_tilde == pytest.approx(0.0002989112919661156, rel=1e-5)


@pytest.mark.parametrize("gen", ['uniform', 'lcg', 'sobol'])
def test_U_init_generators_match_oracle(gen):
    p = chsimpy_amd.Parameters()
    p.N, p.generator, p.kappa_tilde = 32, gen, 3e-4
    s = chsimpy_amd.Solver(p)
    o = orc.OracleSolver(orc.make_params(32, 2, generator=gen))
    assert np.array_equal(s.U_init, o.U_init)


def test_wrong_shape_exits_like_the_reference():
    p = chsimpy_amd.Parameters()
    p.N, p.kappa_tilde = 16, 3e-4
    with pytest.raises(SystemExit):
        chsimpy_amd.Solver(p, U_init=np.zeros((4, 4)))


def test_not_prepared_asserts():
    p = chsimpy_amd.Parameters()
    p.N, p.kappa_tilde = 16, 3e-4
    s = chsimpy_amd.Solver(p)
    with pytest.raises(AssertionError):
        s.solve_or_resume(3)


def test_no_gpu_fails_loudly_not_silently():
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is present")
    p = chsimpy_amd.Parameters()
    p.N, p.kappa_tilde = 16, 3e-4
    s = chsimpy_amd.Solver(p)
    with pytest.raises(_lib.EngineError):
        s.prepare()


def test_timedata_api():
    td = chsimpy_amd.TimeData()
    td.insert(it=0, delt=1.0, E=-1.0, E2=1.0, SA=0, domtime=0, Ra=0.1, L2=0, PS=0.2)
    td.extend(np.array([[1, -1.1, 3.0, 0.5, 0.1, 0.1, 0.2, 0.2, 1.0], [2, -1.2, 2.0, 0.5, 0.2, 0.1, 0.2, 0.2, 1.0]]))
    assert td.data().shape == (3, 9)
    assert list(td.E2) == [1.0, 3.0, 2.0] and list(td.it_range) == [0, 1, 2]
    assert td.energy_falls(2) is True and td.energy_falls(1) is False
    with pytest.raises(AssertionError):
        td.insert(it=3, delt=1.0, E=float('nan'), E2=1.0, SA=0, domtime=0, Ra=0, L2=0, PS=0)
    for _ in range(200):
        td.extend(np.ones((3, 9)))
    assert len(td) == 604


def test_csv_roundtrip(tmp_path):
    A = np.random.default_rng(1).random((5, 7))
    for ext in ('csv', 'csv.bz2'):
        f = str(tmp_path / f"a.{ext}")
        utils.csv_export_matrix(A, f)
        assert np.allclose(utils.csv_import_matrix(f), A, rtol=1e-13, atol=0)  # pandas fast float parser
    v = np.arange(4.0)
    f = str(tmp_path / "v.csv")
    utils.csv_export_matrix(v, f)
    assert np.array_equal(utils.csv_import_matrix(f), v)


def test_log_table_header_is_what_the_generator_writes():
    """chsimpy_amd/csrc/chs_log_table.h (the reduction table of the device log) is generated by
    tools/log_table.py: {fl(256/i), fl(-log(fl(256/i)))} for i = 128..256, the entry of i = 256 exact."""
    mp = pytest.importorskip('mpmath')
    import re
    mp.mp.dps = 60
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    txt = open(os.path.join(root, 'chsimpy_amd', 'csrc', 'chs_log_table.h')).read()
    rows = re.findall(r'\{(-?0x[0-9a-f.]+p[+-]\d+), (-?0x[0-9a-f.]+p[+-]\d+)\},\s*// i = (\d+)', txt)
    assert len(rows) == 129 and int(rows[0][2]) == 128 and int(rows[-1][2]) == 256
    for rc_s, lc_s, i_s in rows:
        i = int(i_s)
        rc, lc = float.fromhex(rc_s), float.fromhex(lc_s)
        assert rc == float(mp.mpf(256) / i)
        assert lc == float(-mp.log(mp.mpf(rc)))
        if i == 256:
            assert rc == 1.0 and lc == 0.0
        # the reduced argument stays inside the series' range for every m that rounds to i
        assert abs((i + 0.5) / 256.0 * rc - 1.0) <= 1 / 256 + 1e-12 and abs((i - 0.5) / 256.0 * rc - 1.0) <= 1 / 256 + 1e-12


def test_simplex_generator_wiring(monkeypatch):
    """generator='simplex' (solver.py:72-75): `opensimplex.noise2array(linspace(0,48,N), linspace(0,48,N))`
    feeds the usual start-field expression.  The package is a third-party dependency that this image does
    not ship: a stand-in module checks the wiring of product and oracle; without it both raise ImportError."""
    import sys
    import types
    import chsimpy_amd
    from oracle import chs_oracle as orc
    N = 24
    p = chsimpy_amd.Parameters()
    p.N, p.kappa_tilde, p.generator = N, 3e-4, 'simplex'
    if 'opensimplex' not in sys.modules:
        try:
            import opensimplex  # noqa: F401
        except ImportError:
            with pytest.raises(ImportError):
                chsimpy_amd.Solver(p)
            with pytest.raises(ImportError):
                orc.OracleSolver(orc.make_params(N, 2, generator='simplex'))
    calls = []

    def noise2array(x, y):
        calls.append((x.copy(), y.copy()))
        return np.sin(np.outer(y, np.ones_like(x)) * 0.3) * np.cos(np.outer(np.ones_like(y), x) * 0.2)

    monkeypatch.setitem(sys.modules, 'opensimplex', types.SimpleNamespace(noise2array=noise2array))
    s = chsimpy_amd.Solver(p)
    o = orc.OracleSolver(orc.make_params(N, 2, generator='simplex'))
    ls = np.linspace(0, 48, N)
    expect = p.XXX + (p.XXX * 0.01 * (noise2array(ls, ls) - 0.5))
    assert np.array_equal(s.U_init, expect) and np.array_equal(o.U_init, expect)
    assert all(np.array_equal(a, ls) and np.array_equal(b, ls) for a, b in calls)
    assert np.array_equal(s.create_rand(N), noise2array(ls, ls))   # kept for the jitter draws (solver.py:211)


def test_simulator_chunk_sizes():
    """Chunked driving (simulator.py:56-81): update_every steps per call, the last call shortened to hit ntmax."""
    from chsimpy_amd.simulator import chunk_sizes
    import itertools
    assert list(chunk_sizes(47, 10)) == [10, 10, 10, 10, 7]
    assert list(chunk_sizes(40, 10)) == [10, 10, 10, 10]
    assert list(chunk_sizes(7, 10)) == [7]            # dsteps = min(steps_end, update_every)
    assert list(chunk_sizes(0, 10)) == []
    assert list(itertools.islice(chunk_sizes(None, 100), 3)) == [100, 100, 100]   # time_max set: only the limit ends the run


def test_in_place_edit_of_a_downloaded_field_is_noticed():
    """`sol.U[i, j] = x` on the array the solution handed out (an update callback editing the field) must be seen
    by the next solve_or_resume, which the reference starts from that very array (solver.py:158)."""
    import chsimpy_amd
    p = chsimpy_amd.Parameters()
    p.N, p.kappa_tilde = 16, 0.0002989112919661156
    sol = chsimpy_amd.Solution(p)
    dev = np.full((16, 16), 0.875)
    sol._bind_device_U(None, lambda: dev.copy())
    assert not sol._host_edited()          # nothing downloaded yet: nothing to compare
    u = sol.U
    assert not sol._host_edited()
    u[3, 4] = 0.87
    assert sol._host_edited()
    sol._bind_device_U(None, lambda: dev.copy())
    assert not sol._host_edited()
    sol.U = dev                            # assignment takes the other road (dirty flag)
    assert sol.__dict__['_U_dirty'] and not sol._host_edited()


def test_edit_of_a_host_mirror_between_prepare_and_solve_is_noticed():
    """prepare() with a host U_init binds that array as the mirror of the device field (`_bind_device_U(host_copy)`):
    an edit in place before the first solve_or_resume is what the reference would start from (solver.py:158), so it
    must be noticed as well; a field holding NaN must not look edited forever (NaN-safe comparison)."""
    p = chsimpy_amd.Parameters()
    p.N, p.kappa_tilde = 16, 0.0002989112919661156
    sol = chsimpy_amd.Solution(p)
    host = np.full((16, 16), 0.875)
    sol._bind_device_U(host)
    assert not sol._host_edited()
    host[2, 3] = 0.9
    assert sol._host_edited()
    bad = np.full((16, 16), 0.875)
    bad[0, 0] = np.nan
    sol._bind_device_U(bad)
    assert not sol._host_edited()          # NaN == NaN for this purpose
    sol._bind_device_U(host, track=False)  # close(fetch_U=False): nothing to compare afterwards
    assert sol.__dict__['_U_print'] is None


def test_download_path_never_enters_blas(monkeypatch):
    """Round 3's host-side trap: a BLAS call (np.vdot) in the download path woke a thread pool sized for every core of
    the host, which used up the container's CPU quota and throttled the NEXT run's kernel launches.  The mechanism, not
    the pace, is asserted: with numpy's BLAS-backed entry points booby-trapped, fingerprinting, the lazy download and
    the in-place-edit check still run."""
    def boom(*a, **k):
        raise AssertionError("BLAS entry point called from the download path")
    for name in ('dot', 'vdot', 'inner', 'matmul', 'tensordot'):
        monkeypatch.setattr(np, name, boom)
    monkeypatch.setattr(np.linalg, 'norm', boom)
    from chsimpy_amd import solution
    p = chsimpy_amd.Parameters()
    p.N, p.kappa_tilde = 64, 0.0002989112919661156
    sol = chsimpy_amd.Solution(p)
    dev = np.random.default_rng(1).random((64, 64))
    sol._bind_device_U(None, lambda: dev.copy())
    u = sol.U
    assert not sol._host_edited()
    u[0, 0] += 1.0
    assert sol._host_edited()
    assert solution._fingerprint(dev.ravel())[0] == (4096,)
    # np.einsum('ij,ij->') must not route through BLAS either (it does only with optimize=True / tensordot paths)
    import inspect
    assert 'optimize' not in inspect.getsource(solution._fingerprint)


def test_a_stale_or_foreign_library_is_refused(tmp_path, monkeypatch, hip_lib):
    """Build provenance (chsimpy_amd/_build.py): the product library carries the sha256 of the sources it was built
    from.  A library whose hash is not the tree's -- an experiment variant copied over lib/libchs_hip.so, a build
    older than an edit -- is refused by the loader (CHS_NO_REBUILD=1; without it the loader rebuilds), whatever its
    modification time says; the real library passes; CHS_LIB_PATH loads a variant only when told to, aloud."""
    from chsimpy_amd import _build
    assert _build.is_current(_lib.LIB_PATH)
    have, flags = _build.embedded_provenance(_lib.LIB_PATH)
    assert have == _build.source_hash() and flags == ''
    assert have in _lib.load().chs_version().decode()          # chs_version() reports it
    blob = open(_lib.LIB_PATH, 'rb').read()
    # (a) a library built from other sources: same file, another hash -- and NEWER than every source
    stale = tmp_path / 'libchs_hip.so'
    other = ('0' * 64).encode()
    stale.write_bytes(blob.replace(have.encode(), other))
    assert not _build.is_current(str(stale))
    monkeypatch.setattr(_lib, 'LIB_PATH', str(stale))
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setenv('CHS_NO_REBUILD', '1')
    monkeypatch.delenv('CHS_LIB_PATH', raising=False)
    with pytest.raises(_lib.EngineError, match='built from other sources'):
        _lib.load()
    # (b) the right sources but experiment flags (a variant left behind as the product)
    variant = tmp_path / 'variant.so'
    variant.write_bytes(blob.replace(b'CHS_FLAGS=;', b'CHS_FLAGS=-DX;', 1))
    assert _build.embedded_provenance(str(variant)) == (have, '-DX')
    monkeypatch.setattr(_lib, 'LIB_PATH', str(variant))
    with pytest.raises(_lib.EngineError, match="extra flags '-DX'"):
        _lib.load()
    # (c) without CHS_NO_REBUILD the loader rebuilds instead of loading it
    called = []
    monkeypatch.delenv('CHS_NO_REBUILD')
    monkeypatch.setattr(_build, 'build_hip', lambda *a, **k: called.append(1) or (_ for _ in ()).throw(RuntimeError('no hipcc here')))
    with pytest.raises(_lib.EngineError, match='rebuilding it failed'):
        _lib.load()
    assert called
    # (d) an explicit CHS_LIB_PATH is honoured and announced
    monkeypatch.setenv('CHS_LIB_PATH', str(variant))
    path, product = _lib.resolve_library()
    assert path == str(variant) and product is False


def test_step_kernels_do_not_spill():
    """No scratch in any row kernel at N = 4096 and N = 8192, either element type (tools/scratch_check.py: both
    translation units compiled to gfx950 assembly, `ScratchSize` of every k_row_* instantiation; no GPU needed).  A
    spilled value's reload sits in the in-order vector-memory queue behind every store issued before it: in a kernel
    that streams its results out, a 4-byte spill is a full drain."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'scratch_check.py'), '--quiet', '--row-zero'],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    # the dominant kernels of BASELINE.json's configs[2] / configs[3] / configs[4] sizes (k_col) are listed when they spill
    for line in r.stdout.splitlines():
        assert not re.search(r'k_col<FCfg<(double|float),(2048|4096|8192),', line), line
