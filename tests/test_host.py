"""CPU tests of the host-side mirror (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

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
