"""CPU tests: the oracle against every pin available for this path.

Pins: the LCG known-answer matrix of the reference's
tests/test.py:25-37; the figures observed from reference runs before the
permission denial (recorded in SURVEY.md, not regenerated); analytic known
answers; the transform convention against direct summation.
"""
import numpy as np
import pytest
import scipy.fftpack as scifft

from oracle import chs_oracle as orc

# tests/test.py:25-37 of the reference (a data fixture: 20 numbers)
LCG_KAT = np.array([
    [0.5475444293336684, 0.29257702841077793, 0.3117376865408093, 0.9844947126621821],
    [0.8031704429551821, 0.03775238992541674, 0.37862920778739695, 0.5387215616827465],
    [0.7217314246677474, 0.7984879318617694, 0.8011069301520972, 0.8502945903922872],
    [0.5455620291389348, 0.34767496602035824, 0.8863348965003783, 0.8019890788951838],
    [0.9676096443867356, 0.12967026239711338, 0.008214473728190397, 0.4722352030092083]])

KAPPA = 0.0002989112919661156


def test_lcg_known_answer():
    assert np.allclose(orc.lcg_sample(5, 4, 2023), LCG_KAT)
    assert np.array_equal(orc.lcg_sample(5, 4, 2023), LCG_KAT)


def test_default_constants():
    # SURVEY.md 8(c)(viii)
    s = orc.OracleSolver(orc.make_params(128, 2))
    assert s.A0 == pytest.approx(-151.26151, rel=1e-12)
    assert s.A1 == pytest.approx(-85.612615, rel=1e-12)
    assert s.RT == 7.675496165948127
    assert s.BRT == 98.7068806940929
    assert s.delx == 2 / 127


def test_reference_observations_n128():
    """Observed from the reference itself before the denial."""
    s = orc.OracleSolver(orc.make_params(128, 200))
    assert np.allclose(s.U_init[0, :3], [0.87139548, 0.87255385, 0.87161524], atol=5e-9)
    s.prepare()
    s.solve_or_resume()
    d = s.timedata.data()
    assert d.shape == (200, 9)
    assert d[0, 1] == pytest.approx(-5.453703816633322e-11, rel=1e-13)
    assert d[0, 2] == pytest.approx(2.2230041431712046e-18, rel=1e-12)
    assert d[-1, 1] == pytest.approx(-5.453709377934683e-11, rel=1e-13)
    assert d[-1, 2] == pytest.approx(8.73563379288026e-18, rel=1e-10)
    assert s.U.min() == pytest.approx(0.865457105573249, rel=1e-11)
    assert s.U.max() == pytest.approx(0.8871809239216062, rel=1e-11)
    assert s.stop_reason == 'None' and s.computed_steps == 200


@pytest.mark.parametrize("N", [8, 64, 128])
def test_dct_convention_direct_sum(N):
    rng = np.random.default_rng(N)
    x = rng.standard_normal(N)
    X = scifft.dct(x, type=2, norm='ortho')
    assert np.allclose(X, np.asarray(orc.dct2_ortho_direct(x), dtype=np.float64), rtol=0, atol=1e-13)
    assert np.allclose(scifft.idct(X, type=2, norm='ortho'),
                       np.asarray(orc.dct3_ortho_direct(X), dtype=np.float64), rtol=0, atol=1e-13)
    A = rng.standard_normal((N, N))
    assert np.allclose(scifft.idctn(scifft.dctn(A, norm='ortho'), norm='ortho'), A, atol=1e-12)
    # DC term of the 2-D transform = sum/N
    assert scifft.dctn(A, norm='ortho')[0, 0] == pytest.approx(A.sum() / N, rel=1e-12)


def test_fixed_point_uniform_field():
    N = 32
    p = orc.make_params(N, 5)
    s = orc.OracleSolver(p, U_init=np.full((N, N), 0.8))
    s.prepare()
    s.solve_or_resume()
    assert np.allclose(s.U, 0.8, rtol=0, atol=1e-14)
    d = s.timedata.data()
    assert np.all(np.abs(d[:, 2]) < 1e-40)  # E2 = 0
    c = 0.8
    e = s.RT * (c * (np.log(c) - p.B) + (1 - c) * np.log(1 - c)) + (s.A0 + s.A1 * (1 - 2 * c)) * c * (1 - c)
    assert d[-1, 1] == pytest.approx(s.Amr * p.L ** 2 * e, rel=1e-12)


def test_mass_conservation_and_single_mode_growth():
    N = 64
    c, eps, pm = 0.875, 1e-7, 3
    i = np.arange(N)
    mode = np.cos(np.pi * pm * (2 * i + 1) / (2 * N))
    U0 = c + eps * np.outer(mode, np.ones(N))
    p = orc.make_params(N, 3)
    s = orc.OracleSolver(p, U_init=U0)
    s.prepare()
    s.solve_or_resume()
    assert s.U.mean() == pytest.approx(U0.mean(), rel=1e-14)
    gpp = s.RT / (c * (1 - c)) - 2 * s.A0 - 6 * s.A1 * (1 - 2 * c)
    fac = (1 + s.Seig[pm, 0] * gpp) / s.CHeig[pm, 0]
    amp = (s.U[:, 0] - c) @ mode / (mode @ mode)
    assert amp == pytest.approx(eps * fac ** 2, rel=1e-5)


def test_norm_ord_minus1_is_min_column_abs_sum():
    A = np.random.default_rng(0).standard_normal((7, 5))
    assert np.linalg.norm(A, ord=-1) == pytest.approx(np.abs(A).sum(axis=0).min())


def test_gradient_on_ramp():
    N = 16
    dx = 2 / (N - 1)
    U = np.outer(np.arange(N) * dx * 3.0, np.ones(N)) + np.outer(np.ones(N), np.arange(N) * dx * -2.0)
    gx, gy = np.gradient(U, dx, axis=[0, 1], edge_order=1)
    assert np.allclose(gx, 3.0) and np.allclose(gy, -2.0)


def test_first_call_runs_nsteps_minus_one_and_resume():
    s = orc.OracleSolver(orc.make_params(32, 10))
    s.prepare()
    s.solve_or_resume(4)
    assert s.computed_steps == 4 and s.timedata.data().shape[0] == 4
    s.solve_or_resume(3)
    assert s.computed_steps == 7


def test_golden_fixture_matches_oracle():
    """The committed fixture (tests/golden/make_golden.py) is what the oracle produces."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'n64_lcg_40steps.npz'))
    p = orc.make_params(64, 40, generator='lcg', seed=2023)
    s = orc.OracleSolver(p)
    s.prepare()
    s.solve_or_resume()
    assert np.array_equal(s.U_init, g['U_init'])
    assert np.allclose(s.timedata.data(), g['timedata'], rtol=1e-12, atol=0)
    assert np.allclose(s.U, g['U_final'], rtol=1e-12, atol=0)
