"""GPU tests of the device math primitives (chsimpy_amd/csrc/chs_math.h) against numpy."""
import numpy as np
import pytest

from chsimpy_amd import _lib
from oracle import chs_oracle as orc

pytestmark = pytest.mark.gpu


def ulp_err(got, ref):
    return np.abs(got - ref) / np.spacing(np.abs(ref))


def test_log_accuracy(gpu):
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.random(200000), 1 - rng.random(200000) * 1e-3, rng.random(50000) * 1e-6,
                        np.exp(rng.uniform(-700, 700, 100000)), np.linspace(0.70, 0.72, 50001),
                        np.linspace(0.999, 1.001, 50001), [0.5, 1.0, 2.0, 0.875, 0.125]])
    got = _lib.test_math(0, x)
    ref = np.log(np.asarray(x, dtype=np.longdouble)).astype(np.float64)
    ok = ref != 0
    e = ulp_err(got[ok], ref[ok])
    assert e.max() <= 2.5, e.max()
    assert got[x == 1.0].tolist() == [0.0] * int((x == 1.0).sum())
    # domain: numpy semantics
    sp = _lib.test_math(0, np.array([0.0, -1.0, np.inf, np.nan]))
    assert sp[0] == -np.inf and np.isnan(sp[1]) and sp[2] == np.inf and np.isnan(sp[3])


def test_log_pos_accuracy(gpu):
    rng = np.random.default_rng(4)
    x = np.concatenate([rng.random(300000), 1 - rng.random(100000) * 1e-3, np.exp(rng.uniform(-700, 700, 100000)),
                        np.linspace(0.70, 0.72, 50001), np.linspace(0.999, 1.001, 50001)])
    x = x[x > 0]
    got = _lib.test_math(4, x)
    ref = np.log(np.asarray(x, dtype=np.longdouble)).astype(np.float64)
    ok = ref != 0
    assert ulp_err(got[ok], ref[ok]).max() <= 2.5


def test_log_unit_table_accuracy(gpu):
    """The table-driven log (no division, no selects) of the fused row kernel's pointwise part,
    chs_log_unit_tab_f64: the kernel only ever asks for log U and log(1-U) with U in (0,1)."""
    rng = np.random.default_rng(5)
    grid = np.arange(128, 257) / 256.0
    x = np.concatenate([rng.random(400000), 1 - rng.random(100000) * 1e-3, 1 - rng.random(50000) * 1e-9,
                        np.exp(rng.uniform(-700, 0, 100000)), np.linspace(0.70, 0.72, 50001),
                        np.linspace(0.49, 0.51, 50001), np.linspace(0.99, 1.0, 200001), grid,
                        (grid[:-1] + 0.5 / 256.0), np.nextafter(grid[:-1] + 0.5 / 256.0, 0),
                        2.0 ** -np.arange(1, 1000, 7.0), np.nextafter(2.0 ** -np.arange(1, 1000, 7.0), 0)])
    x = x[(x > 0) & (x <= 1)]
    got = _lib.test_math(5, x)
    ref = np.log(np.asarray(x, dtype=np.longdouble)).astype(np.float64)
    ok = ref != 0
    e = ulp_err(got[ok], ref[ok])
    assert e.max() <= 2.5, (e.max(), x[ok][np.argmax(e)])
    assert np.all(got[x == 1.0] == 0.0)
    # x > 1 (never asked for by the timestep): e ln2 and the table term cancel, the absolute error stays small
    y = np.concatenate([1 + rng.random(100000), np.exp(rng.uniform(0, 700, 100000))])
    goty = _lib.test_math(5, y)
    refy = np.log(np.asarray(y, dtype=np.longdouble)).astype(np.float64)
    assert np.max(np.abs(goty - refy) / np.maximum(1.0, np.abs(refy))) < 4e-16
    # domain: the table index flags x <= 0 (the row kernel poisons its sums then); NaN and inf propagate
    sp = _lib.test_math(5, np.array([0.0, -0.0, -1.0, -1e-300, -np.inf, np.nan, np.inf]))
    assert np.all(np.isnan(sp[:6])) and not np.isfinite(sp[6])
    assert np.all(np.isfinite(_lib.test_math(5, np.array([5e-324, 1e-310, 2.2250738585072014e-308]))))  # denormals are fine


def test_log_ratio_accuracy(gpu):
    rng = np.random.default_rng(1)
    U = np.concatenate([rng.uniform(0.5, 1 - 1e-9, 300000), rng.uniform(1e-9, 0.5, 300000),
                        rng.uniform(0.86, 0.89, 100000), rng.uniform(0.499, 0.501, 100000)])
    Uinv = 1 - U
    got = _lib.test_math(1, U, Uinv)
    # log(U/Uinv) = log1p((U-Uinv)/Uinv): the difference is exact in extended precision, so the
    # reference stays accurate where the quotient is close to 1
    Ul, Il = np.asarray(U, dtype=np.longdouble), np.asarray(Uinv, dtype=np.longdouble)
    near1 = np.abs(np.asarray(U) - 0.5) < 0.1
    ref = np.where(near1, np.log1p((Ul - Il) / Il), np.log(Ul / Il)).astype(np.float64)
    ok = np.abs(ref) > 1e-300
    e = ulp_err(got[ok], ref[ok])
    assert e.max() <= 2.5, e.max()
    # numpy's own log(U/Uinv) (what the reference evaluates) agrees to a few ulp
    # (its own error is ~1 ulp of max(1, |log|) because the quotient is rounded first)
    npv = np.log(U / Uinv)
    assert np.all(np.abs(got - npv) <= 4 * np.spacing(np.maximum(np.abs(npv), 1.0)))
    sp = _lib.test_math(1, np.array([1.5, 0.0, 1.0, np.nan]), np.array([-0.5, 1.0, 0.0, 1.0]))
    assert np.all(~np.isfinite(sp))


def test_mu_and_energy_density(gpu):
    o = orc.OracleSolver(orc.make_params(64, 2))
    rng = np.random.default_rng(2)
    U = rng.uniform(0.6, 0.99, 500000)
    mu = _lib.test_math(2, U, np.array([o.RT, o.BRT, o.A0, o.A1]))
    ref = o.mu(U)
    assert np.max(np.abs(mu - ref)) < 4e-14  # |mu| ~ 100, absolute error of a few ulp of RT*log
    e = _lib.test_math(3, U, np.array([o.RT, o.params.B, o.A0, o.A1]))
    Uinv = 1 - U
    eref = o.RT * (U * (np.log(U) - o.params.B) + Uinv * np.log(Uinv)) + (o.A0 + o.A1 * (Uinv - U)) * U * Uinv
    assert np.allclose(e, eref, rtol=1e-14, atol=0)
