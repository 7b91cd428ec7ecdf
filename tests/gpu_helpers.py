"""Shared helpers of the ``-m gpu`` parity tests: parameter construction and the run-against-the-oracle
comparison (timedata columns, final U, counters, stop reason)."""
import os

import numpy as np
import pytest

import chsimpy_amd
from oracle import chs_oracle as orc

KAPPA = 0.0002989112919661156
RTOL = 1e-9
GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def make(N, ntmax, engine='auto', **kw):
    p = chsimpy_amd.Parameters()
    p.N, p.ntmax, p.full_sim, p.kappa_tilde, p.engine = N, ntmax, True, KAPPA, engine
    for k, v in kw.items():
        setattr(p, k, v)
    if 'threshold' not in kw:
        p.threshold = p.XXX
    return p


def relerr(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


def log_line(text):
    """Measured parity margins go to gpurun_out/parity.log on the GPU box (copied to profiles/ at round end)."""
    d = os.path.join(os.path.dirname(os.path.dirname(__file__)), 'gpurun_out')
    if os.path.isdir(d):
        with open(os.path.join(d, 'parity.log'), 'a') as f:
            f.write(text.rstrip() + '\n')


def _log_parity(p, engine, eu, cols):
    d = os.path.join(os.path.dirname(os.path.dirname(__file__)), 'gpurun_out')
    if os.path.isdir(d):
        with open(os.path.join(d, 'parity.log'), 'a') as f:
            f.write(f"N={p.N} ntmax={p.ntmax} engine={engine} adaptive={p.adaptive_time} jitter={p.jitter} "
                    f"delt={p.delt}: max rel err U={eu:.3e} E={cols[0]:.3e} E2={cols[1]:.3e} Ra={cols[2]:.3e} "
                    f"L2={cols[3]:.3e} PS={cols[4]:.3e}\n")


def compare_run(p, okw, U_init=None, rtol=RTOL, cols=(1, 2, 3, 4, 5, 6, 7, 8)):
    s = chsimpy_amd.Solver(p, U_init)
    s.prepare()
    sol = s.solve_or_resume()
    o = orc.OracleSolver(orc.make_params(p.N, p.ntmax, **okw), U_init)
    o.prepare()
    o.solve_or_resume()
    td, to = sol.timedata.data(), o.timedata.data()
    assert td.shape == to.shape
    assert np.array_equal(td[:, 0], to[:, 0])
    for c in cols:
        assert np.allclose(td[:, c], to[:, c], rtol=rtol, atol=1e-300), (c, relerr(td[:, c], to[:, c]))
    _log_parity(p, s._engine.engine, relerr(sol.U, o.U), [relerr(td[:, c], to[:, c]) for c in (1, 2, 5, 6, 7)])
    assert np.allclose(sol.U, o.U, rtol=rtol, atol=0), relerr(sol.U, o.U)
    assert sol.computed_steps == o.computed_steps
    assert sol.stop_reason == o.stop_reason
    assert sol.tau0 == o.tau0 and sol.t0 == pytest.approx(o.t0, rel=1e-12)
    assert s.time_passed == pytest.approx(o.time_passed, rel=1e-12)
    s.close()
    return sol, o


