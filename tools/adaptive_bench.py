"""Timesteps/s of the adaptive-time path (BASELINE.json configs[3]: N=8192 fp32, adaptive_time on).
The adaptive branch is live beyond step 500 (solver.py:177), so the timed region starts at step 520."""
import os
import sys
import time

sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import chsimpy_amd

for N, dtype, dmax in ((8192, 'float32', 6e-11), (4096, 'float64', 1.2e-10)):
    p = chsimpy_amd.Parameters()
    p.N, p.ntmax, p.full_sim, p.kappa_tilde = N, 10 ** 9, True, 0.0002989112919661156
    p.dtype, p.adaptive_time, p.delt_max = dtype, True, dmax
    s = chsimpy_amd.Solver(p)
    s.prepare()
    s.solve_or_resume(520)
    eng = s._engine
    t0 = time.perf_counter()
    rows, rc = eng.step_n(200)
    dt = time.perf_counter() - t0
    assert rc == 0 and rows.shape[0] == 200
    print(f"N={N} {dtype} adaptive: {200 / dt:.1f} timesteps/s, {dt / 200 * 1e3:.4f} ms/step "
          f"(device {eng.last_step_ms() / 200:.4f} ms), delt {rows[0, 8]:.3e} -> {rows[-1, 8]:.3e}")
    s.close()
