"""Per-call overhead of solve_or_resume's device side: chs_step_n of various lengths, wall vs device time."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import chsimpy_amd
p = chsimpy_amd.Parameters()
p.N, p.ntmax, p.full_sim, p.kappa_tilde = 4096, 10 ** 9, True, 0.0002989112919661156
s = chsimpy_amd.Solver(p); s.prepare(); s.solve_or_resume(21)
eng = s._engine
for n in (200, 200, 1, 1, 2, 10, 100, 1000, 200):
    t0 = time.perf_counter(); rows, rc = eng.step_n(n); dt = time.perf_counter() - t0
    print(f"step_n({n:5d}): wall {dt*1e3:9.3f} ms  device {eng.last_step_ms():9.3f} ms  per step {dt*1e3/n:8.4f}")
s.close()
