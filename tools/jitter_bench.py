import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import chsimpy_amd
for dev in (True, False):
    p = chsimpy_amd.Parameters()
    p.N, p.ntmax, p.full_sim, p.kappa_tilde, p.jitter = 4096, 10**9, True, 0.0002989112919661156, 0.001
    s = chsimpy_amd.Solver(p); s.device_rng = dev; s.prepare(); s.solve_or_resume(6)
    n = 40 if dev else 10
    t0 = time.perf_counter(); s.solve_or_resume(n); dt = time.perf_counter() - t0
    print(f"jitter N=4096 noise on {'device' if dev else 'host'}: {dt/n*1e3:.3f} ms/step ({n/dt:.0f} steps/s)")
    s.close()
