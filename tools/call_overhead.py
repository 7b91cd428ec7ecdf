"""Per-call cost of the device loop: chs_step_n of various lengths (wall vs device time), a call that has to enter
through hat_U = dctn(U) against one that continues the previous call's loop, and Solver.solve_or_resume in chunks."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import chsimpy_amd
for N in (4096, 512):
    p = chsimpy_amd.Parameters()
    p.N, p.ntmax, p.full_sim, p.kappa_tilde = N, 10 ** 9, True, 0.0002989112919661156
    s = chsimpy_amd.Solver(p); s.prepare(); s.solve_or_resume(21)
    eng = s._engine
    eng.step_n(300)
    for n in (200, 200, 1, 1, 2, 10, 20, 100, 1000, 200):
        for rederive in (False, True):
            t0 = time.perf_counter(); rows, rc = eng.step_n(n, rederive_hat=rederive); dt = time.perf_counter() - t0
            print(f"N={N} step_n({n:5d}{', rederive_hat' if rederive else '':14s}): wall {dt*1e3:9.3f} ms  device {eng.last_step_ms():9.3f} ms  "
                  f"per step {dt*1e3/n:8.4f}")
    for chunk in (1, 10, 100):
        k = 20
        t0 = time.perf_counter()
        for _ in range(k):
            s.solve_or_resume(chunk)
        dt = (time.perf_counter() - t0) / k
        print(f"N={N} Solver.solve_or_resume({chunk:4d}): {dt*1e3:8.3f} ms per call, {dt*1e3/chunk:8.4f} per step")
    s.close()
