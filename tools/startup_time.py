"""Time from Parameters to a prepared solver (start field on the device vs drawn on the host and uploaded)."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import chsimpy_amd
for N in (2048, 4096):
    for dev in (True, False, True, False):
        p = chsimpy_amd.Parameters(); p.N, p.ntmax, p.full_sim, p.kappa_tilde = N, 100, True, 0.0002989112919661156
        t0 = time.perf_counter()
        s = chsimpy_amd.Solver(p); s.device_rng = dev; s.prepare()
        dt = time.perf_counter() - t0
        print(f"N={N} start field on {'device' if dev else 'host  '}: Solver() + prepare() = {dt*1e3:7.1f} ms")
        s.close(fetch_U=False)
