import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, chsimpy_amd
for N in (2048, 4096):
    p = chsimpy_amd.Parameters(); p.N, p.ntmax, p.full_sim, p.kappa_tilde = N, 10, True, 0.0002989112919661156
    s = chsimpy_amd.Solver(p); s.prepare(); eng = s._engine
    V = np.random.default_rng(1).random((N, N)) * 0.01 + 0.87
    for rep in range(4):
        t0 = time.perf_counter(); eng.set_U(V); t1 = time.perf_counter(); W = eng.get_U(); t2 = time.perf_counter()
        assert np.array_equal(V, W)
        print(f"N={N}: set_U {1e3*(t1-t0):.2f} ms  get_U {1e3*(t2-t1):.2f} ms ({N*N*8/1e6:.0f} MB)", flush=True)
    s.close(fetch_U=False)
