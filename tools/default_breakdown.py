import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
import chsimpy_amd
for rep in range(3):
    p = chsimpy_amd.Parameters(); p.N, p.kappa_tilde, p.no_gui = 512, 0.0002989112919661156, True
    t0 = time.perf_counter(); sim = chsimpy_amd.Simulator(p); t1 = time.perf_counter()
    sim.solver.prepare(); t2 = time.perf_counter()
    sim.steps_total = 1  # (prepare done)
    sol = sim.solver.solve_or_resume(p.ntmax); t3 = time.perf_counter()
    print(f"rep {rep}: Simulator() {1e3*(t1-t0):.1f} ms, prepare {1e3*(t2-t1):.1f} ms, solve_or_resume {1e3*(t3-t2):.1f} ms, steps {sol.computed_steps}", flush=True)
    sim.solver.close()
