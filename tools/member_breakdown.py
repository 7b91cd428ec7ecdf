"""Where the time of one ensemble member goes (N=2048, 399 steps as in tools/ens_bench.py): engine creation, prepare,
the steps, the field download, engine destruction."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import chsimpy_amd
for rep in range(4):
    p = chsimpy_amd.Parameters(); p.N, p.ntmax, p.full_sim, p.kappa_tilde = 2048, 400, True, 0.0002989112919661156
    t = [time.perf_counter()]
    s = chsimpy_amd.Solver(p); t.append(time.perf_counter())
    s.prepare(); t.append(time.perf_counter())
    sol = s.solve_or_resume(); t.append(time.perf_counter())
    U = sol.U; t.append(time.perf_counter())
    s.close(); t.append(time.perf_counter())
    names = ['Solver()', 'prepare', 'solve_or_resume', 'U download', 'close']
    print(f"rep {rep}: " + ', '.join(f"{n} {1e3*(b-a):.2f} ms" for n, a, b in zip(names, t[:-1], t[1:])), flush=True)
