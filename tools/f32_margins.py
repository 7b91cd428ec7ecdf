import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tests'))
import numpy as np, chsimpy_amd
from oracle import chs_oracle as orc
from gpu_helpers import make, relerr
for N, nt in ((128, 200), (512, 200), (512, 1000)):
    p = make(N, nt, 'fast', dtype='float32')
    s = chsimpy_amd.Solver(p); s.prepare(); sol = s.solve_or_resume()
    o = orc.OracleSolver(orc.make_params(N, nt)); o.prepare(); o.solve_or_resume()
    td, to = sol.timedata.data(), o.timedata.data()
    print(f"N={N} nt={nt}: U {relerr(sol.U, o.U):.3e}  E {relerr(td[:,1], to[:,1]):.3e}  E2 {relerr(td[1:,2], to[1:,2]):.3e}  PS {relerr(td[1:,7], to[1:,7]):.3e}  mean {abs(sol.U.mean()/o.U.mean()-1):.2e}", flush=True)
    s.close()
