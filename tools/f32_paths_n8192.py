import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tests'))
import numpy as np, chsimpy_amd
from gpu_helpers import make, relerr
N, nt, dmax = 8192, 540, 6e-11
runs = {}
for name, dtype, sweep in (('f64', 'float64', False), ('f32 fused', 'float32', False), ('f32 sweep', 'float32', True)):
    if sweep: os.environ['CHS_ADAPT_SWEEP'] = '1'
    else: os.environ.pop('CHS_ADAPT_SWEEP', None)
    p = make(N, nt, 'fast', adaptive_time=True, delt_max=dmax, dtype=dtype)
    s = chsimpy_amd.Solver(p); s.prepare(); s.solve_or_resume(300); sol = s.solve_or_resume(nt - 300)
    runs[name] = (sol.U.copy(), sol.timedata.data().copy()); s.close()
os.environ.pop('CHS_ADAPT_SWEEP', None)
U0, t0 = runs['f64']
for k in ('f32 fused', 'f32 sweep'):
    U, t = runs[k]
    print(f"{k} vs f64: U {relerr(U, U0):.2e} E {relerr(t[:,1], t0[:,1]):.2e} E2 max {relerr(t[1:,2], t0[1:,2]):.2e} E2 last {abs(t[-1,2]/t0[-1,2]-1):.2e} delt {relerr(t[:,8], t0[:,8]):.2e}")
Uf, tf = runs['f32 fused']; Us, ts = runs['f32 sweep']
print(f"fused vs sweep: U {relerr(Uf, Us):.2e} E2 max {relerr(tf[1:,2], ts[1:,2]):.2e} delt {relerr(tf[:,8], ts[:,8]):.2e}")
