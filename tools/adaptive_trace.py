"""Kernel-level view of the adaptive-time loop at N=8192 fp32 (configs[3]): run under rocprofv3 --kernel-trace --stats."""
import os
import sys

sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import chsimpy_amd

N, dtype, dmax = int(os.environ.get('CHS_TRACE_N', 8192)), os.environ.get('CHS_TRACE_DTYPE', 'float32'), 6e-11
p = chsimpy_amd.Parameters()
p.N, p.ntmax, p.full_sim, p.kappa_tilde = N, 10 ** 9, True, 0.0002989112919661156
p.dtype, p.adaptive_time, p.delt_max = dtype, True, dmax
s = chsimpy_amd.Solver(p)
s.prepare()
s.solve_or_resume(520)
rows, rc = s._engine.step_n(200)
assert rc == 0
s.close()
