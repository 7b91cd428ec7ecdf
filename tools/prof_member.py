import os, sys, time, cProfile, pstats
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import chsimpy_amd
from chsimpy_amd import experiment as ex
p = chsimpy_amd.Parameters()
p.N, p.ntmax, p.full_sim, p.kappa_tilde, p.file_id = 2048, 400, True, 0.0002989112919661156, '/tmp/ens'
ep = ex.ExperimentParams(); ep.runs = 4
rv, al, n = ex.make_rand_values(ep)
U0 = None   # every member draws the (shared) start field on the device
ex.run_experiment_gpu(0, p, rv, al, U0, postprocess=False)   # warm
pr = cProfile.Profile(); pr.enable()
t0=time.time()
for i in range(1,4): ex.run_experiment_gpu(i, p, rv, al, U0, postprocess=False)
print('per member', (time.time()-t0)/3)
pr.disable(); pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
