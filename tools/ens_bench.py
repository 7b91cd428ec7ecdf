import sys, time, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import chsimpy_amd
from chsimpy_amd import experiment as ex
concs = [int(x) for x in sys.argv[1:]] or [1, 2, 3, 4]
for conc in concs:
    p = chsimpy_amd.Parameters()
    p.N, p.ntmax, p.full_sim, p.kappa_tilde, p.file_id = 2048, 400, True, 0.0002989112919661156, '/tmp/ens'
    ep = ex.ExperimentParams(); ep.runs = 8
    t0 = time.time()
    recs = ex.run_ensemble(p, ep, run_fn=lambda i, pp, rv, al: ex.run_experiment_gpu(i, pp, rv, al, None, postprocess=False), concurrent=conc)
    dt = time.time() - t0
    print(f"concurrent={conc}: 8 runs x 399 steps at N=2048 in {dt:.2f} s -> {8*399/dt:.0f} timesteps/s aggregate")
