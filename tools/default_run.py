"""The reference's default workflow end to end: Parameters() as chsimpy ships them (N=512, ntmax=1e6,
full_sim=False -> the run ends at the E2 maximum), Simulator.solve().  Prints the stop step and the wall time;
also at N=4096."""
import os
import sys
import time

sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import chsimpy_amd

for N in (512, 4096):
    p = chsimpy_amd.Parameters()
    p.N, p.kappa_tilde, p.no_gui = N, 0.0002989112919661156, True
    sim = chsimpy_amd.Simulator(p)
    t0 = time.perf_counter()
    sol = sim.solve()
    dt = time.perf_counter() - t0
    td = sol.timedata.data()
    print(f"N={N}: stop_reason={sol.stop_reason} computed_steps={sol.computed_steps} tau0={sol.tau0} t0={sol.t0:.6e} "
          f"rows={td.shape[0]} E2max={td[:, 2].max():.6e} at {int(np.argmax(td[:, 2]))}  wall {dt:.2f} s "
          f"({(sol.computed_steps - 1) / dt:.0f} steps/s)  U in [{sol.U.min():.4f}, {sol.U.max():.4f}]", flush=True)
    if N == 512:
        np.savez_compressed(os.path.join(os.path.dirname(__file__), '..', 'gpurun_out', 'default_n512.npz'), td=td, U=sol.U)
    sim.solver.close()
