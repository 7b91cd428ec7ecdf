#!/usr/bin/env python3
"""bench.py -- timesteps/s of the Cahn-Hilliard solver loop on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one iteration of chsimpy/solver.py:165-249 (pointwise chemical
potential -> 2-D DCT-II -> spectral update -> inverse DCT -> energy/statistics
record) on an N x N fp64 grid resident in HBM.  Workload at --gpus 1: BASELINE.json
configs[2] (N=4096, fp64, synthetic U_init of solver.py:78-82 with seed 2023).
With --gpus G > 1 the script starts G ranks itself (or runs as one of the ranks a
launcher such as torch.distributed.run started); every rank (one process per GPU,
RCCL) advances its own independent member of a Monte-Carlo ensemble (chsimpy/experiment.py:84-126, one run
per GPU, A0/A1 scaled per rank) -- weak scaling, no data-path collective; the only
collective is the all_gather of the per-run energy scalars at the end.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

KAPPA = 0.0002989112919661156
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
PREWARM = 300  # untimed steps in front of the timed region, at least


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    # defaults: the GPU needs ~70 ms of sustained work before step times settle (a 200-step call
    # right behind a 20-step warmup measures ~5 % slower than the following ones)
    ap.add_argument('--steps', type=int, default=None, help='timed steps (default: 5000 = ntmax of BASELINE.json configs[2]; '
                    '1000 with --energy-stop, whose run ends at the E2 maximum)')
    ap.add_argument('--warmup', type=int, default=300)
    ap.add_argument('--grid', type=int, default=4096, help='N (default: BASELINE.json configs[2])')
    ap.add_argument('--dtype', default='float64')
    ap.add_argument('--engine', default='auto')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-steps', type=int, default=0, help='0 = size the CPU sample automatically')
    ap.add_argument('--profile-steps', type=int, default=20)
    ap.add_argument('--energy-stop', action='store_true', help='full_sim=False (the reference default: stop at the E2 maximum); not the headline workload')
    ap.add_argument('--rederive-hat', action='store_true', help='recompute hat_U = dctn(U) on entry of the timed call (the literal '
                    'solver.py:159) instead of continuing the device loop of the warm-up call')
    ap.add_argument('--dry-run', action='store_true', help='launcher/collective rehearsal without device work (CPU tests of the '
                    'N>1 path); the line it prints is marked as such and is not a measurement')
    a = ap.parse_args()
    if a.steps is None:
        a.steps = 1000 if a.energy_stop else 5000
    return a


def make_params(N, dtype, engine, device, rank, full_sim=True):
    import chsimpy_amd
    from chsimpy_amd import utils
    p = chsimpy_amd.Parameters()
    p.N, p.ntmax, p.full_sim, p.kappa_tilde = N, 10 ** 9, full_sim, KAPPA
    p.dtype, p.engine, p.device = dtype, engine, device
    if rank > 0:
        # ensemble member: experiment.py:92-96 scales A0/A1 by factors from PCG64(A_seed)
        fac = np.random.Generator(np.random.PCG64(85972)).uniform(0.995, 1.005, size=(64, 2))[rank % 64]
        p.func_A0 = lambda temp, f=fac[0]: utils.A0(temp) * f
        p.func_A1 = lambda temp, f=fac[1]: utils.A1(temp) * f
    return p


def algorithmic_bytes_per_step(N, esz):
    """8 full-array transfers per timestep."""
    return 8 * N * N * esz


# Algorithmic transfers attributed to each per-step kernel slot.
SLOT_TRANSFERS = {
    'fast': {'k_row_fwd (prologue)': 2, 'k_col': 4, 'k_row_inv (fused)': 4},
    'direct': {},
}


def cpu_baseline(N, steps_hint):
    """The oracle (numpy/scipy restatement, 1 core like chsimpy/simulator.py:14,36) on a
    bounded sample of the same workload, timed like examples/benchmark.py:68-76."""
    from threadpoolctl import threadpool_limits
    from oracle import chs_oracle as orc
    with threadpool_limits(limits=1, user_api='blas'):
        # size the sample: ~0.09 us per grid point per step (measured on the GPU box host) on one core
        est = 0.09e-6 * N * N
        steps = steps_hint or int(max(2, min(200, 20.0 / est)))
        p = orc.make_params(N, steps + 1)
        o = orc.OracleSolver(p)
        o.prepare()
        t0 = time.time()
        o.solve_or_resume()
        dt = time.time() - t0
    return {'value': steps / dt, 'unit': 'timesteps/s', 'cores': 1, 'kind': 'port',
            'sample': f'oracle/chs_oracle.py (numpy+scipy.fftpack), N={N} fp64, {steps} timesteps after prepare(), '
                      f'{dt:.1f} s wall, BLAS limited to 1 thread; host has {os.cpu_count()} logical cores'}


def dry_run(a, rank, world, dist, coll_dev):
    """The N>1 protocol without device work: barrier, timed region (a sleep stands in for the steps),
    MAX over ranks, the all_gather of the per-run scalars, one JSON line from rank 0."""
    import torch
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.001 * a.steps)
    dt = time.perf_counter() - t0
    energies = [[float(rank), float(world)]]
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        mine = torch.tensor(energies[0], dtype=torch.float64, device=coll_dev)
        allv = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine)
        energies = [[float(v[0]), float(v[1])] for v in allv]
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({'metric': metric_name(a.grid, a.dtype), 'value': None, 'unit': 'timesteps/s', 'n_gpus': world,
                          'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': None, 'higher_is_better': True,
                          'scaling': 'weak', 'vs_baseline': None, 'data': 'dry-run (launcher rehearsal, no device work: not a measurement)',
                          'energies_last_step': energies}), flush=True)


def metric_name(N, dtype):
    t = 'fp64' if dtype in ('float64', 'f64') else 'fp32'
    return f'timesteps/sec at N={N} {t}; achieved HBM GB/s vs MI355X peak'


def launch_ranks(a):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks ourselves, one process
    per GPU, like the reference's ensemble starts its own worker pool (chsimpy/experiment.py:197-216).
    This parent has not touched the GPU (no torch.cuda, no engine): the ranks are ordinary child
    processes (never an exec of a process that has initialised the device).  Rank 0's JSON line is
    forwarded; any failing rank fails the run."""
    import socket
    import subprocess
    import __graft_entry__ as g
    if not a.dry_run:
        g.build_hip()  # once, before the ranks start (hipcc only; no device needed)
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    line = [ln for ln in (out0 or '').splitlines() if ln.startswith('{')]
    if any(codes) or not line:
        sys.stderr.write(f'bench.py: ranks exited with {codes}\n')
        if out0:
            sys.stderr.write(out0)
        sys.exit(1)
    print(line[-1], flush=True)


def main():
    a = parse()
    if a.gpus > 1 and 'RANK' not in os.environ:
        return launch_ranks(a)
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != a.gpus and world > 1:
        a.gpus = world
    dist = None
    import torch
    # one process per GPU over RCCL (backend "nccl"); CHS_DIST_BACKEND=gloo + CHS_BENCH_SAME_GPU=1 lets the
    # N>1 logic be rehearsed with several ranks on a single-GPU box (collectives on CPU tensors)
    backend = os.environ.get('CHS_DIST_BACKEND', 'nccl')
    same_gpu = os.environ.get('CHS_BENCH_SAME_GPU') == '1'
    device = 0 if (world == 1 or same_gpu) else local_rank
    coll_dev = 'cpu' if backend == 'gloo' else f'cuda:{device}'
    if world > 1:
        import torch.distributed as dist
        if backend == 'nccl':
            torch.cuda.set_device(device)
            dist.init_process_group(backend='nccl', device_id=torch.device('cuda', device))
        else:
            dist.init_process_group(backend=backend)
    if a.dry_run:
        return dry_run(a, rank, world, dist, coll_dev)

    import __graft_entry__ as g
    if rank == 0:
        g.build_hip()
    if dist is not None:
        dist.barrier()
    import chsimpy_amd

    N = a.grid
    p = make_params(N, a.dtype, a.engine, device, rank, not a.energy_stop)
    s = chsimpy_amd.Solver(p)
    s.prepare()
    eng = s._engine
    esz = 8 if a.dtype in ('float64', 'f64') else 4

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # torch's lazy device initialisation takes milliseconds: have it behind us before anything is timed, so that
    # the synchronisation in front of the timed region is the microseconds it should be (an idle gap of ~5 ms
    # there sends the board's power management through a boost-then-throttle transient that lasts longer than
    # the driver's 20 timed steps; tools/timeline.py on a kernel trace shows it)
    sync()

    # warmup (untimed): W steps through the same C-ABI call as the timed region, so that nothing
    # (the 128 MB download of U at the end of Solver.solve_or_resume, tens of ms of an idle GPU)
    # sits between the warmup and the timed steps and lets the clocks drop.
    # The GPU needs ~70 ms of sustained work before step times settle: when fewer than PREWARM warmup
    # steps are requested, the untimed phase is topped up in front of them (reported in config).
    prewarm = max(0, PREWARM - a.warmup)
    if prewarm:
        eng.step_n(prewarm)
    rows_w, rc_w = eng.step_n(a.warmup)
    nocheck = os.environ.get('CHS_BENCH_NOCHECK') == '1'  # timing experiments with deliberately wrong kernels (tools/ab.sh)
    assert nocheck or (rc_w == 0 and rows_w.shape[0] == a.warmup)
    # keep the input resident: nothing is uploaded inside the timed region.  The timed call is the last call of
    # the run and continues the device loop of the warm-up call, as Solver.solve_or_resume does between the
    # chunks of a run: hat_U stays on the device instead of being recomputed as dctn(idctn(hat_U)) at the call
    # boundary (solver.py:159; --rederive-hat times the literal recomputation).
    sync()
    t0 = time.perf_counter()
    rows, rc = eng.step_n(a.steps, rederive_hat=a.rederive_hat, last_call=True)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    assert nocheck or (rows.shape[0] == a.steps and rc == 0), (rows.shape, rc)
    dt = t1 - t0
    dev_ms = eng.last_step_ms()
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # the ensemble's only collective: gather the per-run energy scalars (E, E2 of the last step)
        mine = torch.tensor([float(rows[-1, 1]), float(rows[-1, 2])], dtype=torch.float64, device=coll_dev)
        allv = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine)
        energies = [[float(v[0]), float(v[1])] for v in allv]
    else:
        energies = [[float(rows[-1, 1]), float(rows[-1, 2])]]

    out = None
    if rank == 0:
        value = world * a.steps / dt
        ms_per_step = dt * 1e3 / a.steps
        bytes_step = algorithmic_bytes_per_step(N, esz)
        # dominant kernel, HIP events on the engine's stream
        ms, calls = eng.profile_steps(a.profile_steps)
        names = eng.kernel_names()
        per = {names[i]: ms[i] / calls[i] for i in range(len(names)) if names[i] and calls[i] > 0}
        tot = {names[i]: ms[i] for i in range(len(names)) if names[i] and calls[i] > 0}
        dom = max(tot, key=tot.get)  # dominant = most device time over the profiled steps (not the once-per-call prologue)
        transfers = SLOT_TRANSFERS.get(eng.engine, {}).get(dom)
        if transfers is None:
            # kernels outside the 8-transfer model: price them with the whole-step figure
            dom_bytes = None
            achieved = bytes_step / (sum(per.values()) * 1e-3) / 1e9
        else:
            dom_bytes = transfers * N * N * esz
            achieved = dom_bytes / (per[dom] * 1e-3) / 1e9
        traffic = None
        tf = os.path.join(ROOT, 'profiles', 'traffic.json')
        if os.path.exists(tf):
            try:
                traffic = json.load(open(tf)).get(f'{eng.engine}:{dom}:N{N}')
            except Exception:
                traffic = None
        roofline = {'bound': 'hbm', 'kernel': dom, 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS,
                    'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': traffic,
                    'algorithmic_bytes_per_launch': dom_bytes,
                    'avg_launch_ms': round(per[dom], 5),
                    'kernel_ms': {k: round(v, 5) for k, v in per.items()},
                    'whole_step': {'algorithmic_bytes': bytes_step,
                                   'achieved': round(bytes_step / (ms_per_step * 1e-3) / 1e9, 1),
                                   'frac': round(bytes_step / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}
        out = {
            'metric': metric_name(N, a.dtype),
            'value': round(value, 3), 'unit': 'timesteps/s', 'n_gpus': world, 'steps': a.steps,
            'warmup': a.warmup, 'ms_per_step': round(ms_per_step, 5), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f64' if esz == 8 else 'f32', 'data': 'synthetic',
            'config': {'workload': f'N={N} {"fp64" if esz == 8 else "fp32"} Cahn-Hilliard timestep loop '
                                   f'(BASELINE.json configs[2] at N=4096), U_init = 0.875 + 0.00875*(PCG64(2023).random - 0.5), '
                                   f'kappa_tilde={KAPPA}, ' + ('energy stop armed (full_sim=False)' if a.energy_stop else 'full_sim'),
                       'N': N, 'engine': eng.engine,
                       'ensemble': f'{world} independent run(s), one per GPU' if world > 1 else 'single run',
                       'device_ms_per_step': round(dev_ms / a.steps, 5),
                       'untimed_steps_before_timed_region': max(a.warmup, PREWARM),
                       'call_entry': 'hat_U = dctn(U) recomputed on entry (solver.py:159)' if a.rederive_hat else
                                     'continues the device loop of the warm-up call (hat_U resident)'},
            'roofline': roofline,
            'energies_last_step': energies,
        }
        if not a.no_cpu_baseline and world == 1:
            out['cpu_baseline'] = cpu_baseline(N, a.cpu_steps)
        elif world == 1:
            out['cpu_baseline'] = None
    s.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
