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
    ap.add_argument('--no-extras', action='store_true', help='skip the two extra figures a default --gpus 1 line carries under '
                    'config (full_call_5000_ms_per_step, cfg3_n8192_f32_adaptive_steps_per_s): A/B scripts')
    ap.add_argument('--energy-stop', action='store_true', help='full_sim=False (the reference default: stop at the E2 maximum); not the headline workload')
    ap.add_argument('--continue-loop', action='store_true', help='the timed call continues the device loop of the warm-up call '
                    '(hat_U carried, last call of the run) instead of entering through hat_U = dctn(U) as every '
                    'solve_or_resume call of the reference does (solver.py:159); not the default protocol')
    ap.add_argument('--ensemble-baseline', action='store_true', help='configs[4] beside its CPU comparator: P oracle processes '
                    '(one run each, experiment.py:197-216) against the GPU ensemble at N=2048; prints one JSON line')
    ap.add_argument('--ens-procs', type=int, default=0, help='CPU processes of --ensemble-baseline (0 = the cores this process may use, at most 16)')
    ap.add_argument('--ens-steps', type=int, default=0, help='timesteps per CPU member and repetition (0 = sized to ~5 s)')
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


def time_repetitions(run_once, warmups=1, repetitions=3):
    """The timing protocol of examples/benchmark.py:25-27,68-76: `warmups` untimed runs, then `repetitions`
    timed ones (each = prepare(), then the wall clock around the solve); returns the list of times."""
    for _ in range(warmups):
        run_once()
    return [run_once() for _ in range(repetitions)]


def _oracle_member(args):
    """One CPU run of the oracle: prepare(), then the wall clock around solve_or_resume (a worker process of the
    ensemble comparator, or the single-core baseline)."""
    N, steps, fac = args
    from threadpoolctl import threadpool_limits
    from oracle import chs_oracle as orc
    with threadpool_limits(limits=1, user_api='blas'):
        kw = {}
        if fac is not None:
            kw = dict(func_A0=lambda T, f=fac[0]: orc.A0(T) * f, func_A1=lambda T, f=fac[1]: orc.A1(T) * f)
        o = orc.OracleSolver(orc.make_params(N, steps + 1, **kw))
        o.prepare()
        t0 = time.time()
        o.solve_or_resume()
        return time.time() - t0


def cpu_baseline(N, steps_hint):
    """The oracle (numpy/scipy restatement, 1 core like chsimpy/simulator.py:14,36) on a bounded sample of the
    same workload, timed like examples/benchmark.py:68-76: 1 warm-up + 3 repetitions of prepare() + solve."""
    # size the sample: ~0.09 us per grid point per step on one core (measured on the GPU box host); four runs in ~20 s
    est = 0.09e-6 * N * N
    steps = steps_hint or int(max(2, min(200, 5.0 / est)))
    times = time_repetitions(lambda: _oracle_member((N, steps, None)))
    mean = sum(times) / len(times)
    return {'value': steps / mean, 'unit': 'timesteps/s', 'cores': 1, 'kind': 'port',
            'best': steps / min(times), 'repetitions': len(times), 'warmups': 1,
            'sample': f'oracle/chs_oracle.py (numpy+scipy.fftpack), N={N} fp64, {steps} timesteps after prepare() per run, '
                      f'1 warm-up + {len(times)} timed runs (mean {mean:.1f} s, min {min(times):.1f} s), BLAS limited to 1 thread; '
                      f'host has {os.cpu_count()} logical cores'}


def extra_figures(a, eng, torch, device):
    """Two claims of DESIGN.md section 7 measured inside the driver's own run, behind the
    timed region and the profile leg, each a few hundred ms:
    * full_call_5000_ms_per_step -- BASELINE.json configs[2] as ONE literal solve_or_resume call of 5000 steps (ntmax),
      hat_U = dctn(U) on entry, U stored at the end, on the engine the headline was timed on;
    * cfg3_n8192_f32_adaptive_steps_per_s -- BASELINE.json configs[3]: N=8192, fp32, adaptive_time on; the adaptive
      branch is live beyond step 500 (solver.py:177), so 520 untimed steps, then one 600-step call."""
    import chsimpy_amd
    out = {}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rows, rc = eng.step_n(5000, rederive_hat=True, last_call=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rc == 0 and rows.shape[0] == 5000:
        out['full_call_5000_ms_per_step'] = round(dt * 1e3 / 5000, 5)
        out['full_call_5000_steps_per_s'] = round(5000 / dt, 1)
    p = chsimpy_amd.Parameters()
    p.N, p.ntmax, p.full_sim, p.kappa_tilde = 8192, 10 ** 9, True, KAPPA
    p.dtype, p.engine, p.device, p.adaptive_time, p.delt_max = 'float32', 'fast', device, True, 6e-11
    s3 = chsimpy_amd.Solver(p)
    try:
        s3.prepare()
        e3 = s3._engine
        r0, rc0 = e3.step_n(519)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r1, rc1 = e3.step_n(600)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        # (the adaptive branch runs on every second step of the window; by step 520 the step size has left params.delt)
        if rc0 == 0 and rc1 == 0 and r1.shape[0] == 600 and r1[-1, 8] > 1.5 * p.delt:
            out['cfg3_n8192_f32_adaptive_steps_per_s'] = round(600 / dt, 1)
            out['cfg3_n8192_f32_adaptive_ms_per_step'] = round(dt * 1e3 / 600, 5)
            out['cfg3_delt_first_last'] = [float(r1[0, 8]), float(r1[-1, 8])]
        else:
            out['cfg3_n8192_f32_adaptive_steps_per_s'] = None
            out['cfg3_note'] = f'not measured: rc {rc0}/{rc1}, rows {r1.shape[0]}, delt {float(r1[0, 8]) if r1.shape[0] else None}'
    finally:
        s3.close(fetch_U=False)
    return out


def ensemble_baseline(a):
    """BASELINE.json configs[4] beside its CPU comparator: the reference runs one member per
    physical core in a process pool (experiment.py:197-216); here P processes of the oracle, one N=2048 member
    each with its own (A0, A1) factors, 1 warm-up + 3 repetitions (examples/benchmark.py:68-76), against the GPU
    ensemble on one MI355X (members run `concurrent` at a time, chsimpy_amd/experiment.py)."""
    import multiprocessing as mp
    N = 2048
    try:
        allowed = len(os.sched_getaffinity(0))
    except AttributeError:
        allowed = os.cpu_count() or 1
    P = a.ens_procs or max(1, min(allowed, 16))
    steps = a.ens_steps or 6
    fac = np.random.Generator(np.random.PCG64(85972)).uniform(0.995, 1.005, size=(64, 2))
    ctx = mp.get_context('fork')   # as the reference's pool (experiment.py:16,211); nothing here has touched the GPU yet
    solve_s = []
    with ctx.Pool(P) as pool:
        def run_once():
            t0 = time.time()
            solve_s.append(pool.map(_oracle_member, [(N, steps, tuple(fac[i % 64])) for i in range(P)]))
            return time.time() - t0
        times = time_repetitions(run_once)
    mean = sum(times) / len(times)
    # The members' own solve times (the wall clock around solve_or_resume after prepare(), examples/benchmark.py:68-76 --
    # what `cpu_baseline` of the headline times too) give the rate the GPU figure is held against; the wall clock around
    # the whole pool.map (set-up and prepare() included, amortised over only `steps` timesteps) is reported beside it.
    timed = solve_s[1:]                                   # (the first repetition is the warm-up)
    member_mean = sum(sum(r) for r in timed) / (len(timed) * P)
    cpu = {'value': P * steps / member_mean, 'best': P * steps / min(max(r) for r in timed), 'unit': 'timesteps/s (aggregate)',
           'cores': P, 'cores_cap': 16 if not a.ens_procs else None, 'kind': 'port', 'per_member_steps_per_s': steps / member_mean,
           'wall_incl_setup_steps_per_s': P * steps / mean,
           'comparator_note': f'{P} processes (the cores a one-GPU job is meant to use, capped at 16; the reference sizes its pool '
                              f'to all physical cores, experiment.py:197-202, and this host has {os.cpu_count()} logical cores); rate from '
                              f'the members\' own solve times, set-up and prepare() excluded on both sides',
           'sample': f'{P} oracle processes (fork pool, BLAS 1 thread each), one N={N} fp64 member of {steps} timesteps each, '
                     f'1 warm-up + {len(times)} timed repetitions: pool.map wall mean {mean:.2f} s, min {min(times):.2f} s, '
                     f'member solve mean {member_mean:.2f} s; {allowed} cores allowed to this process, host has {os.cpu_count()} logical cores'}
    out = {'metric': 'ensemble timesteps/s at N=2048 fp64 (BASELINE.json configs[4] members)', 'cpu_ensemble': cpu}
    if not a.dry_run:
        import __graft_entry__ as g
        g.build_hip()
        import chsimpy_amd
        from chsimpy_amd import experiment as ex
        gpu = {}
        runs, nt = 8, 400
        for conc in (1, 3):
            def run_once(conc=conc):
                p = chsimpy_amd.Parameters()
                p.N, p.ntmax, p.full_sim, p.kappa_tilde, p.file_id = N, nt, True, KAPPA, '/tmp/chs_ens'
                ep = ex.ExperimentParams()
                ep.runs = runs
                t0 = time.time()
                ex.run_ensemble(p, ep, run_fn=lambda i, pp, rv, al: ex.run_experiment_gpu(i, pp, rv, al, None, postprocess=False),
                                concurrent=conc)
                return time.time() - t0
            t = time_repetitions(run_once)
            gpu[f'concurrent_{conc}'] = {'value': runs * (nt - 1) / (sum(t) / len(t)), 'best': runs * (nt - 1) / min(t),
                                        'unit': 'timesteps/s (aggregate, one MI355X)',
                                        'sample': f'{runs} members x {nt - 1} timesteps end to end (engine from the pool, start field drawn '
                                                  f'on the device, no sympy post-processing), 1 warm-up + {len(t)} repetitions'}
        out['gpu_ensemble'] = gpu
        best = max(v['value'] for v in gpu.values())
        out['gpu_over_cpu'] = best / cpu['value']
        out['comparison_note'] = ('CPU: members\' solve times only (set-up and prepare() excluded), pool capped at cores_cap; GPU: end to end '
                                  'incl. engine set-up, start field and prepare() of every member (amortised over 399 timesteps) -- the '
                                  'ratio errs against the GPU, and a pool of all physical cores would raise the CPU figure accordingly')
        out['roofline_note'] = (f'{best:.0f} steps/s x {algorithmic_bytes_per_step(N, 8) / 1e6:.1f} MB algorithmic = '
                                f'{best * algorithmic_bytes_per_step(N, 8) / 1e12:.2f} TB/s; a member (T + hat_U = 64 MiB) lives in the '
                                f'256 MiB Infinity Cache, so this is on-die bandwidth, not HBM')
    print(json.dumps(out), flush=True)


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
                          'energies_last_step': energies,
                          'host_cores_per_rank': len(os.sched_getaffinity(0)) if world > 1 else None}), flush=True)


def metric_name(N, dtype):
    t = 'fp64' if dtype in ('float64', 'f64') else 'fp32'
    return f'timesteps/sec at N={N} {t}; achieved HBM GB/s vs MI355X peak'


def launch_ranks(a):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks ourselves, one process
    per GPU, like the reference's ensemble starts its own worker pool (chsimpy/experiment.py:197-216).
    This parent has not touched the GPU (no torch.cuda, no engine): the ranks are ordinary child
    processes (never an exec of a process that has initialised the device; chsimpy_amd/launch.py).  Rank 0's JSON
    line is forwarded; any failing rank fails the run."""
    import __graft_entry__ as g
    from chsimpy_amd import launch
    if not a.dry_run:
        g.build_hip()  # once, before the ranks start (hipcc only; no device needed)
    failed, codes, texts = launch.spawn_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], a.gpus,
                                              timeout_s=float(os.environ.get('CHS_BENCH_LAUNCH_TIMEOUT', '1500')))
    line = [ln for ln in texts[0].splitlines() if ln.startswith('{')]
    if failed is not None or not line:
        launch.report_failure('bench.py', failed or 'rank 0 printed no result line', codes, texts)
        sys.exit(1)
    print(line[-1], flush=True)


def main():
    a = parse()
    if a.ensemble_baseline:
        return ensemble_baseline(a)
    if a.gpus > 1 and 'RANK' not in os.environ:
        return launch_ranks(a)
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if os.environ.get('CHS_BENCH_TEST_DIE_RANK') == str(rank) and world > 1:
        sys.exit(3)  # test hook (tests/test_bench_launch.py): a rank that dies before the rendezvous
    if world != a.gpus and world > 1:
        a.gpus = world
    dist = None
    # every rank keeps to its own share of the host cores, decided before anything touches the GPU (chsimpy_amd/launch.py)
    from chsimpy_amd import launch
    pinned = launch.pin_rank_to_cores(local_rank, world)
    import torch
    launch.quiet_host_threads()  # collectives on a few scalars: no intra-op pool (a pool sized for every core of the host, spinning
                                 # inside a CPU quota, throttles the thread that issues the kernel launches -- DESIGN.md section 9)
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
    # keep the input resident: nothing is uploaded inside the timed region.  The timed call is one
    # solve_or_resume call of the reference, literally: it enters through hat_U = dctn(U) (solver.py:159), runs
    # `steps` complete timesteps and leaves U in HBM (solver.py:251).  --continue-loop times the engine's own
    # mode for the chunks of one run instead (hat_U carried across the call boundary, last call of the run);
    # its figure is reported beside the headline either way (config.other_protocol_ms_per_step).
    literal = dict(rederive_hat=True, last_call=False)
    carried = dict(rederive_hat=False, last_call=True)
    sync()
    t0 = time.perf_counter()
    rows, rc = eng.step_n(a.steps, **(carried if a.continue_loop else literal))
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    assert nocheck or (rows.shape[0] == a.steps and rc == 0), (rows.shape, rc)
    dt = t1 - t0
    dev_ms = eng.last_step_ms()
    # the other protocol, right behind the timed region (same clocks), for the record
    other_ms = None
    if world == 1 and not a.energy_stop:   # (with the stop rule armed the run must still be alive for the profile leg below)
        if not a.continue_loop:
            eng.step_n(a.steps)   # a completed call whose loop the continuing call can take up (a literal call leaves none)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        rows_o, rc_o = eng.step_n(a.steps, **(literal if a.continue_loop else carried))
        torch.cuda.synchronize()
        other_ms = (time.perf_counter() - t2) * 1e3 / a.steps
        if rows_o.shape[0] != a.steps or rc_o != 0:
            other_ms = None   # (a stop rule ended the run inside this extra call: --energy-stop)
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
                traffic = json.load(open(tf)).get(f'{eng.engine}:{dom}:N{N}' + (':f32' if esz == 4 else ''))
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
                       'host_cores_per_rank': None if pinned is None else len(pinned),
                       'device_ms_per_step': round(dev_ms / a.steps, 5),
                       'untimed_steps_before_timed_region': max(a.warmup, PREWARM),
                       'call_entry': 'continues the device loop of the warm-up call (hat_U resident, last call of the run)'
                                     if a.continue_loop else
                                     'one literal solve_or_resume call: hat_U = dctn(U) recomputed on entry (solver.py:159), U stored at the end (251)',
                       'other_protocol': 'literal solve_or_resume call' if a.continue_loop else 'continuing the device loop (hat_U carried)',
                       'other_protocol_ms_per_step': None if other_ms is None else round(other_ms, 5)},
            'roofline': roofline,
            'energies_last_step': energies,
        }
        if world == 1 and not a.no_extras and N == 4096 and esz == 8 and not a.energy_stop and eng.engine == 'fast':
            try:
                out['config'].update(extra_figures(a, eng, torch, device))
            except Exception as e:   # (the extra figures must never cost the line itself)
                out['config']['extras_error'] = f'{type(e).__name__}: {e}'[:300]
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
