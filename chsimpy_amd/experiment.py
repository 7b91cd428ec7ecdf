#!/usr/bin/env python
"""Monte-Carlo ensemble over (A0, A1) factors: the multi-GPU face of the hot path.

Counterpart of ``chsimpy/experiment.py``.  The reference spreads independent runs over a
``multiprocessing.Pool`` (experiment.py:197-216); here the runs are dealt to the GPUs of a
node -- one process per GPU (``torch.distributed``), ``run_id -> rank = run_id mod world`` --
and every run executes the device-resident timestep loop.  A single N x N grid does not
shard, so there is no data-path collective: the only communication is one gather of the
per-run scalar records (12 numbers per run, experiment.py:114-126) at the end, over RCCL
when the ranks sit on GPUs (backend "nccl") and over gloo in the CPU tests.

The per-run factors depend on ``A_seed`` and ``run_id`` only (experiment.py:148-170), so the
results do not depend on how many ranks share the work.

    python -m torch.distributed.run --nproc-per-node 8 -m chsimpy_amd.experiment -R 64 -N 2048 -n 2000
"""
import argparse
import os
import threading

import numpy as np

from . import utils
from .parameters import Parameters

# sympy/mpmath keep their working precision in a process-global context (nsolve(prec=7) switches it
# temporarily): concurrent members of one rank take turns in the thermodynamic post-processing.  (The
# reference runs its members in separate processes, experiment.py:211.)
_SYMPY_LOCK = threading.Lock()

COLS = ['A0', 'A1', 'ca', 'cb', 'sa', 'sb', 'tau0', 't0', 'tsep', 'id', 'fac_A0', 'fac_A1']  # experiment.py:218


class ExperimentParams:
    """experiment.py:22-30"""

    def __init__(self):
        self.runs = 2
        self.jitter_Arellow = 0.995
        self.jitter_Arelhigh = 1.005
        self.processes = -1
        self.independent = False
        self.A_source = 'uniform'
        self.A_seed = 85972


def _factor_columns(ep):
    """Two rows of per-run factors (for A0, for A1) from the seeded source: experiment.py:148-162."""
    lo, hi = ep.jitter_Arellow, ep.jitter_Arelhigh
    if ep.A_source == 'sobol':
        from scipy.stats import qmc
        # the first `runs` points of the 2^ceil(log2 runs) Sobol points, scaled to [lo, hi]
        pts = qmc.Sobol(d=2, seed=ep.A_seed).random_base2(int(np.ceil(np.log2(ep.runs))))
        return qmc.scale(pts, lo, hi)[:ep.runs].T
    return np.random.Generator(np.random.PCG64(ep.A_seed)).uniform(lo, hi, size=(ep.runs, 2)).T


def _lay_out(col_A0, col_A1, independent):
    """Joint layout: run i scales both coefficients; independent: first only A0 varies, then only A1
    (experiment.py:163-170, 176-186)."""
    n = len(col_A0)
    if not independent:
        return np.column_stack([col_A0, col_A1])
    table = np.ones((2 * n, 2))
    table[:n, 0] = col_A0
    table[n:, 1] = col_A1
    return table


def make_rand_values(ep: ExperimentParams):
    """(rand_values, A_list, number of runs): the factor table of experiment.py:148-190 for the four A sources
    (uniform, sobol, grid, or a CSV file of absolute (A0, A1) pairs)."""
    if ep.A_source in ('uniform', 'sobol'):
        f0, f1 = _factor_columns(ep)
        table = _lay_out(f0, f1, ep.independent)
        limit = 2 * ep.runs if ep.independent else ep.runs
        return table, None, min(limit, table.shape[0])
    if ep.A_source == 'grid':
        side = int(np.floor(np.sqrt(ep.runs)))
        ep.runs = side * side                         # (the reference rounds the run count down to a square)
        axis = np.linspace(ep.jitter_Arellow, ep.jitter_Arelhigh, side)
        if ep.independent:
            table = _lay_out(axis, axis, True)
        else:
            g0, g1 = np.meshgrid(axis, axis, indexing='ij')
            table = np.column_stack([g0.ravel(), g1.ravel()])
        return table, None, min(ep.runs, table.shape[0])
    A_list = utils.csv_import_matrix(ep.A_source)
    return None, A_list, min(ep.runs, A_list.shape[0])


def run_params(init_params: Parameters, run_id, rand_values, A_list):
    """Per-run Parameters with scaled A0/A1 (experiment.py:87-101)."""
    params = init_params.deepcopy()
    params.seed = init_params.seed
    params.file_id = f"{init_params.file_id}-run{run_id}"
    if A_list is None:
        fac_A0 = float(rand_values[run_id, 0])
        fac_A1 = float(rand_values[run_id, 1])
        params.func_A0 = lambda temp, f=fac_A0: utils.A0(temp) * f
        params.func_A1 = lambda temp, f=fac_A1: utils.A1(temp) * f
    else:
        a0, a1 = float(A_list[run_id][0]), float(A_list[run_id][1])
        params.func_A0 = lambda temp, a=a0: a
        params.func_A1 = lambda temp, a=a1: a
        fac_A0 = fac_A1 = None
    return params, fac_A0, fac_A1


def run_experiment_gpu(run_id, init_params, rand_values, A_list, U_init=None, postprocess=True):
    """One ensemble member on this rank's GPU; the 12-tuple of experiment.py:114-126."""
    from .simulator import Simulator
    params, fac_A0, fac_A1 = run_params(init_params, run_id, rand_values, A_list)
    simulator = Simulator(params, U_init)
    try:
        solution = simulator.solve()
        simulator.export()
        ca = cb = sa = sb = float('nan')
        if postprocess:
            # experiment.py:110-112 -- a failure here (sympy missing, no common tangent, not exactly two
            # spinodal roots) is an error of the run, as in the reference: a silent NaN would poison
            # -results-agg.csv.  postprocess=False skips the thermodynamic columns explicitly.
            with _SYMPY_LOCK:
                cgap = utils.get_miscibility_gap(params.R, params.temp, params.B, solution.A0, solution.A1)
                ca, cb = float(cgap[0]), float(cgap[1])
                sa, sb = (float(r) for r in utils.get_roots_of_EPP(params.R, params.temp, solution.A0, solution.A1))
        itargmax = int(np.argmax(solution.E2))
    finally:
        # the record needs scalars only; an exception above must not leak the engine either (it goes back to the pool)
        simulator.solver.close(fetch_U=False)
    return (solution.A0, solution.A1, ca, cb, sa, sb, solution.tau0, solution.t0, itargmax, run_id,
            np.nan if fac_A0 is None else fac_A0, np.nan if fac_A1 is None else fac_A1)


def my_run_ids(nr_items, rank, world):
    """run_id -> rank (run_id mod world)."""
    return [i for i in range(nr_items) if i % world == rank]


def gather_records(local, nr_items, rank, world, dist=None, device='cpu'):
    """All per-run records on every rank, ordered by run id.  One all_gather of a
    (ceil(n/world), 12) float64 block per rank -- a few KiB; latency-bound, bandwidth irrelevant."""
    if world == 1 or dist is None:
        return sorted(local, key=lambda r: r[9])
    import torch
    per = (nr_items + world - 1) // world
    buf = torch.full((per, len(COLS)), float('nan'), dtype=torch.float64, device=device)
    for i, rec in enumerate(local):
        buf[i] = torch.tensor([float(x) for x in rec], dtype=torch.float64, device=device)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    recs = []
    for t in out:
        for row in t.cpu().numpy():
            if not np.isnan(row[9]):
                recs.append(tuple(row.tolist()))
    return sorted(recs, key=lambda r: r[9])


def write_metadata(file_id, ep, extra=()):
    """``<file_id>-metadata.csv`` (experiment.py:193-195): system information followed by the
    experiment parameters, one ``name, value`` per line."""
    lines = list(utils.get_system_info()) + list(extra) + utils.vars_to_list(ep)
    utils.csv_export_list(f"{file_id}-metadata.csv", "\n".join(lines))
    return f"{file_id}-metadata.csv"


def write_results(file_id, records):
    """``<file_id>-results.csv`` and ``-results-agg.csv`` exactly as experiment.py:218-225."""
    import pandas as pd
    df = pd.DataFrame(records, columns=COLS)
    df[['tau0', 'id']] = df[['tau0', 'id']].astype(int)
    # tsep = argmax(E2) is an integer in the reference's tuples (experiment.py:113); after the gather of a float64 block
    # it must read the same, or the file would depend on the number of ranks
    df['tsep'] = df['tsep'].astype(int)
    df.to_csv(f"{file_id}-results.csv")
    agg = df.loc[:, df.columns != 'id'].describe()
    agg.loc['cv'] = agg.loc['std'] / agg.loc['mean']
    agg.T.to_csv(f"{file_id}-results-agg.csv")
    return df, agg


def run_ensemble(init_params, ep, run_fn=None, U_init=None, dist=None, rank=0, world=1, device='cpu',
                 concurrent=1):
    """Deal the runs to the ranks, execute, gather.  ``run_fn(run_id, init_params, rand_values,
    A_list)`` defaults to the GPU run; the CPU tests inject a stand-in.

    ``concurrent`` members of a rank run at the same time (one engine handle = one HIP stream each;
    the C ABI calls release the GIL): at ensemble sizes such as N=2048 a single run leaves the GPU
    partly idle between its latency-bound kernels, two or three concurrent runs fill the gaps."""
    rand_values, A_list, nr_items = make_rand_values(ep)
    if run_fn is None:
        def run_fn(run_id, p, rv, al):
            return run_experiment_gpu(run_id, p, rv, al, U_init)
    ids = my_run_ids(nr_items, rank, world)
    if concurrent > 1 and len(ids) > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=concurrent) as pool:
            local = list(pool.map(lambda i: run_fn(i, init_params, rand_values, A_list), ids))
    else:
        local = [run_fn(i, init_params, rand_values, A_list) for i in ids]
    return gather_records(local, nr_items, rank, world, dist, device)


def _dry_member(run_id, init_params, rand_values, A_list):
    """`--dry-run`: a member without device work (launcher / partition / collective rehearsal on CPUs) -- a
    deterministic function of the run's coefficients, so that the result files can be compared across world sizes."""
    params, f0, f1 = run_params(init_params, run_id, rand_values, A_list)
    a0, a1 = params.func_A0(params.temp), params.func_A1(params.temp)
    return (a0, a1, 0.1, 0.9, 0.2, 0.8, 100 + run_id, 1.5 * run_id, 7 * run_id, run_id,
            np.nan if f0 is None else f0, np.nan if f1 is None else f1)


def _launch_own_ranks(a, argv):
    """`--gpus G` without a launcher around it: this process -- which has not touched the GPU -- starts the G ranks as
    ordinary child processes, one per GPU, as the reference starts its own pool (experiment.py:197-216), and forwards
    rank 0's output."""
    import sys
    from . import launch
    if not a.dry_run:
        import importlib.util
        entry = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), '__graft_entry__.py')
        if os.path.exists(entry):   # build once, before the ranks start (hipcc only; no device needed)
            spec = importlib.util.spec_from_file_location('__graft_entry__', entry)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            mod.build_hip()
    args = list(sys.argv[1:] if argv is None else argv)
    failed, codes, texts = launch.spawn_ranks([sys.executable, '-m', 'chsimpy_amd.experiment'] + args, a.gpus,
                                              timeout_s=float(os.environ.get('CHS_LAUNCH_TIMEOUT', '86400')))
    if failed is not None:
        launch.report_failure('chsimpy_amd.experiment', failed, codes, texts)
        raise SystemExit(1)
    sys.stdout.write(texts[0])
    sys.stdout.flush()


def main(argv=None):
    ap = argparse.ArgumentParser(description='chsimpy_amd ensemble (cf. chsimpy-experiment)')
    ap.add_argument('-N', type=int, default=512)
    ap.add_argument('-n', '--ntmax', type=int, default=int(1e6))
    ap.add_argument('-R', '--runs', type=int, default=3)
    ap.add_argument('--independent', action='store_true')
    ap.add_argument('--A-source', default='uniform')
    ap.add_argument('--A-seed', type=int, default=85972)
    ap.add_argument('-K', '--kappa-tilde', type=float, default=None)
    ap.add_argument('--full-sim', action='store_true')
    ap.add_argument('--file-id', default='auto')
    ap.add_argument('--export-csv', default=None)
    ap.add_argument('--Uinit-file', default=None)
    ap.add_argument('--concurrent', type=int, default=2, help='ensemble members running at once per GPU')
    ap.add_argument('--gpus', type=int, default=1, help='start this many ranks (one per GPU) from here when no launcher '
                    'such as torch.distributed.run has set RANK/WORLD_SIZE')
    ap.add_argument('--backend', default=os.environ.get('CHS_DIST_BACKEND', 'nccl'),
                    help='torch.distributed backend: nccl (= RCCL, ranks on GPUs) or gloo (CPU rehearsal)')
    ap.add_argument('--dry-run', action='store_true', help='members without device work: rehearses launcher, partition, '
                    'core placement and the gather on CPUs; the result files are not results')
    a = ap.parse_args(argv)

    if a.gpus > 1 and 'RANK' not in os.environ:
        return _launch_own_ranks(a, argv)

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # Placement first, before anything touches the GPU: this rank's share of the host cores (the reference sizes its
    # pool to the physical cores, experiment.py:197-202; here the cores are dealt to the ranks), one torch intra-op thread
    from . import launch
    pinned = launch.pin_rank_to_cores(local_rank, world)
    dist = None
    device = 'cpu'
    if world > 1:
        import torch
        import torch.distributed as dist
        launch.quiet_host_threads()
        if a.backend == 'nccl':
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend='nccl', device_id=torch.device('cuda', local_rank))
            device = f'cuda:{local_rank}'
        else:
            dist.init_process_group(backend=a.backend)

    p = Parameters()
    p.N, p.ntmax, p.full_sim, p.kappa_tilde = a.N, a.ntmax, a.full_sim, a.kappa_tilde
    p.no_gui, p.export_csv, p.Uinit_file = True, a.export_csv, a.Uinit_file
    # (CHS_SAME_GPU=1: every rank on device 0 -- rehearsing the multi-rank path with real device work on a one-GPU box)
    p.device = 0 if os.environ.get('CHS_SAME_GPU') == '1' else local_rank
    p.file_id = utils.get_or_create_file_id(a.file_id)
    ep = ExperimentParams()
    ep.runs, ep.independent, ep.A_source, ep.A_seed = a.runs, a.independent, a.A_source, a.A_seed
    U_init = utils.csv_import_matrix(p.Uinit_file) if p.Uinit_file else None

    if rank == 0:
        write_metadata(p.file_id, ep, extra=[f"ranks, {world}", f"concurrent_per_rank, {a.concurrent}",
                                             f"host_cores_per_rank, {'all' if pinned is None else len(pinned)}"]
                       + (["dry_run, True"] if a.dry_run else []))
    records = run_ensemble(p, ep, run_fn=_dry_member if a.dry_run else None, U_init=U_init, dist=dist, rank=rank,
                           world=world, device=device, concurrent=a.concurrent)
    if rank == 0:
        df, agg = write_results(p.file_id, records)
        print(agg.T)
        print('Output files:')
        print(f"  {p.file_id}-metadata.csv")
        print(f"  {p.file_id}-results-agg.csv")
        print(f"  {p.file_id}-results.csv")
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
