"""CPU tests of the ensemble partition (N > 1 path): gloo, world_size 2."""
import os
import socket
import sys

import numpy as np
import pytest

import chsimpy_amd
from chsimpy_amd import experiment as ex

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fake_run(run_id, init_params, rand_values, A_list):
    """Stand-in for the GPU run: a deterministic function of the run's factors."""
    params, f0, f1 = ex.run_params(init_params, run_id, rand_values, A_list)
    a0, a1 = params.func_A0(params.temp), params.func_A1(params.temp)
    return (a0, a1, 0.1, 0.9, 0.2, 0.8, 100 + run_id, 1.5 * run_id, 7 * run_id, run_id,
            np.nan if f0 is None else f0, np.nan if f1 is None else f1)


def _worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group(backend='gloo', rank=rank, world_size=world)
    p = chsimpy_amd.Parameters()
    p.N, p.kappa_tilde, p.file_id = 16, 3e-4, 'ens'
    ep = ex.ExperimentParams()
    ep.runs = 7
    recs = ex.run_ensemble(p, ep, run_fn=_fake_run, dist=dist, rank=rank, world=world)
    if rank == 0:
        np.save(out, np.array(recs, dtype=np.float64))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_factor_table_matches_reference_recipe():
    ep = ex.ExperimentParams()
    ep.runs = 64
    rv, al, n = ex.make_rand_values(ep)
    ref = np.random.Generator(np.random.PCG64(85972)).uniform(0.995, 1.005, size=(64, 2))
    assert n == 64 and al is None and np.array_equal(rv, ref)
    ep.independent = True
    rv, _, n = ex.make_rand_values(ep)
    assert n == 128 and np.all(rv[:64, 1] == 1) and np.all(rv[64:, 0] == 1)
    ep = ex.ExperimentParams()
    ep.runs, ep.A_source = 10, 'grid'
    rv, _, n = ex.make_rand_values(ep)
    assert n == 9 and rv.shape == (9, 2) and rv[0, 0] == 0.995 and rv[-1, 1] == 1.005
    ep = ex.ExperimentParams()
    ep.runs, ep.A_source = 5, 'sobol'
    rv, _, n = ex.make_rand_values(ep)
    assert n == 5 and np.all((rv >= 0.995) & (rv <= 1.005))


def test_partition_is_run_id_mod_world():
    assert ex.my_run_ids(7, 0, 2) == [0, 2, 4, 6] and ex.my_run_ids(7, 1, 2) == [1, 3, 5]
    assert sorted(sum((ex.my_run_ids(64, r, 8) for r in range(8)), [])) == list(range(64))


def test_gloo_world2_gather_equals_single_rank(tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / 'recs.npy')
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    p = chsimpy_amd.Parameters()
    p.N, p.kappa_tilde, p.file_id = 16, 3e-4, 'ens'
    ep = ex.ExperimentParams()
    ep.runs = 7
    single = np.array(ex.run_ensemble(p, ep, run_fn=_fake_run), dtype=np.float64)
    assert got.shape == (7, 12)
    assert np.array_equal(got, single)  # independent of the partition
    assert list(got[:, 9]) == list(range(7))


def test_results_csv_layout(tmp_path):
    recs = [_fake_run(i, *_ctx()) for i in range(4)]
    fid = str(tmp_path / 'e')
    df, agg = ex.write_results(fid, recs)
    assert list(df.columns) == ex.COLS
    import pandas as pd
    back = pd.read_csv(fid + '-results.csv', index_col=0)
    assert list(back.columns) == ex.COLS and back['id'].tolist() == [0, 1, 2, 3]
    aggb = pd.read_csv(fid + '-results-agg.csv', index_col=0)
    assert 'cv' in aggb.columns and 'A0' in aggb.index


def _ctx():
    p = chsimpy_amd.Parameters()
    p.N, p.kappa_tilde, p.file_id = 16, 3e-4, 'ens'
    ep = ex.ExperimentParams()
    ep.runs = 4
    rv, al, _ = ex.make_rand_values(ep)
    return p, rv, al


def test_metadata_csv(tmp_path):
    """`<file_id>-metadata.csv` (experiment.py:193-195): system info, then the experiment parameters."""
    ep = ex.ExperimentParams()
    ep.runs, ep.A_source = 12, 'sobol'
    fid = str(tmp_path / 'exp')
    f = ex.write_metadata(fid, ep, extra=['ranks, 2'])
    assert f == fid + '-metadata.csv'
    lines = open(f).read().split('\n')
    keys = [ln.split(',')[0] for ln in lines]
    for k in ('system', 'nodename', 'cores_total', 'localtime', 'argv', 'ranks', 'A_seed', 'A_source', 'independent',
              'jitter_Arelhigh', 'jitter_Arellow', 'processes', 'runs'):
        assert k in keys, k
    assert 'runs, 12' in lines and 'A_source, sobol' in lines and 'A_seed, 85972' in lines


def test_file_A_source(tmp_path):
    """`--A-source <file>` (experiment.py:189-190, 97-101): absolute (A0, A1) pairs from a CSV, no factors."""
    A = np.array([[-151.0, -85.5], [-151.5, -85.75], [-150.25, -86.0]])
    f = str(tmp_path / 'A.csv')
    chsimpy_amd.utils.csv_export_matrix(A, f)
    ep = ex.ExperimentParams()
    ep.runs, ep.A_source = 5, f
    rv, al, n = ex.make_rand_values(ep)
    assert rv is None and n == 3 and np.array_equal(al, A)
    p = chsimpy_amd.Parameters()
    p.N, p.kappa_tilde, p.file_id = 16, 3e-4, 'ens'
    recs = ex.run_ensemble(p, ep, run_fn=_fake_run)
    assert len(recs) == 3
    for i, rec in enumerate(recs):
        assert rec[0] == A[i, 0] and rec[1] == A[i, 1] and rec[9] == i
        assert np.isnan(rec[10]) and np.isnan(rec[11])
    q, f0, f1 = ex.run_params(p, 1, rv, al)
    assert f0 is None and f1 is None and q.file_id == 'ens-run1'
    assert chsimpy_amd.Solution(q).A0 == A[1, 0] and chsimpy_amd.Solution(q).A1 == A[1, 1]


def test_postprocessing_errors_are_raised_not_swallowed(monkeypatch, tmp_path):
    """experiment.py:110-112: the thermodynamic post-processing is part of the run; when it fails the
    run fails (no NaN columns slipping into the aggregate)."""
    from chsimpy_amd import experiment, simulator

    class _Sol:
        A0, A1, tau0, t0 = -151.0, -85.0, 3, 0.5
        E2 = np.array([1.0, 3.0, 2.0])

    class _Sim:
        def __init__(self, params, U_init=None):
            self.solver = type('S', (), {'close': lambda self, fetch_U=True: None})()

        def solve(self):
            return _Sol()

        def export(self):
            return None

    monkeypatch.setattr(simulator, 'Simulator', _Sim)
    monkeypatch.setattr(experiment.utils, 'get_miscibility_gap', lambda *a, **k: (_ for _ in ()).throw(ValueError('no gap')))
    p = chsimpy_amd.Parameters()
    p.N, p.kappa_tilde, p.file_id = 16, 3e-4, str(tmp_path / 'e')
    ep = ex.ExperimentParams()
    rv, al, _ = ex.make_rand_values(ep)
    with pytest.raises(ValueError):
        ex.run_experiment_gpu(0, p, rv, al)
    rec = ex.run_experiment_gpu(0, p, rv, al, postprocess=False)   # explicit opt-out: NaN columns
    assert np.isnan(rec[2]) and rec[8] == 1 and rec[6] == 3


# ---------------------------------------------------------------------------------------------------------------
# world size 8 (the node the ensemble is meant for: one rank per MI355X), rehearsed on CPUs with gloo
# ---------------------------------------------------------------------------------------------------------------
def _worker8(rank, world, port, outdir):
    """One rendezvous, three ensembles: BASELINE.json configs[4]'s 64 runs, an uneven count (61: the last round of the
    deal is short, the gather block is padded) and --independent (2 x 32 items)."""
    import torch.distributed as dist
    from chsimpy_amd import launch
    pinned = launch.pin_rank_to_cores(rank, world)       # before anything else, as experiment.main does
    launch.quiet_host_threads()
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group(backend='gloo', rank=rank, world_size=world)
    p = chsimpy_amd.Parameters()
    p.N, p.kappa_tilde, p.file_id = 16, 3e-4, 'ens'
    for name, runs, indep in (('c64', 64, False), ('c61', 61, False), ('ind', 32, True)):
        ep = ex.ExperimentParams()
        ep.runs, ep.independent = runs, indep
        recs = ex.run_ensemble(p, ep, run_fn=_fake_run, dist=dist, rank=rank, world=world)
        np.save(os.path.join(outdir, f'{name}_rank{rank}.npy'), np.array(recs, dtype=np.float64))
    np.save(os.path.join(outdir, f'cores_rank{rank}.npy'), np.array(sorted(pinned) if pinned else [-1]))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world8_deals_64_61_and_independent_runs(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_worker8, args=(8, _free_port(), str(tmp_path)), nprocs=8, join=True)
    p = chsimpy_amd.Parameters()
    p.N, p.kappa_tilde, p.file_id = 16, 3e-4, 'ens'
    for name, runs, indep, items in (('c64', 64, False, 64), ('c61', 61, False, 61), ('ind', 32, True, 64)):
        ep = ex.ExperimentParams()
        ep.runs, ep.independent = runs, indep
        single = np.array(ex.run_ensemble(p, ep, run_fn=_fake_run), dtype=np.float64)
        assert single.shape == (items, 12)
        for r in range(8):                                  # every rank holds every record, ordered by run id
            got = np.load(str(tmp_path / f'{name}_rank{r}.npy'))
            assert got.shape == single.shape and np.array_equal(got, single, equal_nan=True), (name, r)
        assert list(single[:, 9]) == list(range(items))
    # --independent: the first half varies A0 only, the second A1 only (experiment.py:163-170)
    ind = np.load(str(tmp_path / 'ind_rank0.npy'))
    assert np.all(ind[:32, 11] == 1.0) and np.all(ind[32:, 10] == 1.0) and not np.any(ind[:32, 10] == 1.0)
    # core placement: the ranks' core sets are disjoint slices of what this process may use (when there are >= 8)
    sets = [set(np.load(str(tmp_path / f'cores_rank{r}.npy')).tolist()) for r in range(8)]
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) >= 8:
        assert all(s and -1 not in s for s in sets)
        assert sum(len(s) for s in sets) == len(set().union(*sets)) == len(allowed)
    assert sorted(os.sched_getaffinity(0)) == allowed      # the parent's own affinity is untouched


def test_core_slices():
    from chsimpy_amd import launch
    assert launch.core_slices(range(16), 8) == [[2 * r, 2 * r + 1] for r in range(8)]
    assert launch.core_slices(range(8), 3) == [[0, 1, 2], [3, 4, 5], [6, 7]]
    assert launch.core_slices([4, 9], 4) == [[4], [9], [4], [9]]        # fewer cores than ranks: shared
    assert launch.core_slices([], 2) == [[], []]
    assert launch.pin_rank_to_cores(0, 1) is None                       # one rank: nothing to deal


def test_experiment_starts_its_own_8_ranks_and_the_files_do_not_depend_on_the_world_size(tmp_path):
    """`python -m chsimpy_amd.experiment --gpus 8` without a launcher around it (a parent that never touches the GPU
    starts the ranks, as chsimpy/experiment.py:197-216 starts its pool), `--dry-run --backend gloo` standing in for
    the device work: 61 runs over 8 ranks give the result files of a single rank."""
    import subprocess
    import pandas as pd
    e = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    e['PYTHONPATH'] = ROOT + os.pathsep + e.get('PYTHONPATH', '')
    base = ['-N', '16', '-K', '3e-4', '-R', '61', '--dry-run', '--backend', 'gloo']
    r8 = subprocess.run([sys.executable, '-m', 'chsimpy_amd.experiment', '--gpus', '8', '--file-id', str(tmp_path / 'w8')] + base,
                        env=e, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r8.returncode == 0, r8.stderr[-3000:]
    r1 = subprocess.run([sys.executable, '-m', 'chsimpy_amd.experiment', '--file-id', str(tmp_path / 'w1')] + base,
                        env=e, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r1.returncode == 0, r1.stderr[-3000:]
    a, b = pd.read_csv(str(tmp_path / 'w8-results.csv'), index_col=0), pd.read_csv(str(tmp_path / 'w1-results.csv'), index_col=0)
    assert a.shape == (61, 12) and a.equals(b)
    assert open(str(tmp_path / 'w8-results-agg.csv')).read() == open(str(tmp_path / 'w1-results-agg.csv')).read()
    meta = open(str(tmp_path / 'w8-metadata.csv')).read()
    assert 'ranks, 8' in meta and 'dry_run, True' in meta and 'host_cores_per_rank' in meta
    assert 'Output files:' in r8.stdout            # rank 0's report is forwarded by the launching parent


def test_experiment_as_ranks_of_torch_distributed_run(tmp_path):
    """The other way in: `python -m torch.distributed.run --nproc-per-node 2 -m chsimpy_amd.experiment ...` (RANK / WORLD_SIZE
    come from the launcher: no ranks of its own then), gloo + --dry-run on CPUs; rank 0 writes the files of a single rank."""
    import subprocess
    import pandas as pd
    e = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    e['PYTHONPATH'] = ROOT + os.pathsep + e.get('PYTHONPATH', '')
    base = ['-N', '16', '-K', '3e-4', '-R', '7', '--dry-run', '--backend', 'gloo', '--gpus', '2']
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', str(_free_port()), '-m', 'chsimpy_amd.experiment', '--file-id', str(tmp_path / 't2')] + base,
                       env=e, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    r1 = subprocess.run([sys.executable, '-m', 'chsimpy_amd.experiment', '--file-id', str(tmp_path / 't1')] + base[:-2],
                        env=e, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r1.returncode == 0, r1.stderr[-3000:]
    a, b = pd.read_csv(str(tmp_path / 't2-results.csv'), index_col=0), pd.read_csv(str(tmp_path / 't1-results.csv'), index_col=0)
    assert a.shape == (7, 12) and a.equals(b)
    assert 'ranks, 2' in open(str(tmp_path / 't2-metadata.csv')).read()
