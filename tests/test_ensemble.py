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
