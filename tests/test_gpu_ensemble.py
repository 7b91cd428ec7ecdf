"""GPU tests (``-m gpu``) of the ensemble path: BASELINE.json configs[4] (N=2048 members with scaled
A0/A1, several at once per GPU) and the `file` A-source, members against the oracle."""
import os

import numpy as np
import pytest

from chsimpy_amd import experiment as ex, utils
from oracle import chs_oracle as orc
from gpu_helpers import make, relerr

pytestmark = pytest.mark.gpu


def test_config4_members_n2048_concurrent(gpu, tmp_path):
    """configs[4] at its own grid size: four of the 64 (A0, A1) samples of PCG64(85972), N=2048, two
    members at a time on one GPU (one engine handle = one HIP stream each), every member against the
    oracle run with the same factors: A0, A1, tsep = argmax(E2) exactly, the E/E2 records to 1e-9."""
    N, nt, runs = 2048, 24, 4
    p = make(N, nt, 'fast')
    p.file_id = str(tmp_path / 'c4')
    p.export_csv = 'E,E2'
    ep = ex.ExperimentParams()
    ep.runs = 64
    rv, _, n = ex.make_rand_values(ep)
    assert n == 64
    ep.runs = runs                      # the first four samples of the same stream
    assert np.array_equal(ex.make_rand_values(ep)[0], rv[:runs])
    ex.write_metadata(p.file_id, ep)
    recs = ex.run_ensemble(p, ep, concurrent=2)
    assert [int(r[9]) for r in recs] == list(range(runs))
    for i, rec in enumerate(recs):
        f0, f1 = rv[i]
        o = orc.OracleSolver(orc.make_params(N, nt, func_A0=lambda T, f=f0: orc.A0(T) * f,
                                             func_A1=lambda T, f=f1: orc.A1(T) * f))
        o.prepare()
        o.solve_or_resume()
        to = o.timedata.data()
        assert rec[0] == o.A0 and rec[1] == o.A1 and rec[10] == f0 and rec[11] == f1
        assert rec[8] == int(np.argmax(to[:, 2]))
        E = utils.csv_import_matrix(f"{p.file_id}-run{i}.solution.E.csv")
        E2 = utils.csv_import_matrix(f"{p.file_id}-run{i}.solution.E2.csv")
        assert E.shape == E2.shape == (nt,)
        assert np.allclose(E, to[:, 1], rtol=1e-9, atol=0), relerr(E, to[:, 1])
        assert np.allclose(E2, to[:, 2], rtol=1e-9, atol=0), relerr(E2, to[:, 2])
        assert 0.7 < rec[2] < 0.9 < rec[3] < 1.0          # common-tangent compositions (sympy), experiment.py:110
        assert 0.0 < rec[4] < rec[5] < 1.0                # spinodal compositions
    df, agg = ex.write_results(p.file_id, recs)
    for suffix in ('-metadata.csv', '-results.csv', '-results-agg.csv'):
        assert os.path.exists(p.file_id + suffix)
    assert list(df['id']) == list(range(runs)) and 'cv' in agg.index


def test_file_A_source_members_on_gpu(gpu, tmp_path):
    """`--A-source <file>`: absolute (A0, A1) per member (experiment.py:97-101, 189-190)."""
    A = np.array([[-151.0, -85.5], [-151.5, -85.75]])
    f = str(tmp_path / 'A.csv')
    utils.csv_export_matrix(A, f)
    p = make(128, 30, 'fast')
    p.file_id = str(tmp_path / 'fa')
    ep = ex.ExperimentParams()
    ep.runs, ep.A_source = 2, f
    recs = ex.run_ensemble(p, ep)
    for i, rec in enumerate(recs):
        o = orc.OracleSolver(orc.make_params(128, 30, func_A0=lambda T, a=A[i, 0]: a, func_A1=lambda T, a=A[i, 1]: a))
        o.prepare()
        o.solve_or_resume()
        assert rec[0] == A[i, 0] and rec[1] == A[i, 1] and np.isnan(rec[10])
        assert rec[8] == int(np.argmax(o.timedata.data()[:, 2]))
