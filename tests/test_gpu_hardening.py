"""GPU tests (``-m gpu``) of what concurrency and long runs lean on: gated launches from several streams at once,
the gate's bounded timeout, per-handle staging, the timedata ring beyond one lap, fields edited between calls."""
import os
import threading

import numpy as np
import pytest

import chsimpy_amd
from chsimpy_amd import _lib
from chsimpy_amd import experiment as ex
from oracle import chs_oracle as orc
from gpu_helpers import KAPPA, make, relerr

pytestmark = pytest.mark.gpu


def test_adaptive_members_concurrent_three_streams_vs_oracle(gpu, tmp_path):
    """Three adaptive-dt members at once on one GPU (one handle = one stream each): every k_col launch beyond step
    500 is a GATED launch (its workgroups wait for the riding bookkeeping's delt), now from three streams
    interleaved.  Each member against the oracle with the same (A0, A1) factors: delt history, E, E2
    (cf. chsimpy/experiment.py:84-126 with parameters.py:57 adaptive_time)."""
    N, nt, runs = 256, 560, 6
    kw = dict(adaptive_time=True, delt_max=4.9e-7 / N)   # (the dynamic step is a column SUM: it scales with N, solver.py:183)
    p = make(N, nt, 'fast', **kw)
    p.file_id = str(tmp_path / 'ad')
    p.export_csv = 'E,E2,delt'
    ep = ex.ExperimentParams()
    ep.runs = runs
    rv, _, _ = ex.make_rand_values(ep)
    recs = ex.run_ensemble(p, ep, run_fn=lambda i, pp, r, al: ex.run_experiment_gpu(i, pp, r, al, None, postprocess=False),
                           concurrent=3)
    assert [int(r[9]) for r in recs] == list(range(runs))
    from chsimpy_amd import utils
    for i in range(runs):
        f0, f1 = rv[i]
        o = orc.OracleSolver(orc.make_params(N, nt, func_A0=lambda T, f=f0: orc.A0(T) * f,
                                             func_A1=lambda T, f=f1: orc.A1(T) * f, **kw))
        o.prepare()
        o.solve_or_resume()
        to = o.timedata.data()
        assert to[-1, 8] > 1.5 * to[0, 8]                      # the adaptive step did fire
        for name, col, tol in (('delt', 8, 1e-9), ('E', 1, 1e-9), ('E2', 2, 1e-8)):
            got = utils.csv_import_matrix(f"{p.file_id}-run{i}.solution.{name}.csv")
            assert got.shape == (nt,)
            assert np.allclose(got, to[:, col], rtol=tol, atol=0), (i, name, relerr(got, to[:, col]))
        assert recs[i][8] == int(np.argmax(to[:, 2]))


def test_gate_timeout_is_an_error_and_the_handle_stays_usable(gpu, monkeypatch):
    """Test hook CHS_TEST_GATE_WITHHOLD=1 (read at every chs_step_n): the riding bookkeeping never publishes its
    decision, the waiting workgroups of the gated k_col give up after their bounded number of polls -> chs_step_n
    returns CHS_EHIP (no hang, no stepping on with a column pass that did not happen).  With the hook off the SAME
    handle then runs prepare() + solve to the oracle's result: the error left no stale halt flag, sequence number
    or residency behind."""
    N, nt = 128, 40
    kw = dict(adaptive_time=True, delt_max=4.9e-7 / N)   # an adaptive time step: every k_col after the first is a gated launch
    monkeypatch.setenv('CHS_TEST_GATE_WITHHOLD', '1')
    s = chsimpy_amd.Solver(make(N, nt, 'fast', **kw))
    s.prepare()
    with pytest.raises(_lib.EngineError, match='gave up waiting'):
        s.solve_or_resume(30)
    monkeypatch.delenv('CHS_TEST_GATE_WITHHOLD')
    s.prepare()
    sol = s.solve_or_resume()
    o = orc.OracleSolver(orc.make_params(N, nt, **kw))
    o.prepare(); o.solve_or_resume()
    assert sol.computed_steps == o.computed_steps == nt
    assert np.allclose(sol.timedata.data(), o.timedata.data(), rtol=1e-9, atol=1e-300)
    assert np.allclose(sol.U, o.U, rtol=1e-9, atol=0), relerr(sol.U, o.U)
    s.close()


def test_concurrent_handles_move_their_fields_at_once(gpu):
    """Per-handle pinned staging: three handles up- and download different fields from three threads at the same
    time; every one gets its own bytes back (fp64 and fp32)."""
    N = 512
    rng = np.random.default_rng(7)
    fields = [0.8 + 0.1 * rng.random((N, N)) for _ in range(3)]
    solvers = [chsimpy_amd.Solver(make(N, 5, 'fast', dtype=('float32' if i == 2 else 'float64')), fields[i]) for i in range(3)]
    out = [None] * 3

    def work(i):
        for _ in range(4):
            solvers[i]._get_engine().set_U(fields[i])
            out[i] = solvers[i]._engine.get_U()

    th = [threading.Thread(target=work, args=(i,)) for i in range(3)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert np.array_equal(out[0], fields[0]) and np.array_equal(out[1], fields[1])
    assert np.array_equal(out[2], fields[2].astype(np.float32).astype(np.float64))
    for s in solvers:
        s.close(fetch_U=False)


def test_timedata_ring_wraps_70000_steps_one_call_equals_chunks(gpu):
    """The device keeps the rows of a call in a ring of 65536 (chs_api.hip): a full_sim call of 70 000 steps laps
    it.  One call and five chunks must give the same 70 000 rows bit for bit (the chunks continue the device loop),
    with consecutive step numbers and a field that still conserves mass."""
    N, total = 128, 70001
    a = chsimpy_amd.Solver(make(N, total, 'fast', delt=2e-9))
    a.prepare()
    sa = a.solve_or_resume()
    ta = sa.timedata.data()
    assert ta.shape == (total, 9)
    assert np.array_equal(ta[:, 0], np.arange(total, dtype=np.float64))
    b = chsimpy_amd.Solver(make(N, total, 'fast', delt=2e-9))
    b.prepare()
    for n in (20001, 30000, 15000, 4999, 1):
        sb = b.solve_or_resume(n)
    tb = sb.timedata.data()
    assert tb.shape == ta.shape
    assert np.array_equal(ta, tb)
    assert np.array_equal(sa.U, sb.U)
    assert abs(sa.U.mean() - a.U_init.mean()) < 1e-12
    a.close(); b.close()


def test_field_edited_in_place_between_calls_is_uploaded(gpu):
    """An update callback that edits `solution.U` in place: the next call must continue from the edited array,
    like the reference does (solver.py:158), not from the stale device field."""
    N = 128
    s = chsimpy_amd.Solver(make(N, 100, 'fast'))
    o = orc.OracleSolver(orc.make_params(N, 100))
    s.prepare(); o.prepare()
    sol = s.solve_or_resume(10); o.solve_or_resume(10)
    u = sol.U
    u[5:9, 7:20] *= 1.001
    o.U[5:9, 7:20] *= 1.001
    sol = s.solve_or_resume(15); o.solve_or_resume(15)
    assert np.allclose(sol.U, o.U, rtol=1e-9, atol=0), relerr(sol.U, o.U)
    s.close()


def test_pool_clear_frees_parked_engines(gpu):
    s = chsimpy_amd.Solver(make(1024, 5, 'fast'))
    s.prepare()
    s.close(fetch_U=False)            # parked
    assert _lib.pool_count() >= 1
    _lib.pool_clear()
    assert _lib.pool_count() == 0
    s2 = chsimpy_amd.Solver(make(1024, 5, 'fast'))   # a new engine is simply created
    s2.prepare()
    sol = s2.solve_or_resume()
    assert sol.computed_steps == 5
    s2.close()
    assert _lib.pool_count() == 1


def test_default_workflow_repeated_in_one_process_keeps_its_pace(gpu):
    """The reference's default run (Parameters() as shipped: N=512, energy stop at step 1674), six times in one process
    with the field downloaded after each: the runs must not fall to half the pace of the fastest one.  (Round 3: the download
    path had called a BLAS routine; the thread pool it woke used up the CPU quota of the container and the kernel
    launches of the NEXT run were throttled for ~68 ms -- 97 instead of 35 ms per run; nothing a parity test sees.)"""
    ms = []
    for _ in range(6):
        p = chsimpy_amd.Parameters()
        p.N, p.kappa_tilde, p.no_gui = 512, 0.0002989112919661156, True
        s = chsimpy_amd.Solver(p)
        s.prepare()
        sol = s.solve_or_resume(p.ntmax)
        assert sol.stop_reason == 'energy' and sol.computed_steps == 1674
        ms.append(s._engine.last_step_ms())
        s.close()            # downloads the field
    # The mechanism itself (no BLAS entry point in the download path) is asserted on the CPU:
    # tests/test_host.py::test_download_path_never_enters_blas.  Device wall times on a shared box depend on its load,
    # so the pace is reported, not asserted; only a collapse far beyond the old fault (2.8x) fails.
    slow = [m for m in ms if m > 2.0 * min(ms)]
    if len(slow) > 1:
        import warnings
        warnings.warn(f"default workflow: {len(slow)} of 6 runs slower than twice the fastest: {ms}")
    assert max(ms) < 20.0 * min(ms), ms
