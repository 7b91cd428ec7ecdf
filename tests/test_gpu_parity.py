"""GPU parity tests (``-m gpu``): the HIP path, called through the C ABI, against the
oracle on identical inputs.  Tolerance: BASELINE.json's north_star asks for per-step
U within rtol 1e-9 of the numpy/scipy path (fp64); energies E/E2 likewise.
"""
import os

import numpy as np
import pytest
import scipy.fftpack as scifft

import chsimpy_amd
from chsimpy_amd import _lib
from oracle import chs_oracle as orc

pytestmark = pytest.mark.gpu

from gpu_helpers import GOLD, KAPPA, RTOL, compare_run, log_line, make, relerr  # noqa: F401


@pytest.mark.parametrize("engine,N", [('direct', 64), ('direct', 100), ('direct', 128), ('fast', 128), ('fast', 256),
                                      ('fast', 512), ('fast', 1024), ('fast', 2048), ('fast', 4096), ('fast', 8192)])
def test_dctn_matches_scipy(gpu, engine, N):
    p = make(N, 2, engine)
    s = chsimpy_amd.Solver(p)
    eng = s._get_engine()
    assert eng.engine == engine
    X = np.random.default_rng(N).standard_normal((N, N))
    Y = eng.dctn(X)
    Yr = scifft.dctn(X, norm='ortho')
    assert np.max(np.abs(Y - Yr)) < 1e-12 * np.max(np.abs(Yr))
    Z = eng.dctn(Yr, inverse=True)
    assert np.max(np.abs(Z - X)) < 1e-12 * np.max(np.abs(X))
    s.close()


@pytest.mark.parametrize("engine", ['direct', 'fast'])
def test_mu_pointwise(gpu, engine):
    p = make(128, 2, engine)
    s = chsimpy_amd.Solver(p)
    eng = s._get_engine()
    eng.set_U(s.U_init)
    o = orc.OracleSolver(orc.make_params(128, 2))
    assert np.allclose(eng.get_mu(), o.mu(o.U_init), rtol=1e-13, atol=0)
    s.close()


@pytest.mark.parametrize("engine", ['direct', 'fast'])
def test_config0_n128_200steps_vs_oracle_and_golden(gpu, engine):
    """BASELINE.json configs[0]: N=128, ntmax=200, seed=2023, cinit=0.875."""
    sol, o = compare_run(make(128, 200, engine), {})
    g = np.load(os.path.join(GOLD, 'n128_seed2023_200steps.npz'))
    assert np.allclose(sol.timedata.data()[:, 1:3], g['timedata'][:, 1:3], rtol=RTOL, atol=0)
    assert np.allclose(sol.U, g['U_final'], rtol=RTOL, atol=0)
    # reference observations
    assert sol.E[-1] == pytest.approx(-5.453709377934683e-11, rel=1e-9)
    assert sol.E2[-1] == pytest.approx(8.73563379288026e-18, rel=1e-9)


@pytest.mark.parametrize("engine,N,nt", [('direct', 64, 40), ('direct', 128, 60), ('fast', 128, 60)])
def test_golden_lcg_fixture(gpu, engine, N, nt):
    """The reference-pinned start field (`generator='lcg'`, solver.py:66, known-answer test
    tests/test.py:25-37) through both transform engines against the committed fixtures."""
    g = np.load(os.path.join(GOLD, f'n{N}_lcg_{nt}steps.npz'))
    p = make(N, nt, engine, generator='lcg')
    s = chsimpy_amd.Solver(p)
    assert np.array_equal(s.U_init, g['U_init'])
    s.prepare()
    sol = s.solve_or_resume()
    assert s._engine.engine == engine
    assert np.allclose(sol.timedata.data(), g['timedata'], rtol=RTOL, atol=1e-300)
    assert np.allclose(sol.U, g['U_final'], rtol=RTOL, atol=0)
    s.close()


def test_non_power_of_two_n100(gpu):
    """`benchmark.py -N 100` of the reference's smoke script (tests/run-tests.sh:15)."""
    compare_run(make(100, 60, 'auto'), {})


def test_config1_n512_1000steps(gpu):
    """BASELINE.json configs[1]: N=512, ntmax=1000, fp64 -- U/E/E2 vs numpy at rtol 1e-9."""
    compare_run(make(512, 1000, 'auto'), {})


@pytest.mark.parametrize("N,nt", [(1024, 100), (2048, 40), (4096, 12)])
def test_large_grids_against_the_oracle(gpu, N, nt, monkeypatch):
    """The larger fast-engine configurations step by step against the oracle itself, as far as the
    oracle's CPU time allows (about 0.1 us per grid point and step): every configuration has its own
    workgroup shapes and radices, N=4096 is the headline size.  The call is issued in batches of 5 steps
    (CHS_BATCH_STEPS; 1024 in production, which the oracle cannot reach at these sizes), so the window does
    cross issue batches -- the device-state poll between them, the row ring, the alternating partial-sum sets."""
    monkeypatch.setenv('CHS_BATCH_STEPS', '5')
    monkeypatch.setenv('CHS_ENGINE_POOL', '0')     # (a new engine, so that it reads the batch size)
    p = make(N, nt, 'fast')
    compare_run(p, {})


def test_resume_chunks_match_oracle_chunks(gpu):
    """solve_or_resume in chunks (simulator.py:62-81): nsteps-1 on the first call,
    hat_U re-derived per call."""
    p = make(128, 0)
    s = chsimpy_amd.Solver(p)
    o = orc.OracleSolver(orc.make_params(128, 0))
    s.prepare(); o.prepare()
    for chunk in (5, 7, 1, 12):
        sol = s.solve_or_resume(chunk)
        o.solve_or_resume(chunk)
        assert sol.computed_steps == o.computed_steps
        assert np.allclose(sol.U, o.U, rtol=RTOL, atol=0)
    assert np.allclose(sol.timedata.data(), o.timedata.data(), rtol=RTOL, atol=1e-300)
    # prepare() again resets the record but not delt/time_delta_sum (as the reference)
    s.prepare(); o.prepare()
    sol = s.solve_or_resume(4); o.solve_or_resume(4)
    assert sol.timedata.data().shape == (4, 9)
    assert np.allclose(sol.timedata.data(), o.timedata.data(), rtol=RTOL, atol=1e-300)
    s.close()


@pytest.mark.parametrize("N,engine", [(100, 'direct'), (512, 'fast')])
def test_start_field_drawn_on_the_device(gpu, N, engine):
    """Default generator: prepare() lets the device draw XXX + XXX*0.01*(rand - 0.5) (solver.py:78-82)
    from numpy's PCG64 stream -- bit for bit what the host expression gives -- instead of uploading it."""
    p = make(N, 3, engine)
    s = chsimpy_amd.Solver(p)
    assert s._U_init is None                       # not drawn on the host
    s.prepare()
    on_dev = s._engine.get_U()
    assert s._U_init is None
    rng = np.random.Generator(np.random.PCG64(p.seed))
    expect = p.XXX + (p.XXX * 0.01 * (rng.random((N, N)) - 0.5))
    assert np.array_equal(on_dev, expect)
    assert np.array_equal(s.U_init, expect)         # the attribute, computed on demand
    assert np.array_equal(s.solution.U, expect)
    assert s._pcg.bit_generator.state == rng.bit_generator.state   # the generator moved on by N*N draws
    s.close()


def test_solution_U_is_downloaded_on_demand(gpu):
    """Solution.U (solution.py:21) is fetched from the device when it is looked at, not after every
    chunk: chunked driving (simulator.py:62-81) then costs no N*N*8-byte PCIe transfer per chunk."""
    p = make(128, 0)
    s = chsimpy_amd.Solver(p)
    o = orc.OracleSolver(orc.make_params(128, 0))
    s.prepare(); o.prepare()
    sol = s.solve_or_resume(6); o.solve_or_resume(6)
    assert sol.__dict__['_U_fetch'] is not None and sol.__dict__['_U'] is None     # nothing downloaded yet
    sol = s.solve_or_resume(9); o.solve_or_resume(9)
    assert sol.__dict__['_U_fetch'] is not None
    U = sol.U                                                                        # now
    assert sol.__dict__['_U_fetch'] is None and U is sol.U
    assert np.allclose(U, o.U, rtol=RTOL, atol=0)
    sol = s.solve_or_resume(3); o.solve_or_resume(3)
    s.close()                                                                        # a pending download happens at close
    assert np.allclose(sol.U, o.U, rtol=RTOL, atol=0)
    s2 = chsimpy_amd.Solver(p); s2.prepare(); sol2 = s2.solve_or_resume(4)
    s2.close(fetch_U=False)
    assert sol2.U is None and 'U' not in sol2.scalars()


def test_energy_stop_without_full_sim(gpu):
    """full_sim=False: stop at the first step with E2[it-1] > E2[it] > E2[0]; the
    returned U is the one of the stopping step (solver.py:242-251).  delt=3e-6 reaches
    the E2 peak after 71 steps on the default N=64 start field."""
    p = make(64, 6000, 'auto', full_sim=False, delt=3e-6)
    sol, o = compare_run(p, dict(full_sim=False, delt=3e-6), rtol=1e-8)
    assert sol.stop_reason == 'energy' and sol.computed_steps == 72 and sol.tau0 == 72
    # with full_sim the same run continues and only records tau0/t0 once
    p = make(64, 100, 'auto', full_sim=True, delt=3e-6)
    sol, o = compare_run(p, dict(full_sim=True, delt=3e-6), rtol=1e-7)
    assert sol.stop_reason == 'None' and sol.computed_steps == 100 and sol.tau0 == 72


def test_resume_after_energy_stop(gpu):
    """The device loop runs one column pass past an energy stop (deferred bookkeeping); the stop
    step's U is what comes back and a further solve_or_resume continues from it exactly like the
    reference (hat_U re-derived from U, solver.py:159; the stop rule fires again at once while
    E2 keeps falling below its start value... or not at all -- whatever the oracle does)."""
    kw = dict(full_sim=False, delt=3e-6)
    p = make(64, 6000, 'auto', **kw)
    s = chsimpy_amd.Solver(p)
    o = orc.OracleSolver(orc.make_params(64, 6000, **kw))
    s.prepare(); o.prepare()
    sol = s.solve_or_resume(); o.solve_or_resume()
    assert sol.stop_reason == 'energy' and sol.computed_steps == o.computed_steps == 72
    assert np.allclose(sol.U, o.U, rtol=1e-8, atol=0)
    for chunk in (1, 5, 30):
        sol = s.solve_or_resume(chunk); o.solve_or_resume(chunk)
        assert sol.computed_steps == o.computed_steps
        assert sol.stop_reason == o.stop_reason and sol.tau0 == o.tau0
        assert np.allclose(sol.U, o.U, rtol=1e-8, atol=0)
    td, to = sol.timedata.data(), o.timedata.data()
    assert td.shape == to.shape and np.allclose(td, to, rtol=1e-8, atol=1e-300)
    s.close()


def test_time_limit_stop(gpu):
    p = make(64, 500, 'auto', time_max=30 * 3e-8 / 1.71e-8 / 60)  # ~30 steps of simulated time
    sol, o = compare_run(p, dict(time_max=p.time_max))
    assert sol.stop_reason == 'time-limit' and sol.computed_steps < 40


def test_time_limit_stop_fast_engine(gpu):
    """The same on the fused pipeline (N=128): U is not written between steps there, the field of the
    last completed step is rebuilt from hat_U when the limit ends the call (solver.py:197-199)."""
    p = make(128, 500, 'fast', time_max=30 * 3e-8 / 1.71e-8 / 60)
    sol, o = compare_run(p, dict(time_max=p.time_max))
    assert sol.stop_reason == 'time-limit' and sol.computed_steps < 40
    # ... and with the energy rule armed as well
    p = make(128, 500, 'fast', time_max=30 * 3e-8 / 1.71e-8 / 60, full_sim=False)
    sol, o = compare_run(p, dict(time_max=p.time_max, full_sim=False))
    assert sol.stop_reason == 'time-limit'


def test_adaptive_time(gpu):
    """adaptive_time beyond step 500 (solver.py:177-193) against the golden delt history."""
    g = np.load(os.path.join(GOLD, 'n64_adaptive_600.npz'))
    assert len(np.unique(g['timedata'][:, 8])) > 3  # the step really adapts
    p = make(64, 600, 'auto', adaptive_time=True)
    s = chsimpy_amd.Solver(p)
    s.prepare()
    sol = s.solve_or_resume()
    assert np.allclose(sol.timedata.data()[:, 8], g['timedata'][:, 8], rtol=1e-9, atol=0)
    assert np.allclose(sol.timedata.data()[:, 1:3], g['timedata'][:, 1:3], rtol=1e-8, atol=0)
    assert np.allclose(sol.U, g['U_final'], rtol=1e-8, atol=0)
    s.close()


@pytest.mark.parametrize("N,engine", [(64, 'auto'), (128, 'fast')])
def test_jitter_host_noise_stream(gpu, N, engine):
    """solver.py:210-211: U += jitter*(2*rand - 1) between the inverse transform and the record, the
    noise from the run's own generator (drawn on the device for numpy's PCG64, see below)."""
    p = make(N, 12, engine, jitter=0.001)
    compare_run(p, dict(jitter=0.001))


@pytest.mark.parametrize("N,engine", [(64, 'direct'), (256, 'fast')])
def test_jitter_noise_drawn_on_the_device_continues_the_host_stream(gpu, N, engine):
    """With the reference's default generator (numpy PCG64) the device draws the jitter noise itself
    (chs_set_jitter_pcg64): same stream as `create_rand(N)` on the host (solver.py:211), so the run is
    bit-identical to the host-noise path, chunk after chunk, and the host generator ends up in the
    same state."""
    nt = 25
    runs = {}
    for dev in (True, False):
        p = make(N, nt, engine, jitter=0.02)
        s = chsimpy_amd.Solver(p)
        s.device_rng = dev
        s.prepare()
        s.solve_or_resume(10)
        sol = s.solve_or_resume(nt - 10)
        runs[dev] = (sol.U.copy(), sol.timedata.data().copy(), s._pcg.bit_generator.state['state']['state'])
        s.close()
    assert runs[True][2] == runs[False][2]                      # generator state
    assert np.array_equal(runs[True][1], runs[False][1])        # every recorded scalar
    assert np.array_equal(runs[True][0], runs[False][0])        # the field


def test_nan_raises_assertion_like_the_reference(gpu):
    N = 64
    U0 = np.full((N, N), 0.875)
    U0[3, 4] = 1.5  # outside (0,1): log of a negative number
    p = make(N, 5)
    s = chsimpy_amd.Solver(p, U0)
    with pytest.raises(AssertionError):
        s.prepare()
        s.solve_or_resume()
    s.close()


def test_properties_at_full_size(gpu):
    """Size-independent properties at BASELINE's headline size (N=4096): transform
    round trip, mass conservation, fixed point, linearity of the transform."""
    N = 4096
    p = make(N, 4)
    s = chsimpy_amd.Solver(p)
    eng = s._get_engine()
    rng = np.random.default_rng(1)
    X = rng.standard_normal((N, N))
    Y = eng.dctn(X)
    assert Y[0, 0] == pytest.approx(X.sum() / N, rel=1e-10, abs=1e-9)
    assert np.sum(Y * Y) == pytest.approx(np.sum(X * X), rel=1e-12)  # orthonormal: Parseval
    Z = eng.dctn(Y, inverse=True)
    assert np.max(np.abs(Z - X)) < 1e-11
    # spot-check 3 columns of the transform against scipy along axis 0 then 1
    T = scifft.dct(scifft.dct(X[:, :], axis=1, norm='ortho')[:, [0, 17, N - 1]], axis=0, norm='ortho')
    assert np.max(np.abs(Y[:, [0, 17, N - 1]] - T)) < 1e-10
    s.prepare()
    sol = s.solve_or_resume(4)
    assert sol.U.mean() == pytest.approx(s.U_init.mean(), rel=1e-13)
    assert sol.timedata.data().shape == (4, 9)
    s.close()
    # fixed point
    p = make(N, 3)
    s = chsimpy_amd.Solver(p, np.full((N, N), 0.8))
    s.prepare()
    sol = s.solve_or_resume()
    assert np.max(np.abs(sol.U - 0.8)) < 1e-13
    assert np.all(np.abs(sol.E2) < 1e-40)
    s.close()


def test_config2_n4096_5000steps_invariants(gpu):
    """BASELINE.json configs[2] itself (N=4096, ntmax=5000, fp64), one solve_or_resume call on the fused
    pipeline (deferred bookkeeping, U kept in registers between steps): invariants of the scheme that do
    not need the oracle at this size -- mass conservation, energy decay (the scheme is energy stable at
    this step size), E2 growing from its start value, a complete record, and the same run cut into
    calls of 1000 steps (hat_U re-derived per call as in solver.py:159) agreeing to round-off."""
    N, nt = 4096, 5000
    p = make(N, nt, 'fast')
    s = chsimpy_amd.Solver(p)
    s.prepare()
    m0 = s.U_init.mean()
    sol = s.solve_or_resume()
    td = sol.timedata.data()
    assert td.shape == (nt, 9) and not np.any(np.isnan(td))
    assert np.array_equal(td[:, 0], np.arange(nt))
    assert sol.U.mean() == pytest.approx(m0, rel=1e-12)
    E = td[:, 1]
    assert np.all(np.diff(E) <= 1e-12 * np.abs(E[:-1]))        # total energy never grows
    assert td[-1, 2] > td[0, 2]                                  # the gradient energy has started to grow
    assert 0.0 < sol.U.min() and sol.U.max() < 1.0
    U1, td1 = sol.U.copy(), td.copy()
    s.close()
    s = chsimpy_amd.Solver(p)
    s.prepare()
    for _ in range(5):
        sol = s.solve_or_resume(1000)
    td2 = sol.timedata.data()
    assert td2.shape == td1.shape
    assert np.allclose(td2[:, 1:3], td1[:, 1:3], rtol=1e-9, atol=0)
    assert np.allclose(sol.U, U1, rtol=1e-9, atol=0), relerr(sol.U, U1)
    s.close()


def test_simulator_export_csv(gpu, tmp_path):
    p = make(64, 10)
    p.export_csv = 'U,E,E2,SA'
    p.file_id = str(tmp_path / 'run')
    sim = chsimpy_amd.Simulator(p)
    sol = sim.solve()
    base = sim.export()
    assert base == p.file_id + '.solution'
    from chsimpy_amd import utils
    assert np.array_equal(utils.csv_import_matrix(base + '.U.csv'), sol.U)
    assert np.array_equal(utils.csv_import_matrix(base + '.E2.csv'), sol.E2)
    assert utils.csv_import_matrix(base + '.E.csv').shape == (10,)


def test_ensemble_members_on_gpu(gpu, tmp_path):
    """Two Monte-Carlo members (experiment.py:84-126) through the GPU path vs the oracle with
    the same scaled A0/A1."""
    from chsimpy_amd import experiment as ex, utils
    p = make(128, 40)
    p.file_id = str(tmp_path / 'ens')
    ep = ex.ExperimentParams()
    ep.runs = 2
    recs = ex.run_ensemble(p, ep)
    rv, _, n = ex.make_rand_values(ep)
    assert n == 2 and len(recs) == 2
    for i, rec in enumerate(recs):
        f0, f1 = rv[i]
        o = orc.OracleSolver(orc.make_params(128, 40, func_A0=lambda T, f=f0: orc.A0(T) * f,
                                             func_A1=lambda T, f=f1: orc.A1(T) * f))
        o.prepare()
        o.solve_or_resume()
        assert rec[0] == pytest.approx(o.A0, rel=1e-15) and rec[1] == pytest.approx(o.A1, rel=1e-15)
        assert rec[8] == int(np.argmax(o.timedata.data()[:, 2])) and rec[9] == i
        assert rec[10] == f0 and rec[11] == f1
    df, agg = ex.write_results(p.file_id, recs)
    assert os.path.exists(p.file_id + '-results.csv')


# ---------------------------------------------------------------------------
# fp32 (BASELINE.json configs[3]: N=8192, fp32, adaptive_time).  The reference is float64 only
# (numpy defaults everywhere), so fp32 runs are validated against the fp64 path with a stated,
# looser tolerance: per-step U within rtol 2e-4 and E within rtol 1e-5 after 560 steps.
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("N", [128, 256, 512, 1024, 2048, 4096, 8192])
def test_fp32_dctn(gpu, N):
    p = make(N, 2, 'fast', dtype='float32')
    s = chsimpy_amd.Solver(p, np.full((N, N), 0.5))
    eng = s._get_engine()
    assert eng.engine == 'fast'
    rng = np.random.default_rng(3)
    X = rng.standard_normal((N, N)).astype(np.float32).astype(np.float64)
    Y = eng.dctn(X)
    T = scifft.dct(scifft.dct(X, axis=1, norm='ortho')[:, [0, 5, N - 1]], axis=0, norm='ortho')
    assert np.max(np.abs(Y[:, [0, 5, N - 1]] - T)) < 2e-5 * np.max(np.abs(T))
    assert np.sum(Y * Y) == pytest.approx(np.sum(X * X), rel=1e-5)
    Z = eng.dctn(Y, inverse=True)
    assert np.max(np.abs(Z - X)) < 2e-5 * np.max(np.abs(X))
    s.close()


@pytest.mark.parametrize("N,nt", [(128, 200), (512, 200), (2048, 40)])
def test_fp32_fast_engine_small_grids_vs_oracle(gpu, N, nt):
    """fp32 on the fast engine below N=4096 (the ensemble size N=2048 among them) against the fp64 oracle.
    The reference is float64 only: tolerance per step U rtol 2e-4, E rtol 1e-5 (as for the N=4096/8192
    fp32 runs); the record's counters are exact."""
    p = make(N, nt, 'fast', dtype='float32')
    s = chsimpy_amd.Solver(p)
    s.prepare()
    sol = s.solve_or_resume()
    assert s._engine.engine == 'fast'
    o = orc.OracleSolver(orc.make_params(N, nt))
    o.prepare()
    o.solve_or_resume()
    td, to = sol.timedata.data(), o.timedata.data()
    assert td.shape == to.shape and np.array_equal(td[:, 0], to[:, 0])
    assert np.allclose(sol.U, o.U, rtol=2e-4, atol=0), relerr(sol.U, o.U)
    assert np.allclose(td[:, 1], to[:, 1], rtol=1e-5, atol=0)          # E
    assert np.allclose(td[1:, 7], to[1:, 7], rtol=2e-3, atol=0)        # PS
    assert sol.U.mean() == pytest.approx(o.U.mean(), rel=2e-6)          # mass conservation in fp32
    s.close()


def test_fp64_n8192_steps_vs_oracle(gpu):
    """fp64 at N=8192 (four wavefronts per transform): a few steps against the oracle."""
    compare_run(make(8192, 4, 'fast'), {})


@pytest.mark.parametrize("N,dmax,dtype", [(2048, 2.4e-10, 'float64'), (4096, 1.2e-10, 'float64'),
                                           (4096, 1.2e-10, 'float32'), (8192, 6e-11, 'float32')])
def test_fused_adaptive_column_sums_match_the_sweep(gpu, N, dmax, dtype):
    """Adaptive dt on the fast engine: the fused row kernel adds up the integrand of solver.py:183 per
    column itself (no sweep of U, U not written between steps).  CHS_ADAPT_SWEEP=1 keeps the
    separate sweep kernel (the path the N=64 golden test and the oracle pin): same delt history."""
    nt = 540
    runs = {}
    for mode in ('sweep', 'fused'):
        if mode == 'sweep':
            os.environ['CHS_ADAPT_SWEEP'] = '1'
        else:
            os.environ.pop('CHS_ADAPT_SWEEP', None)
        try:
            p = make(N, nt, 'fast', adaptive_time=True, delt_max=dmax, dtype=dtype)
            s = chsimpy_amd.Solver(p)
            s.prepare()
            sol = s.solve_or_resume(300)      # chunked: the first step of a call sweeps U in both modes
            sol = s.solve_or_resume(nt - 300)
            runs[mode] = (sol.U.copy(), sol.timedata.data().copy())
            s.close()
        finally:
            os.environ.pop('CHS_ADAPT_SWEEP', None)
    Us, ts = runs['sweep']
    Uf, tf = runs['fused']
    assert ts.shape == tf.shape == (nt, 9)
    assert len(np.unique(ts[:, 8])) > 3                        # the step size adapts after step 500
    # fp32: the sweep recomputes mu from the stored field with another (equally rounded) formula than
    # the fused kernel's shared-log one, so the two agree to fp32 rounding amplified over the steps
    tol = 1e-9 if dtype == 'float64' else 2e-4
    assert np.allclose(tf[:, 8], ts[:, 8], rtol=1e-12 if dtype == 'float64' else 1e-4, atol=0), relerr(tf[:, 8], ts[:, 8])   # delt history
    # E2 of these early steps is the start noise growing exponentially (1e-19): in fp32 both paths sit 6.6e-4
    # (fused) and 8.9e-4 (sweep) from the fp64 run at N=8192 and 2.4e-4 from each other (tools/f32_paths_n8192.py)
    assert np.allclose(tf[:, 1:3], ts[:, 1:3], rtol=tol if dtype == 'float64' else 1e-3, atol=0)
    assert np.allclose(Uf, Us, rtol=tol, atol=0), relerr(Uf, Us)


def test_fp32_adaptive_vs_fp64_n4096(gpu):
    # The reference's dynamic step is a column SUM (np.linalg.norm(.., ord=-1), solver.py:183), so it
    # grows with N: with the default delt_max the step explodes at N=4096 (NaN at step 506, which the
    # engine reports as the reference would).  delt_max is chosen so that the dynamic step is ~2x delt.
    nt = 560
    runs = {}
    for dt in ('float64', 'float32'):
        p = make(4096, nt, 'fast', dtype=dt, adaptive_time=True, delt_max=1.2e-10)
        s = chsimpy_amd.Solver(p)
        s.prepare()
        sol = s.solve_or_resume()
        runs[dt] = (sol.U.copy(), sol.timedata.data().copy())
        s.close()
    U64, t64 = runs['float64']
    U32, t32 = runs['float32']
    assert t32.shape == t64.shape == (nt, 9)
    assert len(np.unique(t64[:, 8])) > 3                       # the step size adapts after step 500
    assert np.allclose(t32[:, 8], t64[:, 8], rtol=2e-3)        # same delt history
    assert np.allclose(U32, U64, rtol=2e-4, atol=0), relerr(U32, U64)
    assert np.allclose(t32[:, 1], t64[:, 1], rtol=1e-5)        # E
    assert np.allclose(t32[:, 7], t64[:, 7], rtol=1e-3)        # PS


def test_fp32_n8192_adaptive_runs(gpu):
    """configs[3] itself: N=8192, fp32, adaptive_time -- size-independent properties."""
    N = 8192
    p = make(N, 506, 'fast', dtype='float32', adaptive_time=True, delt_max=6e-11)
    s = chsimpy_amd.Solver(p)
    s.prepare()
    sol = s.solve_or_resume()
    td = sol.timedata.data()
    assert td.shape == (506, 9) and not np.any(np.isnan(td))
    assert len(np.unique(td[:, 8])) >= 3                        # adaptive dt fired (steps 502, 504)
    assert sol.U.mean() == pytest.approx(s.U_init.mean(), rel=2e-6)   # mass conservation in fp32
    assert 0.8 < sol.U.min() and sol.U.max() < 0.95
    assert np.all(td[501:, 8] >= td[500, 8])                    # the step only grows here
    s.close()


def test_fp32_n8192_adaptive_steps_vs_oracle_seeded_at_step_499(gpu):
    """configs[3] against the ORACLE: the adaptive branch (solver.py:177-193) only fires beyond step 500, and 500
    oracle steps at N=8192 would take an hour.  Both codes are therefore seeded at computed_steps = 499 from the
    same start field -- the engine through chs_set_state, the oracle by its attribute -- and run 8 steps: records
    499..506, delt re-evaluated at steps 502, 504 and 506 from the min column sum of the integrand over a
    8192 x 8192 fp32 field (the fused row kernel's column partials, the two-stage reduction, lam1/lam2 regenerated
    on the device, the gated tail).  fp32 against the fp64 oracle: delt 1e-4, E 1e-5, U 2e-4."""
    N, steps, dmax = 8192, 8, 6e-11
    kw = dict(adaptive_time=True, delt_max=dmax)
    p = make(N, 10 ** 6, 'fast', dtype='float32', **kw)
    s = chsimpy_amd.Solver(p)
    s.prepare()
    eng = s._engine
    st = eng.get_state()
    st.computed_steps = 499
    st.skip_check = 1          # (the energy rule indexes the record by step number, timedata.py:63: not with a seeded counter)
    eng.set_state(st)
    rows, rc = eng.step_n(steps)
    assert rc == 0 and rows.shape == (steps, 9)
    U = eng.get_U()
    o = orc.OracleSolver(orc.make_params(N, 10 ** 6, **kw))
    o.prepare()
    o.computed_steps = 499
    o.skip_check = True
    o.solve_or_resume(steps)
    to = o.timedata.data()[1:]
    assert np.array_equal(rows[:, 0], to[:, 0]) and rows[0, 0] == 499 and rows[-1, 0] == 506
    assert len(np.unique(to[:, 8])) == 4                       # delt: the seed value, then three re-evaluations
    errs = {c: relerr(rows[:, c], to[:, c]) for c in (1, 2, 5, 6, 7, 8)}
    log_line(f"N=8192 fp32 adaptive, seeded at step 499, 8 steps vs fp64 oracle: delt {errs[8]:.3e} E {errs[1]:.3e} "
             f"E2 {errs[2]:.3e} Ra {errs[5]:.3e} L2 {errs[6]:.3e} PS {errs[7]:.3e} U {relerr(U, o.U):.3e}")
    assert errs[8] < 1e-4, errs                                # the delt history (solver.py:183-188)
    assert errs[1] < 1e-5 and errs[7] < 2e-3, errs             # E, PS
    # E2 (a difference quotient of an fp32 field), Ra (one row's mean absolute deviation) and L2 = ||mu||/N^2 in fp32
    # against the fp64 oracle -- measured 1.1e-4 / 4.0e-5 / 7.9e-6 (profiles/r03_parity_margins.txt); stated
    # tolerances a factor ~5-10 above that
    assert errs[2] < 1e-3 and errs[5] < 4e-4 and errs[6] < 1e-4, errs
    assert np.allclose(rows[:, 4], to[:, 4], rtol=1e-4)        # domtime = (sum delt / M_tilde)^(1/3)
    assert np.allclose(U, o.U, rtol=2e-4, atol=0), relerr(U, o.U)
    s.close(fetch_U=False)


def test_n4096_gated_stop_path_vs_oracle_time_limit_with_energy_rule_armed(gpu):
    """The reference's default mode (full_sim=False, parameters.py:50) plus a time limit (solver.py:197-199) at the
    HEADLINE size.  At N >= 4096 the stop rules do not use the two-buffer hat_U of the small grids but the gate:
    the bookkeeping of step s rides in k_col(s+1), whose tile workgroups wait for its decision in front of their
    first write (gate_wait); a stop leaves hat_U as step s left it and run_steps rebuilds U = idctn(hat_U)
    (chs_fast_recover_u) -- a code path no test below N=4096 runs.  time_max is chosen so that the limit ends the
    run after 8 completed steps (about 1.4 s per oracle step on one core).  Chunks: 3 and 3 steps through the gate
    without a stop, then a call of 10 that the limit ends after 2 (gate -> halt -> rebuilt U), then a resumed
    call that completes nothing but still advances time_delta_sum by one delt (solver.py:195 runs before the
    check).  U, the record, counters and time bookkeeping against the oracle after every call, rtol 1e-9."""
    N = 4096
    tmax = 8.5 * 3e-8 / 1.71e-8 / 60          # minutes of simulated time = 8.5 steps
    kw = dict(full_sim=False, time_max=tmax)
    s = chsimpy_amd.Solver(make(N, 10 ** 6, 'fast', **kw))
    s.rederive_hat = True                      # the literal solver.py:159 at every call, as the oracle does
    o = orc.OracleSolver(orc.make_params(N, 10 ** 6, **kw))
    s.prepare(); o.prepare()
    expect = {4: (4, 'None'), 3: (7, 'None'), 10: (9, 'time-limit'), 2: (9, 'time-limit')}
    for chunk in (4, 3, 10, 2):
        sol = s.solve_or_resume(chunk)
        o.solve_or_resume(chunk)
        assert (sol.computed_steps, sol.stop_reason) == (o.computed_steps, o.stop_reason) == expect[chunk], chunk
        eu = relerr(sol.U, o.U)
        log_line(f"N=4096 gated stop path (full_sim=False + time limit), after chunk {chunk}: steps {sol.computed_steps} "
                 f"stop {sol.stop_reason} U {eu:.3e}")
        assert eu < RTOL, (chunk, eu)
        assert s.time_delta_sum == pytest.approx(o.time_delta_sum, rel=1e-13)
        assert s.time_passed == pytest.approx(o.time_passed, rel=1e-13)
    td, to = sol.timedata.data(), o.timedata.data()
    assert td.shape == to.shape == (9, 9)
    for c in range(1, 9):
        assert np.allclose(td[:, c], to[:, c], rtol=RTOL, atol=1e-300), (c, relerr(td[:, c], to[:, c]))
    assert sol.tau0 == o.tau0 and sol.t0 == o.t0
    s.close(fetch_U=False)


def test_n4096_fp64_adaptive_steps_vs_oracle_seeded_at_step_499(gpu):
    """adaptive_time (solver.py:177-193) in fp64 at the headline size against the ORACLE: the branch fires beyond
    step 500 only, and 500 oracle steps at N=4096 take twelve minutes -- so both codes are seeded at
    computed_steps = 499 from the same start field (chs_set_state / the oracle's attribute), as the N=8192 fp32 test
    does, and run 8 steps: records 499..506, delt re-evaluated at steps 502, 504, 506 from the fused row kernel's
    column partials, the two-stage reduction, lam1/lam2 regenerated on the device behind the gate.
    delt / E / E2 / U at rtol 1e-9."""
    N, steps, dmax = 4096, 8, 1.2e-10
    kw = dict(adaptive_time=True, delt_max=dmax)
    s = chsimpy_amd.Solver(make(N, 10 ** 6, 'fast', **kw))
    s.prepare()
    eng = s._engine
    st = eng.get_state()
    st.computed_steps = 499
    st.skip_check = 1          # (the energy rule indexes the record by step number, timedata.py:63: not with a seeded counter)
    eng.set_state(st)
    rows, rc = eng.step_n(steps)
    assert rc == 0 and rows.shape == (steps, 9)
    U = eng.get_U()
    o = orc.OracleSolver(orc.make_params(N, 10 ** 6, **kw))
    o.prepare()
    o.computed_steps = 499
    o.skip_check = True
    o.solve_or_resume(steps)
    to = o.timedata.data()[1:]
    assert np.array_equal(rows[:, 0], to[:, 0]) and rows[0, 0] == 499 and rows[-1, 0] == 506
    assert len(np.unique(to[:, 8])) == 4                       # delt: the seed value, then three re-evaluations
    assert to[-1, 8] > 2.0 * to[0, 8]                          # ... that did move it
    errs = {c: relerr(rows[:, c], to[:, c]) for c in (1, 2, 4, 5, 6, 7, 8)}
    eu = relerr(U, o.U)
    log_line(f"N=4096 fp64 adaptive, seeded at step 499, 8 steps vs oracle: delt {errs[8]:.3e} E {errs[1]:.3e} "
             f"E2 {errs[2]:.3e} Ra {errs[5]:.3e} L2 {errs[6]:.3e} PS {errs[7]:.3e} U {eu:.3e}")
    for c in (1, 2, 4, 5, 6, 7, 8):
        assert errs[c] < RTOL, errs
    assert eu < RTOL, eu
    s.close(fetch_U=False)


def test_adaptive_default_delt_max_blows_up_at_large_n_like_the_reference_quirk(gpu):
    """delt_dyn = min column SUM scales with N: at N=4096 the default
    delt_max makes the step jump by ~800x at step 502 and U leaves (0,1) -> the NaN assertion."""
    p = make(4096, 520, 'fast', adaptive_time=True)
    s = chsimpy_amd.Solver(p)
    s.prepare()
    with pytest.raises(AssertionError):
        s.solve_or_resume()
    assert 502 <= s.solution.computed_steps <= 510
    assert s.solution.delt[-1] > 1e-6
    s.close()
