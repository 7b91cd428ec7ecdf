"""GPU parity tests (``-m gpu``) of the run modes on the FAST transform engine (every N the benchmarks
use): the reference's default mode -- the energy stop rule with ``full_sim=False`` (parameters.py:50,
solver.py:242-251) --, the same rule with ``full_sim=True`` (deferred bookkeeping), ``adaptive_time``
(solver.py:177-193) in one call and in chunks, and the batched issue of a long call.  Everything
against ``OracleSolver`` on the same inputs.
"""
import os
import time

import numpy as np
import pytest

import chsimpy_amd
from oracle import chs_oracle as orc
from gpu_helpers import KAPPA, RTOL, compare_run, log_line, make, relerr

pytestmark = pytest.mark.gpu

# (N, delt, step at which timedata.energy_falls first holds on the default start field; oracle runs)
STOPS = [(128, 1e-6, 113), (256, 1.8e-7, 682)]

# These runs go through the whole spinodal growth up to the E2 maximum with a large time step: round-off
# differences between two correct transform implementations grow with the unstable modes, hence rtol 1e-8
# (as in the N=64 energy-stop tests) instead of the 1e-9 of the plain runs.  L2 = ||mu||_F / N^2
# (solver.py:225) is worse conditioned than U itself: mu ~ 0.24 is the sum of terms of size ~100
# (RT ln(U/(1-U)) - BRT + ...), so a relative error eps in U shows up as ~400 eps in L2.
# Conditioning of the chosen runs, measured on the oracle itself (one-ulp perturbation of U_init -> change
# of the final U): N=128/delt=1e-6: 1.2e-9, N=256/delt=1.8e-7: 4.5e-10 (N=256/delt=2.5e-7 would be
# 4e-7: too close to the stability limit of the explicit nonlinear term to compare two codes at 1e-8).
RTOL_LONG = 1e-8
RTOL_LONG_L2 = 400 * RTOL_LONG


def close_long(td, to, what=''):
    assert td.shape == to.shape
    errs = [relerr(td[:, c], to[:, c]) for c in range(9)]
    log_line(f"{what}: rows {td.shape[0]} max rel err E={errs[1]:.3e} E2={errs[2]:.3e} Ra={errs[5]:.3e} L2={errs[6]:.3e} "
             f"PS={errs[7]:.3e} delt={errs[8]:.3e}")
    for c in range(9):
        assert errs[c] <= (RTOL_LONG_L2 if c == 6 else RTOL_LONG), (c, errs)


@pytest.mark.parametrize("N,delt,stop", STOPS)
def test_energy_stop_full_sim_false_fast_engine(gpu, N, delt, stop):
    """full_sim=False on the fused pipeline: the tail runs in stream order, raises `halt` at the step
    where E2[it-1] > E2[it] > E2[0] (timedata.py:63) and the field of that step is rebuilt from hat_U
    (chs_fast_recover_u): same stop step, tau0, t0, stop_reason, U and record as the oracle."""
    kw = dict(full_sim=False, delt=delt)
    p = make(N, 6000, 'fast', **kw)
    s = chsimpy_amd.Solver(p)
    o = orc.OracleSolver(orc.make_params(N, 6000, **kw))
    s.prepare(); o.prepare()
    sol = s.solve_or_resume(); o.solve_or_resume()
    assert s._engine.engine == 'fast'
    assert sol.stop_reason == o.stop_reason == 'energy'
    assert sol.computed_steps == o.computed_steps == stop and sol.tau0 == o.tau0 == stop
    assert sol.t0 == pytest.approx(o.t0, rel=1e-12)
    close_long(sol.timedata.data(), o.timedata.data(), f"energy stop full_sim=False fast N={N} delt={delt} (stop {stop}, U {relerr(sol.U, o.U):.3e})")
    assert np.allclose(sol.U, o.U, rtol=RTOL_LONG, atol=0), relerr(sol.U, o.U)
    # resume after the stop (hat_U re-derived from the rebuilt U, solver.py:159)
    for chunk in (1, 5, 30):
        sol = s.solve_or_resume(chunk); o.solve_or_resume(chunk)
        assert sol.computed_steps == o.computed_steps
        assert sol.stop_reason == o.stop_reason and sol.tau0 == o.tau0
        assert np.allclose(sol.U, o.U, rtol=1e-8, atol=0), (chunk, relerr(sol.U, o.U))
    close_long(sol.timedata.data(), o.timedata.data())
    s.close()


@pytest.mark.parametrize("N,delt,stop", STOPS)
def test_energy_rule_full_sim_true_fast_engine(gpu, N, delt, stop):
    """full_sim=True through the E2 maximum: the bookkeeping of step s rides in k_col of step s+1
    (deferred tail) and must record tau0/t0 once and set skip_check (solver.py:242-249)."""
    kw = dict(full_sim=True, delt=delt)
    nt = stop + 40
    s = chsimpy_amd.Solver(make(N, nt, 'fast', **kw))
    o = orc.OracleSolver(orc.make_params(N, nt, **kw))
    s.prepare(); o.prepare()
    sol = s.solve_or_resume(); o.solve_or_resume()
    assert sol.stop_reason == o.stop_reason == 'None' and sol.computed_steps == o.computed_steps == nt
    assert sol.tau0 == o.tau0 == stop and sol.t0 == pytest.approx(o.t0, rel=1e-12)
    assert s.skip_check and o.skip_check
    close_long(sol.timedata.data(), o.timedata.data(), f"energy rule full_sim=True fast N={N} delt={delt} (tau0 {stop}, U {relerr(sol.U, o.U):.3e})")
    assert np.allclose(sol.U, o.U, rtol=RTOL_LONG, atol=0), relerr(sol.U, o.U)
    s.close()
    # ... and cut into calls around the maximum: skip_check survives the calls (solver.py:50,249)
    s = chsimpy_amd.Solver(make(N, nt, 'fast', **kw))
    o = orc.OracleSolver(orc.make_params(N, nt, **kw))
    s.prepare(); o.prepare()
    for chunk in (stop - 3, 2, 1, 1, 1, 38):
        sol = s.solve_or_resume(chunk); o.solve_or_resume(chunk)
        assert (sol.computed_steps, sol.tau0) == (o.computed_steps, o.tau0)
        assert s.skip_check == o.skip_check
    assert sol.t0 == pytest.approx(o.t0, rel=1e-12)
    close_long(sol.timedata.data(), o.timedata.data())
    assert np.allclose(sol.U, o.U, rtol=RTOL_LONG, atol=0), relerr(sol.U, o.U)
    s.close()


@pytest.mark.parametrize("N,chunks", [(128, (600,)), (128, (300, 300)), (128, (300, 221, 7, 72)),
                                      (256, (600,)), (256, (300, 221, 7, 72)),
                                      (1024, (520, 30, 50))])
def test_adaptive_time_fast_engine(gpu, N, chunks):
    """adaptive_time on the fused pipeline (integrand column sums added up inside the row kernel,
    k_colmin_rows, lam1/lam2 regenerated on the device -- N >= 1024; the smaller grids sweep U with k_mu)
    against the oracle: the delt history to 1e-9, U/E/E2 to 1e-8 -- in one call and in chunks.  A chunk that starts beyond step 500 exercises the
    reference's resume quirk: the call reloads the grids of params.delt (solver.py:154-155) while
    self.delt keeps its adapted value until the next even step re-evaluates it (185-193).
    delt_dyn is a column SUM (np.linalg.norm(.., ord=-1)) and grows with N: delt_max is scaled so that
    the step settles at a few times params.delt instead of blowing up."""
    kw = dict(adaptive_time=True, delt_max=4.9e-7 / N)
    s = chsimpy_amd.Solver(make(N, 600, 'fast', **kw))
    o = orc.OracleSolver(orc.make_params(N, 600, **kw))
    s.prepare(); o.prepare()
    for c in chunks:
        sol = s.solve_or_resume(c); o.solve_or_resume(c)
        assert sol.computed_steps == o.computed_steps
    assert s._engine.engine == 'fast'
    td, to = sol.timedata.data(), o.timedata.data()
    assert td.shape == to.shape == (600, 9)
    assert len(np.unique(to[:, 8])) > 20                         # the step really adapts
    assert np.allclose(td[:, 8], to[:, 8], rtol=1e-9, atol=0), relerr(td[:, 8], to[:, 8])
    for c in (1, 2, 4, 5, 6, 7):
        assert np.allclose(td[:, c], to[:, c], rtol=1e-8, atol=1e-300), (c, relerr(td[:, c], to[:, c]))
    log_line(f"adaptive fast N={N} chunks={chunks}: max rel err delt={relerr(td[:, 8], to[:, 8]):.3e} E={relerr(td[:, 1], to[:, 1]):.3e} "
             f"E2={relerr(td[:, 2], to[:, 2]):.3e} U={relerr(sol.U, o.U):.3e}")
    assert np.array_equal(td[:, 3], to[:, 3])                    # SA: a count
    assert np.allclose(sol.U, o.U, rtol=1e-8, atol=0), relerr(sol.U, o.U)
    assert s.delt == pytest.approx(o.delt, rel=1e-9) and s.time_passed == pytest.approx(o.time_passed, rel=1e-9)
    s.close()


def test_adaptive_resume_quirk_direct_engine(gpu):
    """The same resume quirk on the direct engine (N=64, the reference's default delt_max)."""
    kw = dict(adaptive_time=True)
    s = chsimpy_amd.Solver(make(64, 600, 'direct', **kw))
    o = orc.OracleSolver(orc.make_params(64, 600, **kw))
    s.prepare(); o.prepare()
    for c in (520, 1, 2, 77):
        sol = s.solve_or_resume(c); o.solve_or_resume(c)
    td, to = sol.timedata.data(), o.timedata.data()
    assert np.allclose(td[:, 8], to[:, 8], rtol=1e-9, atol=0), relerr(td[:, 8], to[:, 8])
    assert np.allclose(td[:, 1:3], to[:, 1:3], rtol=1e-8, atol=0)
    assert np.allclose(sol.U, o.U, rtol=1e-8, atol=0), relerr(sol.U, o.U)
    s.close()


@pytest.mark.parametrize("engine,N,delt,stop", [('fast', 128, 1e-6, 113), ('direct', 64, 3e-6, 72)])
def test_batched_issue_stops_behind_the_halt_flag(gpu, engine, N, delt, stop, monkeypatch):
    """chs_step_n issues its launches in batches and looks at the device's stop flag between them
    (CHS_BATCH_STEPS, a test hook, makes the batches small): same rows, same field as one batch, and
    the reference's defaults -- ntmax = 1e6, full_sim = False (parameters.py:42,50) -- return right
    after the stop instead of queueing a million empty steps."""
    runs = {}
    for bs in ('7', None):
        if bs:
            monkeypatch.setenv('CHS_BATCH_STEPS', bs)
        else:
            monkeypatch.delenv('CHS_BATCH_STEPS', raising=False)
        p = make(N, int(1e6), engine, full_sim=False, delt=delt)
        s = chsimpy_amd.Solver(p)
        s.prepare()
        t0 = time.time()
        sol = s.solve_or_resume()
        dt = time.time() - t0
        assert sol.stop_reason == 'energy' and sol.computed_steps == stop
        assert dt < 5.0, dt
        runs[bs] = (sol.U.copy(), sol.timedata.data().copy())
        s.close()
    assert np.array_equal(runs['7'][1], runs[None][1]) and np.array_equal(runs['7'][0], runs[None][0])
    # a long full_sim call in small batches: the rows come out of the device ring batch by batch
    monkeypatch.setenv('CHS_BATCH_STEPS', '16')
    sol, o = compare_run(make(N, 150, engine), {})
    assert sol.timedata.data().shape == (150, 9)


def test_field_assigned_between_calls_is_picked_up(gpu):
    """`U = self.solution.U` (solver.py:158): a caller that replaces solution.U between two calls
    continues from that field."""
    N = 128
    s = chsimpy_amd.Solver(make(N, 0, 'fast'))
    o = orc.OracleSolver(orc.make_params(N, 0))
    s.prepare(); o.prepare()
    sol = s.solve_or_resume(6); o.solve_or_resume(6)
    V = np.clip(sol.U[::-1, :].copy() * 1.0001, 0.8, 0.95)
    sol.U = V
    o.U = V.copy()
    sol = s.solve_or_resume(5); o.solve_or_resume(5)
    assert np.allclose(sol.U, o.U, rtol=RTOL, atol=0), relerr(sol.U, o.U)
    assert np.allclose(sol.timedata.data(), o.timedata.data(), rtol=RTOL, atol=1e-300)
    s.close()


@pytest.mark.parametrize("N", [128, 1024])
@pytest.mark.parametrize("full_sim", [True, False])
def test_calls_continue_the_device_loop(gpu, N, full_sim):
    """Fixed time step: a call that follows a completed call continues the device loop where it stopped
    (hat_U, the row transform of EnergieEut(U) and its sum of squares stay on the device; no dctn(U) on
    entry, solver.py:159).  A chunked run then gives bit for bit what one call gives, and both stay within
    tolerance of the oracle's chunks, which recompute hat_U = dctn(U) per call; `rederive_hat` asks the
    device for that literal recomputation."""
    chunks = (5, 1, 17, 2, 30)
    nt = sum(chunks)
    one = chsimpy_amd.Solver(make(N, nt, 'fast', full_sim=full_sim))
    one.prepare()
    sol1 = one.solve_or_resume(nt)
    U1, td1 = sol1.U.copy(), sol1.timedata.data().copy()
    one.close()
    runs = {}
    # rederive 'full': the literal call with NOTHING taken over: every call enters through k_row_fwd2 (hat_U = dctn(U) and the
    # row transform of EnergieEut(U) from one sweep of U) -- the default of rederive_hat;
    # rederive True: hat_U = dctn(U) recomputed, the first step's operand taken over from the previous call's last step
    # (CHS_STEP_KEEP_T1, Solver.rederive_keeps_t1)
    for rederive in (False, True, 'full'):
        s = chsimpy_amd.Solver(make(N, nt, 'fast', full_sim=full_sim))
        s.rederive_hat = bool(rederive)
        s.rederive_keeps_t1 = (rederive is True)
        s.prepare()
        o = orc.OracleSolver(orc.make_params(N, nt, full_sim=full_sim)) if N <= 128 else None
        if o:
            o.prepare()
        for c in chunks:
            sol = s.solve_or_resume(c)
            if o:
                o.solve_or_resume(c)
                assert np.allclose(sol.U, o.U, rtol=RTOL, atol=0), (rederive, c, relerr(sol.U, o.U))
        runs[rederive] = (sol.U.copy(), sol.timedata.data().copy())
        if o:
            assert np.allclose(sol.timedata.data(), o.timedata.data(), rtol=RTOL, atol=1e-300)
            # the call that reached ntmax did not prepare a continuation (CHS_STEP_LAST_CALL): one more call
            # enters through hat_U = dctn(U) again
            sol = s.solve_or_resume(3); o.solve_or_resume(3)
            assert np.allclose(sol.U, o.U, rtol=RTOL, atol=0), relerr(sol.U, o.U)
            assert np.allclose(sol.timedata.data(), o.timedata.data(), rtol=RTOL, atol=1e-300)
        s.close()
    assert np.array_equal(runs[False][0], U1) and np.array_equal(runs[False][1], td1)
    # the literal recomputation differs from the carried array by rounding only
    assert np.allclose(runs[True][0], U1, rtol=1e-11, atol=0), relerr(runs[True][0], U1)
    assert not np.array_equal(runs[True][0], U1)
    # ... and taking the first step's operand over (a function of the unchanged field) changes nothing at all
    assert np.array_equal(runs[True][0], runs['full'][0]) and np.array_equal(runs[True][1], runs['full'][1]), \
        relerr(runs[True][0], runs['full'][0])
    log_line(f"N={N} full_sim={full_sim} chunks {chunks}: carried hat_U == one call bit for bit; "
             f"rederive_hat vs carried max rel err U={relerr(runs[True][0], U1):.3e}")


@pytest.mark.parametrize("full_sim,seed", [(True, 1), (False, 2), (True, 3), (False, 4), (True, 5), (False, 6)])
def test_interleaved_engine_calls_keep_the_loop_state_right(gpu, full_sim, seed):
    """The device keeps hat_U, the row transform of EnergieEut(U) and its sum of squares between calls and the
    last call's state for chs_get_state; everything that can invalidate them is thrown in between the chunks in a
    seeded random order: field downloads, a field assigned by the caller, the scratch-using entry points
    (chs_dctn, chs_get_mu), profiled steps' absence aside.  Every chunk is compared with the oracle doing the same."""
    N = 128
    rng = np.random.default_rng(seed)
    s = chsimpy_amd.Solver(make(N, 10 ** 6, 'fast', full_sim=full_sim))
    o = orc.OracleSolver(orc.make_params(N, 10 ** 6, full_sim=full_sim))
    s.prepare(); o.prepare()
    eng = s._engine
    done = 0
    ops = []
    while done < 140:
        op = rng.choice(['step', 'step', 'step', 'step', 'getU', 'setU', 'dctn', 'mu', 'state', 'toggle'])
        ops.append(op)
        if op == 'step':
            n = int(rng.integers(1, 9))
            sol = s.solve_or_resume(n); o.solve_or_resume(n)
            done += n
            assert sol.computed_steps == o.computed_steps, ops
            assert np.allclose(sol.timedata.data(), o.timedata.data(), rtol=RTOL, atol=1e-300), ops
        elif op == 'getU':
            assert np.allclose(s.solution.U, o.U, rtol=RTOL, atol=0), ops
        elif op == 'setU':
            V = np.clip(o.U * (1.0 + 1e-4 * rng.standard_normal((N, N))), 0.7, 0.97)
            s.solution.U = V
            o.U = V.copy()
        elif op == 'dctn':
            X = rng.random((N, N))
            from scipy.fft import dctn
            assert np.allclose(eng.dctn(X), dctn(X, norm='ortho'), rtol=1e-11, atol=1e-12), ops
        elif op == 'mu':
            eng.get_mu()
        elif op == 'toggle':
            s.rederive_hat = not s.rederive_hat
        else:
            st = eng.get_state()
            assert st.computed_steps == o.computed_steps and st.time_passed == pytest.approx(o.time_passed, rel=1e-12), ops
    assert np.allclose(s.solution.U, o.U, rtol=RTOL, atol=0), (ops, relerr(s.solution.U, o.U))
    s.close()


def test_pooled_engine_equals_fresh_engine(gpu, monkeypatch):
    """chs_destroy parks engines and chs_create re-arms a parked one of the same size for the next run (an ensemble
    creates one engine per member): a run on a re-armed engine -- other constants, other mode, other generator
    seed than the run that parked it -- is bit for bit the run on a freshly created engine."""
    N = 256

    def run(kw, seed):
        p = make(N, 40, 'fast', **kw)
        p.seed = seed
        s = chsimpy_amd.Solver(p)
        s.prepare()
        s.solve_or_resume(25)
        sol = s.solve_or_resume(15)
        out = (sol.U.copy(), sol.timedata.data().copy())
        s.close()
        return out
    first = dict(full_sim=False, adaptive_time=True, delt_max=3e-9)
    second = dict(full_sim=True, kappa_tilde=KAPPA * 1.3, delt=2e-11)
    monkeypatch.setenv('CHS_ENGINE_POOL', '0')
    fresh = run(second, 7)
    monkeypatch.delenv('CHS_ENGINE_POOL')
    run(first, 3)            # parks its engine on close
    pooled = run(second, 7)  # ... which this run takes over
    assert np.array_equal(pooled[0], fresh[0]) and np.array_equal(pooled[1], fresh[1])


def test_simulator_update_every_drives_chunks(gpu, tmp_path):
    """Chunked driving (simulator.py:56-87): `update_every` steps per solve_or_resume, a host snapshot
    handed to the view hook after every chunk, the last chunk shortened to hit ntmax, and the
    tau0/t0 fix-up when the energy rule never fired (simulator.py:84-86)."""
    N, nt, every = 128, 47, 10
    seen = []
    p = make(N, nt, 'fast')
    p.update_every = every
    p.file_id = str(tmp_path / 'chunks')
    sim = chsimpy_amd.Simulator(p, on_update=lambda sm: seen.append((sm.solver.solution.computed_steps,
                                                                     sm.solver.solution.U.copy())))
    sol = sim.solve()
    o = orc.OracleSolver(orc.make_params(N, nt))
    o.prepare()
    steps, snaps = [], []
    left = nt
    while left > 0:
        c = min(every, left)
        o.solve_or_resume(c)
        steps.append(o.computed_steps); snaps.append(o.U.copy())
        left -= c
    assert [c for c, _ in seen] == steps
    for (_, Ug), Uo in zip(seen, snaps):
        assert np.allclose(Ug, Uo, rtol=RTOL, atol=0)
    assert sim.steps_total == nt
    assert np.allclose(sol.timedata.data(), o.timedata.data(), rtol=RTOL, atol=1e-300)
    assert sol.tau0 == sol.computed_steps - 1 and sol.t0 == pytest.approx(o.time_passed, rel=1e-12)


def test_uinit_file_import(gpu, tmp_path):
    """`--Uinit-file` (simulator.py:21-22): the start field comes from a CSV written by
    utils.csv_export_matrix (utils.py:79-83)."""
    from chsimpy_amd import utils
    N = 128
    rng = np.random.default_rng(5)
    U0 = 0.875 + 0.004 * (rng.random((N, N)) - 0.5)
    f = str(tmp_path / 'u0.csv')
    utils.csv_export_matrix(U0, f)
    p = make(N, 12, 'fast')
    p.Uinit_file = f
    p.file_id = str(tmp_path / 'fromfile')
    sim = chsimpy_amd.Simulator(p)
    sol = sim.solve()
    U0r = utils.csv_import_matrix(f)
    o = orc.OracleSolver(orc.make_params(N, 12), U0r)
    o.prepare(); o.solve_or_resume()
    assert np.allclose(sol.U, o.U, rtol=RTOL, atol=0)
    assert np.allclose(sol.timedata.data(), o.timedata.data(), rtol=RTOL, atol=1e-300)
    sim.solver.close()


def test_reference_default_configuration_runs_to_its_energy_stop(gpu):
    """`Parameters()` exactly as chsimpy ships them (parameters.py:24-61: N=512, ntmax=1e6, full_sim=False;
    only kappa_tilde is passed explicitly) through `Simulator.solve()`: the run ends at the
    E2 maximum (step 1674) -- same stop step, tau0, t0, record and field as the oracle, and the million-step
    call returns as soon as the device has stopped."""
    p = chsimpy_amd.Parameters()
    p.kappa_tilde, p.no_gui = KAPPA, True
    assert (p.N, p.ntmax, p.full_sim, p.adaptive_time, p.jitter) == (512, int(1e6), False, False, None)
    sim = chsimpy_amd.Simulator(p)
    t0 = time.time()
    sol = sim.solve()
    dt = time.time() - t0
    o = orc.OracleSolver(orc.OracleParams(N=512))
    o.prepare()
    o.solve_or_resume()
    assert sim.solver._engine.engine == 'fast'
    assert sol.stop_reason == o.stop_reason == 'energy'
    assert sol.computed_steps == o.computed_steps == 1674 and sol.tau0 == o.tau0 == 1674
    assert sol.t0 == pytest.approx(o.t0, rel=1e-12)
    td, to = sol.timedata.data(), o.timedata.data()
    assert td.shape == to.shape == (1674, 9)
    assert np.allclose(td, to, rtol=RTOL, atol=1e-300), [relerr(td[:, c], to[:, c]) for c in range(1, 9)]
    assert np.allclose(sol.U, o.U, rtol=RTOL, atol=0), relerr(sol.U, o.U)
    log_line(f"reference default run N=512 (energy stop at 1674): U={relerr(sol.U, o.U):.3e} E2={relerr(td[:, 2], to[:, 2]):.3e} "
             f"GPU wall {dt:.2f} s")
    assert dt < 10.0
    sim.solver.close()


@pytest.mark.parametrize("N,dtype,tol", [(1024, 'float64', 1e-9), (2048, 'float32', 2e-4)])
def test_adaptive_chunks_of_every_alignment_seeded_at_step_499(gpu, N, dtype, tol):
    """The adaptive path issues its step-size machinery only on the steps whose rule fires (every second one beyond
    step 500, solver.py:177) -- the host follows the device's step counter to know which (Engine::csHost: prepare /
    chs_set_state / end of a call, +1 per issued step) -- and on those steps the reduction's last block sets the coming
    step's coefficients, so that no k_col is gated.  Calls of 1, 2, 1, 3, 1, 1, 4 and 2 steps from a counter seeded at
    499 put a call boundary on every alignment of that pattern (a call's first step takes the sweep kernel and k_pre
    instead, its last step the separate tail), each call reloads the coefficients of params.delt (the resume quirk,
    solver.py:154-155); the delt history, the record and U against the oracle's chunks."""
    chunks = (1, 2, 1, 3, 1, 1, 4, 2)
    kw = dict(adaptive_time=True, delt_max=4.9e-7 / N)
    s = chsimpy_amd.Solver(make(N, 10 ** 6, 'fast', dtype=dtype, **kw))
    s.prepare()
    eng = s._engine
    st = eng.get_state()
    st.computed_steps = 499
    st.skip_check = 1
    eng.set_state(st)
    o = orc.OracleSolver(orc.make_params(N, 10 ** 6, **kw))
    o.prepare()
    o.computed_steps = 499
    o.skip_check = True
    got = []
    for c in chunks:
        rows, rc = eng.step_n(c)
        assert rc == 0 and rows.shape == (c, 9)
        got.append(rows)
        o.solve_or_resume(c)
    rows = np.vstack(got)
    to = o.timedata.data()[1:]
    assert rows.shape == to.shape == (sum(chunks), 9)
    assert np.array_equal(rows[:, 0], to[:, 0]) and rows[0, 0] == 499
    assert len(np.unique(to[:, 8])) >= 4                      # the rule fired several times, on both sides of call boundaries
    errs = {c: relerr(rows[:, c], to[:, c]) for c in (1, 2, 4, 8)}
    eu = relerr(eng.get_U(), o.U)
    log_line(f"N={N} {dtype} adaptive, chunks {chunks} from step 499: delt {errs[8]:.3e} E {errs[1]:.3e} E2 {errs[2]:.3e} U {eu:.3e}")
    assert errs[8] < (1e-9 if dtype == 'float64' else 1e-4), errs
    assert errs[1] < (1e-9 if dtype == 'float64' else 1e-5) and errs[4] < (1e-9 if dtype == 'float64' else 1e-4), errs
    assert eu < tol, eu
    s.close(fetch_U=False)
