"""``Solver``: the Cahn-Hilliard integrator of ``chsimpy/solver.py`` with the
timestep loop running on an MI355X through the C ABI of ``include/chs_hip.h``.

Same constructor, methods and result object as the reference class
(``Solver(params, U_init)``, ``prepare()``, ``solve_or_resume(nsteps)`` ->
``Solution``), same quirks: the first ``solve_or_resume`` after ``prepare`` runs
``nsteps-1`` iterations (solver.py:160-165), ``hat_U`` is re-derived from ``U`` on
every call (solver.py:159), every call starts with the coefficient grids of
``params.delt`` while ``self.delt`` keeps its adapted value until the adaptive
step fires again (solver.py:154-155 against 185-193), ``prepare`` does not reset
``delt``/``time_delta_sum``/``skip_check`` (solver.py:50-54 are only set in
``__init__``), and a field assigned to ``solution.U`` between two calls is what the
next call continues from (solver.py:158).
"""
import numpy as np

from . import _lib, mport
from .solution import Solution
from .timedata import TimeData


class Solver:
    def __init__(self, params=None, U_init=None):
        self.params = params
        self.solution = Solution(self.params)
        N = params.N
        self.skip_check = False
        self.time_delta_sum = 0.0
        self.time_passed = 0.0
        self._prepared = False
        self.delt = self.params.delt
        self._engine = None

        self.create_rand = None
        self._pcg = None
        self._pcg_state0 = None  # the default generator's state before the start field was drawn
        self.device_rng = True   # draw start field and jitter noise on the device when the generator is numpy's PCG64
        # A call that follows a completed call (fixed time step, nothing assigned in between) continues the device
        # loop where it stopped, hat_U included; True recomputes hat_U = dctn(U) at every call as solver.py:159
        # does (the same array up to rounding, one transform more per call)
        self.rederive_hat = False
        # with rederive_hat: take the first step's operand (row transform of EnergieEut(U)) over from the previous call's
        # last step instead of computing it again (CHS_STEP_KEEP_T1; the same run bit for bit, no faster at N=4096)
        self.rederive_keeps_t1 = False
        self._U_init = None
        # initial concentration field, solver.py:59-82
        if U_init is not None:
            if U_init.shape == (N, N):
                self.U_init = U_init
            else:
                print("U_init has wrong shape, must match parameters.N")
                raise SystemExit(1)
        elif params.generator == 'lcg':
            # no "-0.5" here, as in the reference (solver.py:66)
            self.U_init = params.XXX + (params.XXX * 0.01 * mport.matlab_lcg_sample(N, N, params.seed))
        elif params.generator == 'sobol':
            from scipy.stats import qmc
            qrng = qmc.Sobol(d=N, seed=params.seed)
            self.create_rand = lambda n: qrng.random(n)
        elif params.generator == 'simplex':
            try:
                import opensimplex
            except ImportError as e:  # pragma: no cover - depends on the image
                raise ImportError("generator='simplex' needs the `opensimplex` package") from e
            self.create_rand = lambda n: opensimplex.noise2array(np.linspace(0, 48, n), np.linspace(0, 48, n))
        else:
            rng = np.random.Generator(np.random.PCG64(params.seed))
            self.create_rand = lambda n: rng.random((n, n))
            self._pcg = rng   # the device can continue this stream itself (jitter, solve_or_resume)
            # The start field XXX + XXX*0.01*(rng.random((N,N)) - 0.5) (solver.py:78-82) is not drawn
            # here: prepare() lets the device draw it from the same stream (no N*N*8-byte upload), the
            # attribute U_init computes it on the host when somebody looks at it.  The generator moves
            # on by the N*N draws either way.
            self._pcg_state0 = rng.bit_generator.state
            rng.bit_generator.advance(N * N)
        if self._U_init is None and self._pcg_state0 is None:
            self._U_init = params.XXX + (params.XXX * 0.01 * (self.create_rand(N) - 0.5))

    @property
    def U_init(self):
        if self._U_init is None and self._pcg_state0 is not None:
            g = np.random.Generator(np.random.PCG64())
            g.bit_generator.state = self._pcg_state0
            p = self.params
            self._U_init = p.XXX + (p.XXX * 0.01 * (g.random((p.N, p.N)) - 0.5))
        return self._U_init

    @U_init.setter
    def U_init(self, value):
        self._U_init = value

    # -- engine ----------------------------------------------------------------
    def _consts(self):
        p, s = self.params, self.solution
        c = _lib.chs_consts()
        c.N = int(p.N)
        c.dtype = _lib.DTYPES[str(getattr(p, 'dtype', 'float64'))]
        c.device = int(getattr(p, 'device', 0) or 0)
        c.engine = _lib.ENGINES[str(getattr(p, 'engine', 'auto'))]
        c.adaptive_time = 1 if p.adaptive_time else 0
        c.full_sim = 1 if p.full_sim else 0
        c.RT, c.BRT, c.B = float(s.RT), float(s.BRT), float(p.B)
        c.A0, c.A1, c.Amr = float(s.A0), float(s.A1), float(s.Amr)
        c.kappa_tilde, c.L, c.delx = float(s.kappa_tilde), float(p.L), float(s.delx)
        c.delt, c.delt_max, c.M_tilde = float(p.delt), float(p.delt_max), float(p.M_tilde)
        c.threshold = float(p.threshold)
        c.time_limit_s = float(p.time_max * 60) if (p.time_max is not None and p.time_max > 0) else 0.0
        return c

    def _get_engine(self):
        if self._engine is None:
            self._engine = _lib.Engine(self._consts(), self.solution.lam)
            self._push_state()
        return self._engine

    def _push_state(self):
        st = self._engine.get_state()
        st.delt = float(self.delt)
        st.time_delta_sum = float(self.time_delta_sum)
        st.time_passed = float(self.time_passed)
        st.skip_check = 1 if self.skip_check else 0
        self._engine.set_state(st)

    def _pull_state(self):
        st = self._engine.get_state()
        sol = self.solution
        self.delt = st.delt
        self.time_delta_sum = st.time_delta_sum
        self.time_passed = st.time_passed
        self.skip_check = bool(st.skip_check)
        sol.computed_steps = int(st.computed_steps)
        sol.tau0 = int(st.tau0) if float(st.tau0).is_integer() else st.tau0
        sol.t0 = st.t0
        sol.stop_reason = _lib.STOP_NAMES[st.stop_reason]
        return st

    def close(self, fetch_U=True):
        """Free the device engine.  A download of the field that is still pending (Solution.U is
        fetched on first access) happens now unless the caller does not need it."""
        if self._engine is not None:
            if fetch_U:
                _ = self.solution.U
            else:
                # the caller needs scalars only: solution.U stays what the host already has (None when
                # the field was never downloaded)
                self.solution._bind_device_U(self.solution.__dict__.get('_U'), track=False)
            self._engine.close()
            self._engine = None

    # -- solver.py:84-135 ----------------------------------------------------------
    def prepare(self):
        N = self.params.N
        eng = self._get_engine()
        self._push_state()
        on_device = self._U_init is None and self._pcg_state0 is not None and self.device_rng
        if on_device:
            st = self._pcg_state0['state']
            p = self.params
            eng.init_U_pcg64(p.XXX, p.XXX * 0.01, st['state'], st['inc'])
            U = None
        else:
            U = self.U_init.copy()
            assert (U.shape == (N, N))
            eng.set_U(U)
        row = eng.prepare()
        data = TimeData()
        data.insert(it=0, delt=row[8], E=row[1], E2=row[2], SA=0, domtime=0, Ra=row[5], L2=0, PS=row[7])
        # on_device: downloaded when somebody looks at solution.U
        self.solution._bind_device_U(U, eng.get_U if on_device else None)
        self.solution.timedata = data
        self.solution.tau0 = 0.0
        self.solution.t0 = 0.0
        self.solution.stop_reason = 'None'
        self.solution.computed_steps = 1
        self._prepared = True

    # -- solver.py:137-252 ---------------------------------------------------------
    def solve_or_resume(self, nsteps=None):
        """Run the timestep loop on the device and return the solution object."""
        assert (self._prepared is True)
        p = self.params
        if nsteps is None:
            nsteps = max(p.ntmax, 0)
        eng = self._engine
        if self.solution.__dict__.get('_U_dirty') or self.solution._host_edited():
            # the caller replaced the field since the last call -- by assignment, or by editing the array
            # `solution.U` handed out in place: `U = self.solution.U` (solver.py:158) is where the reference starts
            eng.set_U(self.solution.__dict__['_U'])
            self.solution.__dict__['_U_dirty'] = False
            self.solution.__dict__['_U_print'] = None
        itbegin = 1 if self.solution.computed_steps == 1 else 0
        count = max(int(nsteps) - itbegin, 0)

        jitter_on = p.jitter is not None and 0.0 < p.jitter < 0.1
        if not jitter_on:
            eng.set_jitter_noise(0.0, None)
            # (a call that reaches ntmax ends the run: its last step need not prepare a continuation)
            rows, rc = eng.step_n(count, rederive_hat=self.rederive_hat, keep_t1=self.rederive_hat and self.rederive_keeps_t1,
                                  last_call=self.solution.computed_steps + count >= p.ntmax)
            self._absorb(rows, rc, count)
        elif self._pcg is not None and self.device_rng:
            # The reference's default generator (numpy PCG64, solver.py:78-82): the device continues
            # its stream from the generator's current state -- N*N draws per step in C order, exactly
            # what `create_rand(N)` would have returned (solver.py:211) -- and the whole chunk runs as
            # one device loop; afterwards the host generator is moved on by what the device consumed.
            st = self._pcg.bit_generator.state['state']
            eng.set_jitter_pcg64(p.jitter, st['state'], st['inc'])
            rows, rc = eng.step_n(count)
            try:
                self._absorb(rows, rc, count)
            finally:
                self._pcg.bit_generator.advance(int(rows.shape[0]) * p.N * p.N)
        else:
            # Other generators: the noise comes from the host generator so that its stream stays the
            # reference's (solver.py:211); hat_U is carried between the one-step
            # calls exactly as inside the reference's loop.
            first = True
            for _ in range(count):
                eng.set_jitter_noise(p.jitter, self.create_rand(p.N))
                rows, rc = eng.step_n(1, carry_hat=not first)
                first = False
                if self._absorb(rows, rc, 1):
                    break
            if count == 0:
                self._pull_state()
        self.solution._bind_device_U(None, eng.get_U)   # downloaded when somebody looks at solution.U
        return self.solution

    def _absorb(self, rows, rc, requested):
        """Append the rows of one device call; True when the device loop broke early."""
        sol = self.solution
        if rc == _lib.CHS_ENAN:
            if rows.shape[0] > 1:
                sol.timedata.extend(rows[:-1])
            self._pull_state()
            sol._bind_device_U(self._engine.get_U(), track=False)
            # the reference fails the same way: assert in TimeData.insert (timedata.py:10)
            raise AssertionError("NaN in a recorded scalar (U left (0,1)) at step %d" % sol.computed_steps)
        if rows.shape[0]:
            sol.timedata.extend(rows)
        st = self._pull_state()
        if rows.shape[0] < requested:
            return True
        return (st.stop_reason == _lib.CHS_STOP_ENERGY and not self.params.full_sim
                and st.tau0 == st.computed_steps)
