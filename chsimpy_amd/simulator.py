"""``Simulator``: owns a ``Solver`` and drives it one-shot or in ``update_every``
chunks, then exports.  Behaviour of ``chsimpy/simulator.py:15-87,135-156`` without
the matplotlib views (GUI is out of scope): a chunked run hands host snapshots to
an optional ``on_update(simulator)`` callback instead.
"""
import itertools

import numpy as np

from . import parameters, solver, utils


def chunk_sizes(total, every):
    """The calls of a chunked run: ``every`` steps at a time, the last one shortened so that the sum is
    ``total`` (simulator.py:57-81); endless when ``total`` is None (a run bounded by ``time_max`` only)."""
    if total is None:
        assert every > 0
        return itertools.repeat(every)
    if total <= 0:
        return iter(())
    every = min(every, total)
    assert every > 0
    whole, rest = divmod(total, every)
    return itertools.chain(itertools.repeat(every, whole), (rest,) if rest else ())


class Simulator:
    def __init__(self, params=None, U_init=None, on_update=None):
        self.params = parameters.Parameters() if params is None else params
        if U_init is None and self.params.Uinit_file is not None:
            U_init = utils.csv_import_matrix(self.params.Uinit_file)  # simulator.py:21-22
        self.solver = solver.Solver(self.params, U_init)
        self.steps_total = 0
        self.solution_file_id = None
        self.view = None
        self.on_update = on_update
        if on_update is None:
            # nothing an update could be shown on (simulator.py:33-34): one call for the whole run
            self.params.update_every = None
        else:
            # an observer between the chunks: every chunk is a solve_or_resume call of the reference to the letter,
            # hat_U = dctn(U) recomputed on entry (solver.py:159) instead of carried on the device
            self.solver.rederive_hat = True

    def solve(self):
        p = self.params
        sol = self.solver.solution
        self.solution_file_id = utils.get_or_create_file_id(p.file_id)
        if self.steps_total == 0:
            self.solver.prepare()
        if p.update_every is None:
            return self.solver.solve_or_resume(p.ntmax)
        # chunked driving (simulator.py:56-87).  The device keeps the field between the chunks; a chunk costs one
        # call of the C ABI, and the field is downloaded only if on_update looks at it.
        timed = p.time_max is not None and p.time_max > 0   # then only the time limit ends the run (simulator.py:58-59)
        for n in chunk_sizes(None if timed else p.ntmax - self.steps_total, p.update_every):
            stopped = sol.stop_reason == 'time-limit' or (sol.stop_reason != 'None' and p.full_sim is not True)
            if stopped:
                break
            self.solver.solve_or_resume(n)
            self.on_update(self)
            self.steps_total += n
        if sol.tau0 == 0:   # the energy rule never fired: the reference reports the last step (simulator.py:84-86)
            sol.tau0 = sol.computed_steps - 1
            sol.t0 = self.solver.time_passed
        return sol

    def export(self):
        """CSV export of the solution members named in ``export_csv`` as
        ``<file_id>.solution.<member>.csv[.bz2]`` (simulator.py:135-156)."""
        base = f"{self.solution_file_id}.solution"
        wanted = self.params.export_csv
        if wanted is None:
            return base
        ext = 'csv.bz2' if self.params.compress_csv else 'csv'
        sol = self.solver.solution
        for name in (m for m in wanted.replace(' ', '').split(',') if m):
            value = getattr(sol, name, None)
            if isinstance(value, np.ndarray):
                utils.csv_export_matrix(value, fname=f"{base}.{name}.{ext}")
        return base

    def render(self):
        return None  # views are out of scope

    def export_requested(self):
        return self.params.export_csv is not None

    def gui_requested(self):
        return False

    def gui_required(self):
        return False
