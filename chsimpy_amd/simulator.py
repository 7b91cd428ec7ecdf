"""``Simulator``: owns a ``Solver`` and drives it one-shot or in ``update_every``
chunks, then exports.  Behaviour of ``chsimpy/simulator.py:15-87,135-156`` without
the matplotlib views (GUI is out of scope): a chunked run hands host snapshots to
an optional ``on_update(simulator)`` callback instead.
"""
import numpy as np

from . import parameters, solver, utils


class Simulator:
    def __init__(self, params=None, U_init=None, on_update=None):
        if params is None:
            params = parameters.Parameters()
        self.params = params
        if U_init is None and params.Uinit_file is not None:
            U_init = utils.csv_import_matrix(params.Uinit_file)  # simulator.py:21-22
        self.solver = solver.Solver(params, U_init)
        self.steps_total = 0
        self.solution_file_id = None
        self.view = None
        self.on_update = on_update
        if on_update is None:
            # no target an update could be applied to (simulator.py:33-34)
            self.params.update_every = None

    def solve(self):
        self.solution_file_id = utils.get_or_create_file_id(self.params.file_id)
        if self.steps_total == 0:
            self.solver.prepare()
        if self.params.update_every is None:
            return self.solver.solve_or_resume(self.params.ntmax)
        # chunked driving, simulator.py:56-87
        part = 0
        steps_end = self.params.ntmax
        if self.params.time_max is not None and self.params.time_max > 0:
            steps_end = utils.get_int_max_value()
        dsteps = min(steps_end, self.params.update_every)
        assert (dsteps > 0)
        sol = self.solver.solution
        while ((self.steps_total + dsteps) <= steps_end
               and (sol.stop_reason == 'None' or self.params.full_sim is True)
               and (sol.stop_reason != 'time-limit')):
            self.solver.solve_or_resume(dsteps)
            self.on_update(self)
            self.steps_total += dsteps
            part += 1
            diff = steps_end - self.steps_total
            if 0 < diff < dsteps:
                dsteps = diff
            elif diff < 0:
                raise Exception("Something went wrong.")
        if sol.tau0 == 0:
            sol.tau0 = sol.computed_steps - 1
            sol.t0 = self.solver.time_passed
        return sol

    def export(self):
        """CSV export, simulator.py:135-156: ``<file_id>.solution.<member>.csv[.bz2]``."""
        fname_sol = f"{self.solution_file_id}.solution"
        solution = self.solver.solution
        export_csv = self.params.export_csv
        if export_csv is not None:
            fext = 'csv.bz2' if self.params.compress_csv else 'csv'
            for member in export_csv.replace(' ', '').split(','):
                varray = None
                if hasattr(solution, member):
                    varray = getattr(solution, member)
                if isinstance(varray, np.ndarray):
                    utils.csv_export_matrix(varray, fname=f"{fname_sol}.{member}.{fext}")
        return fname_sol

    def render(self):
        return None  # views are out of scope

    def export_requested(self):
        return self.params.export_csv is not None

    def gui_requested(self):
        return False

    def gui_required(self):
        return False
