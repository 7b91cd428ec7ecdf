"""Per-step scalar log of a run: the 9-column table of ``chsimpy/timedata.py``.

Column order (``chsimpy/timedata.py:9``): it, E, E2, SA, domtime, Ra, L2, PS, delt.
The device engine produces whole blocks of rows (one per timestep of a
``solve_or_resume`` call), so besides the reference's row-wise ``insert`` there is
a bulk ``extend``.  Storage grows geometrically instead of by ``np.append`` copies.
"""
import numpy as np

COLUMNS = ('it_range', 'E', 'E2', 'SA', 'domtime', 'Ra', 'L2', 'PS', 'delt')


class TimeData:
    def __init__(self):
        self._buf = np.empty((64, 9), dtype=np.float64)
        self._n = 0

    # -- growth ---------------------------------------------------------------
    def _reserve(self, extra):
        need = self._n + extra
        if need > self._buf.shape[0]:
            cap = max(need, 2 * self._buf.shape[0])
            nb = np.empty((cap, 9), dtype=np.float64)
            nb[:self._n] = self._buf[:self._n]
            self._buf = nb

    def insert(self, it, delt, E, E2, SA, domtime, Ra, L2, PS):
        """Append one row; a NaN in it is an AssertionError (``timedata.py:10``)."""
        self._reserve(1)
        self._buf[self._n] = (it, E, E2, SA, domtime, Ra, L2, PS, delt)
        self._n += 1
        assert not np.any(np.isnan(self._buf[self._n - 1]))

    def extend(self, rows):
        """Append a (k, 9) block produced by the device loop."""
        rows = np.asarray(rows, dtype=np.float64).reshape(-1, 9)
        self._reserve(rows.shape[0])
        self._buf[self._n:self._n + rows.shape[0]] = rows
        self._n += rows.shape[0]
        assert not np.any(np.isnan(rows))

    def data(self):
        return self._buf[:self._n]

    def __len__(self):
        return self._n

    @property
    def it_range(self):
        return self._buf[:self._n, 0]

    @property
    def E(self):
        return self._buf[:self._n, 1]

    @property
    def E2(self):
        return self._buf[:self._n, 2]

    @property
    def SA(self):
        return self._buf[:self._n, 3]

    @property
    def domtime(self):
        return self._buf[:self._n, 4]

    @property
    def Ra(self):
        return self._buf[:self._n, 5]

    @property
    def L2(self):
        return self._buf[:self._n, 6]

    @property
    def PS(self):
        return self._buf[:self._n, 7]

    @property
    def delt(self):
        return self._buf[:self._n, 8]

    def energy_falls(self, it=None):
        """Stop predicate of ``timedata.py:51-63``: E2[it-1] > E2[it] > E2[0].

        The engine evaluates the same predicate on the device every step; this
        host copy exists for API parity and for the tests.
        """
        e2 = self.E2
        return bool(e2[it - 1] > e2[it] > e2[0])
