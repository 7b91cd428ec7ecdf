"""chsimpy_amd -- MI355X-native timestep engine for the Cahn-Hilliard solver loop
of uncertaintyhub/chsimpy, behind chsimpy's own ``Parameters``/``Solver``/
``Simulator``/``Solution`` interface.  Host code is Python; every per-timestep
computation runs in hand-written HIP kernels (``chsimpy_amd/csrc``) reached through
the C ABI of ``include/chs_hip.h``.
"""
from .version import __version__
from .parameters import Parameters
from .solution import Solution
from .timedata import TimeData
from .solver import Solver
from .simulator import Simulator
from . import utils, mport

__all__ = ['Parameters', 'Solution', 'TimeData', 'Solver', 'Simulator', 'utils', 'mport', '__version__']
