"""Attribute bag of all simulation knobs; field names and defaults of
``chsimpy/parameters.py:24-64``.  (The reference's YAML round-trip is out of scope.)"""
import copy

from . import utils
from .version import __version__


class Parameters:
    version = __version__

    def __init__(self):
        self.seed = 2023
        self.N = 512                    # [pixels]
        self.L = 2                      # [um]
        self.XXX = 0.875                # mean initial composition [mole fraction]
        self.temp = 650 + 273.15        # [K]
        self.B = 12.86                  # Gibbs-energy tuning parameter (Charles 1967)
        self.R = 0.0083144626181532     # [kJ/(K mol)]
        self.N_A = 6.02214076e+23
        self.delt = 3e-8
        self.delt_max = 9e-8
        self.M_tilde = 1.71e-8          # mobility factor [um^2/(kJ s)]
        self.kappa_tilde = None         # None: derived from the common tangent
        self.threshold = self.XXX
        self.ntmax = int(1e6)
        self.export_csv = None          # e.g. 'U,E2'
        self.png = False
        self.png_anim = False
        self.yaml = False
        self.no_gui = False
        self.file_id = 'auto'
        self.full_sim = False
        self.compress_csv = False
        self.time_max = None            # minutes of simulated time
        self.generator = 'uniform'      # 'uniform' | 'lcg' | 'sobol' | 'simplex'
        self.adaptive_time = False
        self.jitter = None
        self.update_every = 100
        self.no_diagrams = False
        self.Uinit_file = None
        self.func_A0 = lambda temp: utils.A0(temp)
        self.func_A1 = lambda temp: utils.A1(temp)
        # -- engine knobs (not in the reference) --------------------------------
        self.device = 0                 # HIP device ordinal
        self.dtype = 'float64'          # 'float64' | 'float32'
        self.engine = 'auto'            # 'auto' | 'direct' | 'fast'

    def deepcopy(self):
        return copy.deepcopy(self)

    def is_scalarwise_equal_with(self, other):
        if not isinstance(other, Parameters):
            return False
        skip = ('func_A0', 'func_A1', 'version')
        a = {k: v for k, v in self.__dict__.items() if k not in skip}
        b = {k: v for k, v in other.__dict__.items() if k not in skip}
        return a == b

    def __str__(self):
        d = {k: v for k, v in self.__dict__.items() if k not in ('func_A0', 'func_A1')}
        return str(dict(sorted(d.items())))
