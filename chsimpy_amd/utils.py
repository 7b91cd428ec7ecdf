"""Numeric helpers and on-disk formats of the hot path.

Counterparts of the numeric part of ``chsimpy/utils.py`` (A0/A1 26-31, eigenvalue
grid 34-36, semi-implicit coefficients 39-49, CSV matrix I/O 79-90, file ids
113-117) plus the sympy-based kappa helpers (143-180) that ``Solution`` needs when
``kappa_tilde`` is not given.  YAML, sysinfo and plotting helpers are out of scope.
"""
from datetime import datetime

import numpy as np


def A0(T):
    """Redlich-Kister coefficient A0(T) [kJ/mol] (Kim & Sanders), ``utils.py:26-27``."""
    return 186.0575 - 0.3654 * T


def A1(T):
    """Redlich-Kister coefficient A1(T) [kJ/mol], ``utils.py:30-31``."""
    return 43.7207 - 0.1401 * T


def eigenvalues_1d(N):
    """lam_i = 2cos(pi i/(N-1)) - 2, i = 0..N-1: the 1-D table the device engine keeps.

    Same expression (and therefore the same float64 values) as the factor of the
    outer sums at ``utils.py:35-36`` -- including the reference's ``N-1`` in the
    cosine argument.
    """
    return 2 * np.cos(np.pi * (np.arange(0, N - 1 + 1)) / (N - 1)) - 2


def eigenvalues(N):
    """N x N grid leig_ij = lam_i + lam_j (``utils.py:34-36``)."""
    lam = eigenvalues_1d(N)
    return lam.reshape(N, 1) + lam.reshape(1, N)


def get_coefficients(N, kappa_tilde, delt, delx2):
    """(CHeig, Seig) of the semi-implicit update, ``utils.py:39-49``.

    Host-side convenience only; the device forms both on the fly.
    """
    lam1 = delt / delx2
    lam2 = kappa_tilde * lam1 / delx2
    leig = eigenvalues(N)
    CHeig = np.ones((N, N)) + lam2 * leig * leig
    Seig = lam1 * leig
    return CHeig, Seig


# -- on-disk formats (utils.py:79-90) -----------------------------------------

def csv_export_matrix(V, fname):
    if fname.endswith('bz2'):
        import pandas as pd
        pd.DataFrame(V).to_csv(fname, index=False, header=None, sep=',', compression='bz2')
    else:
        np.savetxt(fname, V, delimiter=',', fmt='%s')


def csv_import_matrix(fname):
    if fname.endswith('bz2'):
        import pandas as pd
        return pd.read_csv(fname, sep=',', header=None, compression='bz2').values
    return np.loadtxt(fname, delimiter=',')


def get_or_create_file_id(file_id):
    """``utils.py:113-117``"""
    if file_id == 'auto' or file_id is None or file_id == '' or str(file_id).lower() == 'none':
        return datetime.now().strftime('%d%m%Y-%H%M%S')
    return file_id


def vars_to_list(obj):
    """Public non-callable attributes as ``name, value`` strings (``utils.py:213-223``)."""
    out = []
    for x in dir(obj):
        if x.startswith('_') or not hasattr(obj, x):
            continue
        v = getattr(obj, x)
        if callable(v):
            continue
        out.append(f"{x}, {v}")
    return out


def csv_export_list(fname, text):
    """``utils.py:226-228``"""
    with open(fname, 'w') as f:
        f.writelines(text)


def get_system_info():
    """Host description for ``-metadata.csv`` (cf. ``utils.py:122-142``), plus the GPUs torch can see."""
    import platform
    import sys
    from .version import __version__
    uname = platform.uname()
    info = [f"system, {uname.system}", f"nodename, {uname.node}", f"kernel-release, {uname.release}",
            f"machine, {uname.machine}"]
    try:
        import psutil
        info += [f"cores_phys, {psutil.cpu_count(logical=False)}", f"cores_total, {psutil.cpu_count(logical=True)}"]
    except ImportError:  # pragma: no cover
        import os
        info += [f"cores_total, {os.cpu_count()}"]
    try:
        import torch
        n = torch.cuda.device_count()
        info += [f"gpus, {n}"]
    except Exception:  # pragma: no cover
        pass
    info += [f"localtime, {datetime.now().strftime('%Y-%m-%d %H:%M:%S')}", f"argv, '{' '.join(sys.argv)}'",
             f"chsimpy_amd-version, {__version__}"]
    return info


# -- thermodynamic helpers (sympy; utils.py:143-180) ---------------------------

def _energy_expr(sym, c, R, T, B, A0_, A1_):
    return (R * T * (c * (sym.log(c) - B) + (1 - c) * sym.log(1 - c))
            + (A0_ + A1_ * (1 - 2 * c)) * c * (1 - c))


def get_miscibility_gap(R, T, B, A0, A1, xlower=0.7, xupper=0.9999, prec=7):
    """Common-tangent compositions (c_A, c_B); same equations, start values and
    ``prec=7`` as ``utils.py:143-160`` so the default kappa agrees with the reference."""
    import sympy as sym
    x1 = sym.Symbol('x1', real=True)
    x2 = sym.Symbol('x2', real=True)
    y1 = _energy_expr(sym, x1, R, T, B, A0, A1)
    y2 = _energy_expr(sym, x2, R, T, B, A0, A1)
    dy1 = sym.diff(y1, x1, 1)
    dy2 = sym.diff(y2, x2, 1)
    eq1 = sym.Eq(dy1, dy2)
    eq2 = sym.Eq(dy1, (y2 - y1) / (x2 - x1))
    return sym.nsolve((eq1, eq2), (x1, x2), (xlower, xupper), prec=prec)


def get_distance_common_tangent(R, T, B, A0, A1, at):
    """Distance between E and its common tangent at composition ``at`` (``utils.py:163-171``)."""
    import sympy as sym
    x = sym.Symbol('x', real=True)
    E = _energy_expr(sym, x, R, T, B, A0, A1)
    ca, cb = get_miscibility_gap(R=R, T=T, B=B, A0=A0, A1=A1)
    m = (E.subs(x, cb) - E.subs(x, ca)) / (cb - ca)
    dist = (E - m * (x - ca) - E.subs(x, ca)).subs(x, at)
    return np.float64(dist)


def get_roots_of_EPP(R, T, A0, A1):
    """Spinodal compositions: roots of E'' on (0,1) (``utils.py:174-180``)."""
    import sympy as sym
    x = sym.Symbol('x', real=True, positive=True)
    c = x
    EPP = (-2 * A0 * c ** 2 + 2 * A0 * c + 12 * A1 * c ** 3 - 18 * A1 * c ** 2 + 6 * A1 * c - R * T) / (c ** 2 - c)
    roots = sym.solveset(EPP, x, domain=sym.Interval(0, 1))
    return list(roots)
