"""Portable linear-congruential sampler used for reproducible initial fields.

Mirrors the behaviour of the reference's ``chsimpy/mport.py:8-32`` ("BSD rand"
constants, recurrence carried out in IEEE float64, MATLAB-style column-major
fill, division by ``2**31 - 1``).  The float64 arithmetic is essential: the
product ``a*x`` exceeds 2**53, so every step works on a *rounded* double and the
modulo is the float64 ``fmod`` -- exact 64-bit integer math gives a different
stream and fails the known-answer matrix of the reference's ``tests/test.py:25-37``.
"""
import math

import numpy as np

_A = 1103515245.0
_C = 12345.0
_M = float(2 ** 31)


def lcg_stream(seed, count):
    """First ``count`` states of x <- fmod(a*x + c, m), in float64."""
    out = np.empty(count, dtype=np.float64)
    x = float(seed)
    for i in range(count):
        x = math.fmod(_A * x + _C, _M)
        out[i] = x
    return out


def matlab_lcg_sample(n1, n2, seed):
    """n1 x n2 matrix on [0, 1): the stream laid out column by column."""
    stream = lcg_stream(seed, n1 * n2)
    sample = stream.reshape(n2, n1).T.copy()  # element i -> [i % n1, i // n1]
    sample /= (_M - 1.0)
    return sample
