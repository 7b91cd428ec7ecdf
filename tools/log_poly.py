"""Derives the polynomial used by the device log (chs_math.h).

  log(q) = 2 atanh(s),  s = (q-1)/(q+1),  |s| <= (sqrt2-1)/(sqrt2+1) = 0.17157...
         = 2 s + s * z * P(z),   z = s^2,   P(z) = 2 (1/3 + z/5 + z^2/7 + ...)
P is interpolated at Chebyshev nodes of [0, zmax] in 60-digit arithmetic (near-minimax).
"""
import mpmath as mp
import numpy as np

mp.mp.dps = 60


def P(z):
    if z == 0:
        return mp.mpf(2) / 3
    t = mp.sqrt(z)
    return 2 * (mp.atanh(t) / t - 1) / z


def fit(deg, zmax):
    n = deg + 1
    nodes = [zmax / 2 * (1 + mp.cos(mp.pi * (2 * i + 1) / (2 * n))) for i in range(n)]
    A = mp.matrix(n, n)
    b = mp.matrix(n, 1)
    for i, x in enumerate(nodes):
        for j in range(n):
            A[i, j] = x ** j
        b[i] = P(x)
    c = mp.lu_solve(A, b)
    return [c[i] for i in range(n)]


if __name__ == '__main__':
    smax = (mp.sqrt(2) - 1) / (mp.sqrt(2) + 1) * mp.mpf('1.02')
    zmax = smax ** 2
    for deg in (4, 6, 7):
        c = fit(deg, zmax)
        err = 0
        for i in range(2001):
            z = zmax * i / 2000
            approx = sum(ci * z ** k for k, ci in enumerate(c))
            s = mp.sqrt(z) if z > 0 else mp.mpf(0)
            full_exact = 2 + z * P(z)
            full_apx = 2 + z * approx
            err = max(err, abs(full_apx - full_exact) / full_exact)
        print(f"deg {deg}: max rel err of log(q)/s = {mp.nstr(err, 3)}")
        print("  coeffs:", ", ".join(repr(float(x)) for x in c))
