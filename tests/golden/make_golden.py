"""Generates the committed golden vectors from the oracle (NOT from the reference,
which may neither be imported nor run: SURVEY.md section 8c).

    python tests/golden/make_golden.py

n64_lcg_40steps.npz : N=64, generator='lcg' (seed-portable, pinned by the reference's
                      LCG known-answer test), ntmax=40, full_sim, kappa_tilde explicit.
n128_lcg_60steps.npz : the same generator at N=128, the smallest size of the fast transform engine
                      (the reference-pinned LCG start field through the FFT-based kernels).
n128_seed2023_200steps.npz : configs[0] of BASELINE.json (N=128, ntmax=200, seed 2023,
                      cinit 0.875): timedata + final U; its E/E2/min/max agree with the
                      reference observations recorded in SURVEY.md section 8(c).
n64_adaptive_600.npz : adaptive_time=True beyond step 500 (delt history + final U).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import chs_oracle as orc  # noqa: E402


def run(p):
    s = orc.OracleSolver(p)
    s.prepare()
    s.solve_or_resume()
    return s


def main():
    s = run(orc.make_params(64, 40, generator='lcg', seed=2023))
    np.savez_compressed(os.path.join(HERE, 'n64_lcg_40steps.npz'), U_init=s.U_init, U_final=s.U,
                        timedata=s.timedata.data())
    s = run(orc.make_params(128, 60, generator='lcg', seed=2023))
    np.savez_compressed(os.path.join(HERE, 'n128_lcg_60steps.npz'), U_init=s.U_init, U_final=s.U,
                        timedata=s.timedata.data())
    s = run(orc.make_params(128, 200))
    np.savez_compressed(os.path.join(HERE, 'n128_seed2023_200steps.npz'), U_final=s.U,
                        timedata=s.timedata.data())
    s = run(orc.make_params(64, 600, adaptive_time=True))
    np.savez_compressed(os.path.join(HERE, 'n64_adaptive_600.npz'), U_final=s.U, timedata=s.timedata.data())


if __name__ == '__main__':
    main()
