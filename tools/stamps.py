"""Diagnostic: per-phase cycle shares of the row / column kernels (build with -DCHS_STAMPS)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import chsimpy_amd
from chsimpy_amd import _lib

NST = 12
p = chsimpy_amd.Parameters()
NN = int(os.environ.get('CHS_STAMP_N', '4096'))
p.N, p.ntmax, p.full_sim, p.kappa_tilde = NN, 10 ** 9, os.environ.get('CHS_STAMP_FULLSIM', '1') == '1', 0.0002989112919661156
p.dtype = os.environ.get('CHS_STAMP_DTYPE', 'float64')
s = chsimpy_amd.Solver(p)
s.prepare()
s.solve_or_resume(6)
lib = _lib.load()
lib.chs_debug_stamps.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.c_int]
names = {0: ['recombine^T (T2 loads)', 'inv passes', 'store U', 'pointwise (+edges)', 'fwd passes', 'recombine (T1 stores)'],
         1: ['stage in (tile loads)', 'fwd passes', 'spectral (hat r/w)', 'inv passes', 'stage out']}
NB = int(os.environ.get('CHS_STAMP_BLOCKS', '2048'))
for which, nblk in ((0, NB), (1, NB)):
    buf = np.zeros(8192 * NST, dtype=np.uint64)
    rc = lib.chs_debug_stamps(which, buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size)
    st = buf.reshape(8192, NST)[:nblk].astype(np.int64)
    n = len(names[which])
    if which == 1:
        # program order of the k_col stamps: 0 entry, 1 staged, 2 fwd done, 5 spectral done, 3 reductions done, 4 inv done
        st = st[:, [0, 1, 2, 5, 3, 4] + list(range(6, NST))]
        names[1] = ['stage in (tile loads)', 'fwd passes', 'recombine+spectral (hat r/w)', 'block reduction', 'inv passes']
    if which == 0:
        r1 = st[:, 7:11]   # wave 1 of k_col (plain recombination path) parks its stamps in kernel 0's free slots
        k1 = r1[:, 0] > 0
        if k1.any(): print('   k_col wave 1 (plain path): slot 0', int(np.median(r1[k1, 1] - r1[k1, 0])), '| slots 1-3',
              int(np.median(r1[k1, 2] - r1[k1, 1])), '| slots 4-7', int(np.median(r1[k1, 3] - r1[k1, 2])))
    if which == 1:
        r = st[:, 6:10]
        okk = (r[:, 0] > 0) & (st[:, 2] > 0)
        if okk.any(): print('   wave 0 inside the recombination: fwd done -> enter', int(np.median(r[okk, 0] - st[okk, 2])),
              '| special slot', int(np.median(r[okk, 1] - r[okk, 0])), '| slot 0', int(np.median(r[okk, 2] - r[okk, 1])),
              '| slots 1-3', int(np.median(r[okk, 3] - r[okk, 2])), '| slots 4-7', int(np.median(st[okk, 10] - r[okk, 3])),
              '| end of the slots -> past the join (wait for the hat_U stores)', int(np.median(st[okk, 3] - st[okk, 10])))
    d = np.diff(st[:, :n + 1], axis=1)
    ok = np.all(d >= 0, axis=1) & (st[:, 0] > 0)
    d = d[ok]
    tot = (st[ok, n] - st[ok, 0])
    print(f"kernel {which}: {ok.sum()} workgroups, lifetime median {np.median(tot)} ticks, span of starts {st[ok,0].max()-st[ok,0].min()} ticks, kernel span {st[ok,n].max()-st[ok,0].min()}")
    for i, nm in enumerate(names[which]):
        print(f"   {nm:28s} median {np.median(d[:, i]):9.0f}  p90 {np.percentile(d[:, i], 90):9.0f}   {100*np.median(d[:, i])/np.median(tot):5.1f}%")
s.close()
