"""Repeated runs of the reference's default workflow (N=512, energy stop at step 1674) in one process: device time per run.
Variants isolate what, between two runs, makes a later run stall."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
import chsimpy_amd
from chsimpy_amd import _lib

def run(tag, variant, reps=8):
    out, keep = [], []
    pre = np.empty((512, 512))
    for rep in range(reps):
        p = chsimpy_amd.Parameters(); p.N, p.kappa_tilde, p.no_gui = 512, 0.0002989112919661156, True
        s = chsimpy_amd.Solver(p)
        s.prepare()
        sol = s.solve_or_resume(p.ntmax)
        eng = s._engine
        out.append("%.1f" % eng.last_step_ms())
        if variant == 'close': s.close()
        elif variant == 'close_nofetch': s.close(fetch_U=False)
        elif variant == 'get_into_prealloc':      # the C download, no fresh host memory
            eng._check(eng.lib.chs_get_U(eng._h, _lib._dptr(pre)), 'chs_get_U'); s.close(fetch_U=False)
        elif variant == 'prefault_then_get':      # fresh array, pages touched by numpy first, then the C download
            a = np.empty((512, 512)); a.fill(0.0)
            eng._check(eng.lib.chs_get_U(eng._h, _lib._dptr(a)), 'chs_get_U'); keep.append(a); s.close(fetch_U=False)
        elif variant == 'touch_fresh_2mb':        # no download at all: only a fresh 2 MB array written by the CPU
            a = np.empty((512, 512)); a[:] = 1.0; keep.append(a); s.close(fetch_U=False)
        elif variant == 'touch_fresh_2mb_free':
            a = np.empty((512, 512)); a[:] = 1.0; del a; s.close(fetch_U=False)
    print(f"{tag}: device ms per run {' '.join(out)}", flush=True)

if __name__ == '__main__':
    v = sys.argv[1]
    run(v, v)
