#!/usr/bin/env python3
"""A/B timing of library variants INSIDE ONE PROCESS: every variant (chsimpy_amd/lib/variants/*.so, or paths given)
is loaded side by side through its own ctypes binding, gets its own engine on the same GPU, and the timed calls
alternate A, B, C, A, B, C, ... back to back -- same box, same thermal / power state, no process start between them.
Run-to-run spread is a few tenths of a per cent (one process per variant, tools/ab.sh: +-1.5 %); but WHERE an engine's buffers
land in memory is worth up to ~1.2 % between identical libraries -- use --copies 2..3 and/or both creation orders (--reverse)
before believing a difference below ~1.5 %.

    python tools/ab_inproc.py [--grid 4096] [--dtype float64] [--steps 300] [--rounds 12] [--mode literal|continue|adaptive]
                              [--glob 'a_*'] [paths...]
Prints per variant: median / min ms per step (device time of the call, HIP events) and the median of the per-round
ratio against the first variant."""
import argparse
import glob
import importlib.util
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

KAPPA = 0.0002989112919661156


def bind(path, tag):
    """A private copy of chsimpy_amd._lib bound to `path`."""
    spec = importlib.util.spec_from_file_location(f'chsimpy_amd._lib_{tag}', os.path.join(ROOT, 'chsimpy_amd', '_lib.py'),
                                                  submodule_search_locations=None)
    mod = importlib.util.module_from_spec(spec)
    mod.__package__ = 'chsimpy_amd'
    spec.loader.exec_module(mod)
    os.environ['CHS_LIB_PATH'] = path
    try:
        mod.load()
    finally:
        del os.environ['CHS_LIB_PATH']
    return mod


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--grid', type=int, default=4096)
    ap.add_argument('--dtype', default='float64')
    ap.add_argument('--steps', type=int, default=300)
    ap.add_argument('--rounds', type=int, default=12)
    ap.add_argument('--warm', type=int, default=300)
    ap.add_argument('--mode', default='continue', choices=['literal', 'continue', 'adaptive', 'estop'])
    ap.add_argument('--delt-max', type=float, default=None)
    ap.add_argument('--glob', default='*')
    ap.add_argument('--envs', default='', help="';'-separated environment settings, each 'K=V[,K=V]': every variant gets one engine per setting (set while the engine is created), e.g. 'CHS_SLAB=0;CHS_SLAB=1'")
    ap.add_argument('--copies', type=int, default=1, help='engines per variant, created interleaved (a b c a b c ...): where an engine\'s buffers land in memory is worth up to ~1.2 %% between IDENTICAL libraries; several engines per variant average that out')
    ap.add_argument('--reverse', action='store_true', help='create the engines in reverse order (buffer placement differs by engine: a ratio that flips with the order is placement, not code)')
    ap.add_argument('--profile', type=int, default=0, help='afterwards: per-kernel device time (HIP events) over this many steps, per variant')
    ap.add_argument('paths', nargs='*')
    a = ap.parse_args()
    paths = a.paths or sorted(glob.glob(os.path.join(ROOT, 'chsimpy_amd', 'lib', 'variants', a.glob + '.so')))
    assert paths, 'no variants'
    if a.reverse:
        paths = paths[::-1]
    import chsimpy_amd
    engines = []
    mods = {}
    envs = [e for e in a.envs.split(';') if e] or ['']
    todo = [(path, env) for path in paths for env in envs] * a.copies
    for i, (path, env) in enumerate(todo):
        if path not in mods:
            mods[path] = bind(os.path.abspath(path), str(i))
        mod = mods[path]
        saved = {}
        for kv in [x for x in env.split(',') if x]:
            k, v = kv.split('=', 1)
            saved[k] = os.environ.get(k)
            os.environ[k] = v
        p = chsimpy_amd.Parameters()
        p.N, p.ntmax, p.full_sim, p.kappa_tilde, p.dtype, p.engine = a.grid, 10 ** 9, a.mode != 'estop', KAPPA, a.dtype, 'fast'
        if a.mode == 'adaptive':
            p.adaptive_time = True
            p.delt_max = a.delt_max or (6e-11 if a.grid == 8192 else 1.2e-10 * (4096 / a.grid))
        s = chsimpy_amd.Solver(p)
        c0 = s._consts()
        consts = mod.chs_consts()             # (the structure type of THIS binding)
        for name, _t in c0._fields_:
            setattr(consts, name, getattr(c0, name))
        eng = mod.Engine(consts, s.solution.lam)
        st = s._pcg_state0['state']
        eng.init_U_pcg64(p.XXX, p.XXX * 0.01, st['state'], st['inc'])
        eng.prepare()
        rows, rc = eng.step_n(max(a.warm, 520 if a.mode == 'adaptive' else 0))
        assert rc == 0, (path, rc)
        for k, v in saved.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
        engines.append((os.path.basename(path)[:-3] + (' ' + env if env else ''), eng))
    kw = dict(rederive_hat=True, last_call=False) if a.mode == 'literal' else {}
    ms = {n: [] for n, _ in engines}      # (several engines of one variant share a name: their times are pooled)
    wall = {n: [] for n, _ in engines}
    for r in range(a.rounds):
        order = engines if r % 2 == 0 else engines[::-1]     # (alternate the order: nobody always runs behind the same one)
        for n, eng in order:
            t0 = time.perf_counter()
            rows, rc = eng.step_n(a.steps, **kw)
            wall[n].append((time.perf_counter() - t0) * 1e3 / a.steps)
            assert rc == 0 and rows.shape[0] == a.steps, (n, rc, rows.shape)
            ms[n].append(eng.last_step_ms() / a.steps)
    ref = engines[-1][0] if a.reverse else engines[0][0]
    names = []
    for n, _ in engines:
        if n not in names:
            names.append(n)
    print(f"# N={a.grid} {a.dtype} mode={a.mode} steps/call={a.steps} rounds={a.rounds} copies={a.copies} (device ms per step; ratio = variant / {ref}, per round)")
    for n in names:
        ratios = [x / y for x, y in zip(ms[n], ms[ref])]
        print(f"{n:28s} median {statistics.median(ms[n]):.5f}  min {min(ms[n]):.5f}  wall median {statistics.median(wall[n]):.5f}  "
              f"ratio median {statistics.median(ratios):.4f}  [{min(ratios):.4f} .. {max(ratios):.4f}]  "
              f"-> {1e3 / statistics.median(ms[n]):.0f} steps/s")
    if a.profile:
        for rep in range(2):
            for n, eng in engines:
                msk, calls = eng.profile_steps(a.profile)
                names = eng.kernel_names()
                if rep == 1:
                    print(f"{n:28s} " + '  '.join(f"{names[i]} {msk[i] / calls[i] * 1e3:.1f}us" for i in range(len(names)) if names[i] and calls[i] > 0))
    for _, eng in engines:
        eng.close()


if __name__ == '__main__':
    main()
