#!/usr/bin/env python3
"""Scratch (spill) bytes, VGPRs and code size of every kernel of the two fast-engine translation units, compiled for
gfx950 (no GPU needed).  `--row-zero`: exit 1 unless every k_row_* instantiation at N = 4096 and N = 8192 -- the step
kernels of BASELINE.json's configs[2] and configs[3] and their once-per-call siblings -- has ScratchSize 0
(tests/test_host.py runs this)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernels(src):
    out = '/tmp/scratch_check_%d_%s.s' % (os.getpid(), os.path.basename(src))
    cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-Wno-unused-value', '-Wno-unused-result', '-Wno-pass-failed',
           '-DCHS_TEST_HOOKS=1', '-I' + os.path.join(ROOT, 'include'), '-I' + os.path.join(ROOT, 'chsimpy_amd', 'csrc'),
           '--cuda-device-only', '-S', src, '-o', out]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    txt = open(out).read()
    os.unlink(out)
    parts = re.split(r'\n(_Z\w+):[^\n]*\n', txt)
    names, bodies = parts[1::2], parts[2::2]
    dem = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.splitlines()
    res = []
    for d, body in zip(dem, bodies):
        sc = re.search(r'; ScratchSize: (\d+)', body)
        vg = re.search(r'; NumVgprs: (\d+)', body)
        cl = re.search(r'; codeLenInByte = (\d+)', body)
        if sc and '__global__' not in d and d.startswith('void k_'):
            short = re.sub(r'FCfg<([^>]*)>', lambda m: 'FCfg<' + m.group(1).replace(' ', '') + '>', d).split('(')[0]
            res.append((short, int(sc.group(1)), int(vg.group(1)) if vg else -1, int(cl.group(1)) if cl else -1))
    return res


def main():
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(2) as pool:   # the two element types compile side by side
        parts = list(pool.map(lambda tu: kernels(os.path.join(ROOT, 'chsimpy_amd', 'csrc', tu)), ('chs_fast_f64.hip', 'chs_fast_f32.hip')))
    rows = parts[0] + parts[1]
    bad = []
    for name, sc, vg, cl in rows:
        big = re.search(r'FCfg<(double|float),(4096|8192),', name) is not None
        if '--quiet' not in sys.argv or sc:
            print(f"{sc:5d} B scratch  {vg:4d} vgpr  {cl:6d} B code  {name[-120:]}")
        if big and name.startswith('void k_row_') and sc != 0:
            bad.append((name, sc))
    if '--row-zero' in sys.argv and bad:
        print('row kernels with scratch at N >= 4096:', bad)
        sys.exit(1)


if __name__ == '__main__':
    main()
