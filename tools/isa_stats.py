#!/usr/bin/env python3
"""Per-kernel instruction statistics of a translation unit compiled for gfx950 (no GPU needed).
usage: tools/isa_stats.py chsimpy_amd/csrc/chs_fast_f32.hip [filter] [-D...]
Prints for every kernel whose (demangled) name contains `filter`: code bytes, VGPRs, scratch, and the
static counts of VALU / packed / fp64 / LDS / VMEM / SALU instructions of the fully unrolled body."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = sys.argv[1]
    flt = [a for a in sys.argv[2:] if not a.startswith('-')]
    extra = [a for a in sys.argv[2:] if a.startswith('-')]
    out = '/tmp/isa_stats_%d.s' % os.getpid()
    cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-Wno-unused-value', '-Wno-unused-result', '-Wno-pass-failed',
           '-I' + os.path.join(ROOT, 'include'), '-I' + os.path.join(ROOT, 'chsimpy_amd', 'csrc'),
           '--cuda-device-only', '-S', src, '-o', out] + extra
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    txt = open(out).read()
    os.unlink(out)
    # split into functions
    parts = re.split(r'\n(_Z\w+):[^\n]*\n', txt)
    names = parts[1::2]
    bodies = parts[2::2]
    dem = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.splitlines()
    for name, d, body in zip(names, dem, bodies):
        if flt and not all(f in d for f in flt):
            continue
        end = body.find('.Lfunc_end')
        if end < 0:
            continue
        code = body[:end]
        ins = [ln.strip().split()[0] for ln in code.splitlines() if ln.startswith('\t') and not ln.strip().startswith(('.', ';'))]
        n = len(ins)
        c = lambda pred: sum(1 for i in ins if pred(i))
        valu = c(lambda i: i.startswith('v_'))
        pk = c(lambda i: i.startswith('v_pk_'))
        f64 = c(lambda i: i.startswith('v_') and 'f64' in i)
        mov = c(lambda i: i.startswith('v_mov') or i.startswith('v_accvgpr'))
        lds = c(lambda i: i.startswith('ds_'))
        vmem = c(lambda i: i.startswith(('global_', 'buffer_', 'flat_', 'scratch_')))
        salu = c(lambda i: i.startswith('s_'))
        wait = c(lambda i: i.startswith('s_waitcnt'))
        bar = c(lambda i: i.startswith('s_barrier'))
        vg = re.search(r'; NumVgprs: (\d+)', body)
        sc = re.search(r'; ScratchSize: (\d+)', body)
        cl = re.search(r'; codeLenInByte = (\d+)', body)
        occ = re.search(r'; Occupancy: (\d+)', body)
        short = re.sub(r'FCfg<([^>]*)>', lambda m: 'FCfg<' + m.group(1).replace(' ', '') + '>', d)
        short = short.split('(')[0][-110:]
        print(f"{short}\n    code {cl.group(1) if cl else '?':>7} B  vgpr {vg.group(1) if vg else '?':>3}  scratch {sc.group(1) if sc else '?':>4}  occ {occ.group(1) if occ else '?'}"
              f"  | insts {n:6d}  valu {valu:6d} (pk {pk}, f64 {f64}, mov {mov})  lds {lds:4d}  vmem {vmem:4d}  salu {salu:5d} (wait {wait}, barrier {bar})")


if __name__ == '__main__':
    main()
