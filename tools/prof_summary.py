"""Condenses a tools/prof.sh session into per-kernel averages (time + counters)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    for k in ('k_row_fwd', 'k_row_inv', 'k_col', 'k_diag', 'k_fin', 'k_pre', 'k_mu', 'k_spectral', 'k_gemm', 'k_sum'):
        if k in name:
            if k == 'k_col':
                # template mode is the last integer parameter
                return k
            return k
    return name[:40]


def main(d):
    # kernel trace
    rows = []
    for f in glob.glob(os.path.join(d, 'trace', '**', '*kernel_trace.csv'), recursive=True):
        rows += list(csv.DictReader(open(f)))
    dur = defaultdict(list)
    meta = {}
    for r in rows:
        n = short(r['Kernel_Name'])
        dur[n].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
        meta[n] = (r.get('VGPR_Count'), r.get('Accum_VGPR_Count'), r.get('SGPR_Count'), r.get('LDS_Block_Size'),
                   r.get('Scratch_Size'), r.get('Grid_Size'), r.get('Workgroup_Size'))
    print("== kernel trace (ns): name calls avg min  | vgpr agpr sgpr lds scratch grid wg")
    for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        v2 = sorted(v)
        core = v2[len(v2) // 10: max(len(v2) * 9 // 10, 1)] or v2
        print(f"{n:14s} {len(v):6d} avg {sum(core) / len(core):10.0f} min {v2[0]:9d} | {meta[n]}")
    # counters
    agg = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, 'pmc_*', '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            agg[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
    print("== counters (average per dispatch)")
    for n in agg:
        print(n)
        for c, v in sorted(agg[n].items()):
            print(f"    {c:28s} {sum(v) / len(v):16.1f}   (n={len(v)})")


if __name__ == '__main__' and not (len(sys.argv) > 2 and sys.argv[2] == '--traffic'):
    main(sys.argv[1])


def traffic(d, out_json, N=4096, suffix=''):
    """Adds to profiles/traffic.json from a session: L2<->fabric bytes per launch of the two per-step kernels, from the
    FETCH_SIZE / WRITE_SIZE passes, corrected as MI355X_MICROARCH.md (HBM section) prescribes.  Keys
    `fast:<kernel>:N<N><suffix>` (suffix ':f32' for the fp32 engine) -- what bench.py looks up for `roofline.traffic`."""
    import json
    agg = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, 'pmc_*', '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] in ('FETCH_SIZE', 'WRITE_SIZE'):
                agg[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
    note = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/prof.sh); FETCH_SIZE doubled (gfx950 counts 64 B "
            "per 128-B request of a 16 B/lane stream, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported; average over "
            "all dispatches of the kernel template (incl. the lighter entry-transform / last-step variants). These are "
            "L2<->fabric bytes: requests served by the 256 MB Infinity Cache are included, so this is an upper bound of the HBM traffic")
    out = {}
    if os.path.exists(out_json):
        try:
            out = json.load(open(out_json))
        except Exception:
            out = {}
    for kern, key in (('k_col', 'k_col'), ('k_row_inv', 'k_row_inv (fused)')):
        if kern in agg and agg[kern]['FETCH_SIZE'] and agg[kern]['WRITE_SIZE']:
            fk = sum(agg[kern]['FETCH_SIZE']) / len(agg[kern]['FETCH_SIZE'])
            wk = sum(agg[kern]['WRITE_SIZE']) / len(agg[kern]['WRITE_SIZE'])
            out[f'fast:{key}:N{N}{suffix}'] = {'hbm_bytes_per_launch': int(2 * fk * 1024 + wk * 1024), 'FETCH_SIZE_KB': round(fk, 1),
                                               'WRITE_SIZE_KB': round(wk, 1), 'session': os.path.basename(d.rstrip('/')), 'note': note}
    json.dump(out, open(out_json, 'w'), indent=1)
    print(json.dumps({k: v['hbm_bytes_per_launch'] for k, v in out.items()}, indent=1))


if __name__ == '__main__' and len(sys.argv) > 3 and sys.argv[2] == '--traffic':
    traffic(sys.argv[1], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 4096, sys.argv[5] if len(sys.argv) > 5 else '')
