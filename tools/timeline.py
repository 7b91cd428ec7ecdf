"""Diagnostic: timeline of the kernels of one bench call from a rocprofv3 --kernel-trace CSV.
usage: python tools/timeline.py <dir with *_kernel_trace.csv> [n_last_calls]
Prints, for the longest run of dispatches that ends with the timed call, each kernel's start offset,
duration and the idle gap in front of it."""
import csv
import glob
import sys

d = sys.argv[1]
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ev = [(r['Kernel_Name'].split('(')[0].split('<')[0][-28:], int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
# calls are separated by k_call_begin
idx = [i for i, e in enumerate(ev) if 'k_call_begin' in e[0]]
print('dispatches', len(ev), 'calls', len(idx))
for ci in range(len(idx)):
    a = idx[ci]
    b = idx[ci + 1] if ci + 1 < len(idx) else len(ev)
    seg = ev[a:b]
    n_col = sum(1 for e in seg if 'k_col' in e[0])
    span = (seg[-1][2] - seg[0][1]) / 1e3
    busy = sum(e[2] - e[1] for e in seg) / 1e3
    print(f'call {ci}: {len(seg)} dispatches, {n_col} k_col, span {span:.1f} us, busy {busy:.1f} us, idle {span-busy:.1f} us')
want = int(sys.argv[2]) if len(sys.argv) > 2 else None
if want is not None:
    a = idx[want]
    b = idx[want + 1] if want + 1 < len(idx) else len(ev)
    t0 = ev[a][1]
    prev_end = t0
    for name, s, e in ev[a:b]:
        print(f'{name:30s} start {((s - t0) / 1e3):9.1f} us  dur {((e - s) / 1e3):7.1f}  gap {((s - prev_end) / 1e3):6.1f}')
        prev_end = e
