"""Per-burst summary of a rocprofv3 kernel trace: bursts separated by > 1 ms of idle; kernel time vs span, launches, median gap."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:40]) for r in csv.DictReader(open(f))))
bursts, cur = [], [rows[0]]
for a in rows[1:]:
    if a[0] - cur[-1][1] > 1_000_000: bursts.append(cur); cur = [a]
    else: cur.append(a)
bursts.append(cur)
for b in bursts:
    if len(b) < 500: continue
    span = (b[-1][1] - b[0][0]) / 1e6
    busy = sum(e - s for s, e, _ in b) / 1e6
    gaps = sorted(b[i + 1][0] - b[i][1] for i in range(len(b) - 1))
    durs = {}
    for s, e, n in b: durs.setdefault(n, []).append(e - s)
    top = sorted(durs.items(), key=lambda kv: -sum(kv[1]))[:3]
    print(f"burst of {len(b)} launches: span {span:.1f} ms, kernels {busy:.1f} ms, median gap {gaps[len(gaps)//2]/1e3:.1f} us, p90 gap {gaps[int(len(gaps)*.9)]/1e3:.1f} us; " +
          '; '.join(f"{n} x{len(v)} avg {sum(v)/len(v)/1e3:.1f} us" for n, v in top))
