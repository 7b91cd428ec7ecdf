#!/bin/bash
# One measurement session for DESIGN.md section 7: every configuration of the table on ONE box.
# usage: tools/measure_all.sh <tag>      (writes gpurun_out/<tag>_*.json / .txt)
tag=${1:-rXX}
o=gpurun_out
mkdir -p $o
run() { name=$1; shift; timeout -k 10 400 python bench.py "$@" > $o/${tag}_$name.json 2> $o/${tag}_$name.err || echo "$name failed"; }
run bench                                   # headline: N=4096 fp64, 300 + 5000 steps, with the CPU baseline
run bench_driver --steps 20 --warmup 5 --no-cpu-baseline   # the driver's invocation (literal solve_or_resume call)
run bench_driver_continue --steps 20 --warmup 5 --no-cpu-baseline --continue-loop
run bench_energy_stop --energy-stop --no-cpu-baseline
run bench_n2048 --grid 2048 --no-cpu-baseline
run bench_n1024 --grid 1024 --no-cpu-baseline
run bench_n512 --grid 512 --no-cpu-baseline
run bench_n128 --grid 128 --no-cpu-baseline
run bench_n8192 --grid 8192 --steps 1000 --no-cpu-baseline
run bench_n8192_fp32 --grid 8192 --dtype float32 --steps 1000 --no-cpu-baseline
run bench_n4096_fp32 --dtype float32 --no-cpu-baseline
run bench_n2048_fp32 --grid 2048 --dtype float32 --no-cpu-baseline
timeout -k 10 300 python tools/adaptive_bench.py > $o/${tag}_adaptive.txt 2>&1
timeout -k 10 300 python tools/jitter_bench.py > $o/${tag}_jitter.txt 2>&1
timeout -k 10 300 python tools/call_overhead.py > $o/${tag}_call_overhead.txt 2>&1
timeout -k 10 300 python tools/ens_bench.py > $o/${tag}_ens.txt 2>&1
python - <<PY
import json, glob
for f in sorted(glob.glob('$o/${tag}_bench*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d['roofline']
        print(f"{f.split('/')[-1]:34s} {d['value']:10.1f} steps/s {d['ms_per_step']:.5f} ms  whole {r['whole_step']['frac']:.4f}  {r['kernel']} {r['frac']:.4f}  {r['kernel_ms']}")
    except Exception as e:
        print(f, 'unreadable', e)
PY
cat $o/${tag}_adaptive.txt $o/${tag}_jitter.txt $o/${tag}_call_overhead.txt $o/${tag}_ens.txt
