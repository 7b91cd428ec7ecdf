#!/bin/bash
# A/B timing of prebuilt library variants on one GPU box (box-to-box clocks differ by a few %):
#   tools/ab.sh [rounds] [bench args...]  -- runs bench.py with every chsimpy_amd/lib/variants/*.so, interleaved
#   (AB_GLOB='[ab]_*' restricts the variants).  The variants are selected through CHS_LIB_PATH; the product library
#   is never overwritten (an interrupted A/B cannot leave a variant behind as the product).
rounds=${1:-2}
shift
args=${@:---steps 300 --warmup 300}
mkdir -p gpurun_out
: > gpurun_out/ab.log
for r in $(seq $rounds); do
  for v in chsimpy_amd/lib/variants/${AB_GLOB:-*}.so; do
    CHS_LIB_PATH=$PWD/$v timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras $args > gpurun_out/ab_one.log 2>gpurun_out/ab_one.err || { echo "$v failed"; tail -3 gpurun_out/ab_one.err; continue; }
    python - "$v" <<'PY' | tee -a gpurun_out/ab.log
import json, sys
d = json.loads(open('gpurun_out/ab_one.log').read().strip().splitlines()[-1])
k = d['roofline']['kernel_ms']
print(f"{sys.argv[1].split('/')[-1]:24s} {d['value']:9.1f} steps/s  {d['ms_per_step']:.5f} ms  col {k.get('k_col',0):.5f} row {k.get('k_row_inv (fused)',0):.5f} tail {k.get('k_step_tail',0):.5f}")
PY
  done
done
