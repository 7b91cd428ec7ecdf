#!/bin/bash
# A/B timing of run-time switches of ONE library on one GPU box, interleaved:
#   tools/ab_env.sh <rounds> "<VAR=a VAR2=b>" "<VAR=c>" ... -- [bench args...]
rounds=$1; shift
sets=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do sets+=("$1"); shift; done
shift
args=${@:---steps 300 --warmup 300}
mkdir -p gpurun_out
: > gpurun_out/ab_env.log
for r in $(seq $rounds); do
  for s in "${sets[@]}"; do
    env $s timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras $args > gpurun_out/ab_one.log 2> gpurun_out/ab_one.err || { echo "[$s] failed"; tail -3 gpurun_out/ab_one.err; continue; }
    python - "$s" <<'PY' | tee -a gpurun_out/ab_env.log
import json, sys
d = json.loads(open('gpurun_out/ab_one.log').read().strip().splitlines()[-1])
k = d['roofline']['kernel_ms']
print(f"{sys.argv[1]:28s} {d['value']:9.1f} steps/s  {d['ms_per_step']:.5f} ms  other {d['config'].get('other_protocol_ms_per_step')}  col {k.get('k_col',0):.5f} row {k.get('k_row_inv (fused)',0):.5f} tail {k.get('k_step_tail',0):.5f}")
PY
  done
done
