#!/bin/bash
# Diagnostic: board power and clocks while the headline loop runs (is the step power-limited?).
# usage: tools/power_sample.sh [bench args...]
mkdir -p gpurun_out
python bench.py --no-cpu-baseline --steps 30000 --warmup 300 "$@" > gpurun_out/power_bench.json 2> gpurun_out/power_bench.err &
pid=$!
: > gpurun_out/power_smi.txt
while kill -0 $pid 2>/dev/null; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | tr '\n' ' ' >> gpurun_out/power_smi.txt
  echo >> gpurun_out/power_smi.txt
  sleep 0.5
done
wait $pid
tail -c 300 gpurun_out/power_bench.json; echo
sort gpurun_out/power_smi.txt | uniq -c | sort -rn | head -30
