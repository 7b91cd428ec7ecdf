#!/bin/bash
# A/B of prebuilt library variants with an arbitrary script instead of bench.py:
#   tools/ab_script.sh <rounds> <python script> [args...]   (prints the script's output per variant, interleaved)
rounds=$1; shift
cp chsimpy_amd/lib/libchs_hip.so /tmp/ab_script_default.so
for r in $(seq $rounds); do
  for v in chsimpy_amd/lib/variants/${AB_GLOB:-*}.so; do
    cp $v chsimpy_amd/lib/libchs_hip.so
    echo "== $(basename $v)"
    timeout -k 10 300 python "$@" 2>&1 | tail -${AB_TAIL:-4}
  done
done
cp /tmp/ab_script_default.so chsimpy_amd/lib/libchs_hip.so
