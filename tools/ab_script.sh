#!/bin/bash
# A/B of prebuilt library variants with an arbitrary script instead of bench.py:
#   tools/ab_script.sh <rounds> <python script> [args...]   (prints the script's output per variant, interleaved)
# (variants are selected through CHS_LIB_PATH; the product library is never overwritten)
rounds=$1; shift
for r in $(seq $rounds); do
  for v in chsimpy_amd/lib/variants/${AB_GLOB:-*}.so; do
    echo "== $(basename $v)"
    CHS_LIB_PATH=$PWD/$v timeout -k 10 300 python "$@" 2>&1 | grep -v "CHS_LIB_PATH ->" | tail -${AB_TAIL:-4}
  done
done
