#!/bin/bash
# extra PMC pass: instruction cache and LDS/VMEM levels
set -o pipefail
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
args="--steps 20 --warmup 3 --no-cpu-baseline --profile-steps 2 $@"
for pass in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64" "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU"; do
  n=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/$out/pmc_$n -- python3 $R/bench.py $args > $R/$out/pmc_$n.log 2>&1 || echo "pass $n failed"
done
python3 $R/tools/prof_summary.py $R/$out > $R/$out/summary.txt 2>&1
