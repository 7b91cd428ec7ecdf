#!/bin/bash
# rocprofv3 session on the GPU box: kernel trace + separate PMC passes.
# usage: tools/prof.sh <tag> [bench args...]
set -o pipefail
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
args="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras $@"  # the driver's invocation (without the CPU baseline leg)
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/trace -- python3 $R/bench.py $args > $R/$out/trace.log 2>&1 || exit 1
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_UNALIGNED_STALL SQ_INSTS_FLAT SQ_VALU_MFMA_BUSY_CYCLES"; do
  n=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/$out/pmc_$n -- python3 $R/bench.py $args > $R/$out/pmc_$n.log 2>&1 || echo "pass $n failed"
done
python3 $R/tools/prof_summary.py $R/$out > $R/$out/summary.txt 2>&1
cat $R/$out/summary.txt
