#!/bin/bash
# The profile sessions of a round on ONE GPU box: tools/round_profiles.sh <tag>   (writes gpurun_out/<tag>_*)
# N=4096 fp64 (headline), N=2048 fp64 (the ensemble's member size), N=8192 fp32 (configs[3]): kernel trace + PMC passes each,
# traffic.json keys, the ensemble beside its CPU comparator, a kernel trace of three concurrent members.
tag=${1:-rXX}
R=$GRAFT_REPO_ROOT
o=$R/gpurun_out
tools/prof.sh ${tag}_n4096 > $o/${tag}_prof_n4096.txt 2>&1
tools/prof.sh ${tag}_n2048 --grid 2048 > $o/${tag}_prof_n2048.txt 2>&1
tools/prof.sh ${tag}_n8192f --grid 8192 --dtype float32 > $o/${tag}_prof_n8192f.txt 2>&1
cp profiles/traffic.json $o/${tag}_traffic.json
python tools/prof_summary.py $o/prof_${tag}_n4096 --traffic $o/${tag}_traffic.json 4096 > /dev/null
python tools/prof_summary.py $o/prof_${tag}_n2048 --traffic $o/${tag}_traffic.json 2048 > /dev/null
python tools/prof_summary.py $o/prof_${tag}_n8192f --traffic $o/${tag}_traffic.json 8192 :f32
# per-kernel stats tables of the traces
for s in n4096 n2048 n8192f; do
  f=$(ls $o/prof_${tag}_$s/trace/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp $f $o/${tag}_${s}_kernel_stats.csv
done
timeout -k 10 400 python bench.py --ensemble-baseline > $o/${tag}_ensemble.json 2> $o/${tag}_ensemble.err
tail -c 600 $o/${tag}_ensemble.json
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_${tag}_ens3 -- python3 $R/tools/ens_bench.py 3 > $o/${tag}_ens3_trace.log 2>&1 )
f=$(ls $o/prof_${tag}_ens3/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $o/${tag}_ens3_kernel_stats.csv
tail -2 $o/${tag}_ens3_trace.log
