set -o pipefail
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/pmc_f32; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for pass in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $out/pmc_$pass -- python3 $R/bench.py --grid 8192 --dtype float32 --steps 20 --warmup 3 --no-cpu-baseline --profile-steps 2 > $out/$pass.log 2>&1 || echo fail
done
python3 $R/tools/prof_summary.py $out | grep -A4 "^k_col$\|^k_row_inv$"
