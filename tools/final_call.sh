rm -f gpurun_out/parity.log
python -m pytest tests -m gpu -q > gpurun_out/r03_gputest_final.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_gputest_final.log; tail -3 gpurun_out/r03_gputest_final.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench_driver_final.json 2> gpurun_out/r03_bench_driver_final.err; tail -c 400 gpurun_out/r03_bench_driver_final.json
tools/prof.sh r03f_n8192f --grid 8192 --dtype float32 > gpurun_out/r03f_prof_n8192f.txt 2>&1
cp profiles/traffic.json gpurun_out/r03f_traffic.json
python tools/prof_summary.py gpurun_out/prof_r03f_n8192f --traffic gpurun_out/r03f_traffic.json 8192 :f32
f=$(ls gpurun_out/prof_r03f_n8192f/trace/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f gpurun_out/r03f_n8192f_kernel_stats.csv
python bench.py --grid 8192 --dtype float32 --steps 1000 --no-cpu-baseline > gpurun_out/r03f_bench_n8192_fp32.json 2>/dev/null; tail -c 700 gpurun_out/r03f_bench_n8192_fp32.json
python bench.py --energy-stop --no-cpu-baseline > gpurun_out/r03f_bench_energy_stop.json 2>/dev/null; tail -c 500 gpurun_out/r03f_bench_energy_stop.json
