for n in 128 512 1024 2048; do
  for r in 1 2; do
    for g in 0 1; do
      if [ $g = 1 ]; then export CHS_GRAPH=1; else unset CHS_GRAPH; fi
      timeout -k 10 200 python bench.py --grid $n --steps 10240 --warmup 300 --no-cpu-baseline > gpurun_out/g_one.json 2> gpurun_out/g_one.err || { echo "N=$n graph=$g failed"; tail -2 gpurun_out/g_one.err; continue; }
      python - $n $g <<'PY'
import json,sys
d=json.loads(open('gpurun_out/g_one.json').read().strip().splitlines()[-1])
print(f"N={sys.argv[1]} graph={sys.argv[2]}: {d['value']:.1f} steps/s {d['ms_per_step']*1e3:.2f} us/step  E={d['energies_last_step'][0][0]:.6e}")
PY
    done
  done
done
