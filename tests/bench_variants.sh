#!/bin/bash
# run bench.py against each tuning variant of the library: tests/bench_variants.sh name1 name2 ...
# BENCH_ARGS adds bench.py flags
for v in "$@"; do
  if [ "$v" = base ]; then unset RT_LIB; else export RT_LIB=$PWD/raytracer_amd/variants/librt_$v.so; fi
  timeout -k 10 200 python -u bench.py --steps 3 --warmup 1 --no-cpu-baseline $BENCH_ARGS > gpurun_out/bench_$v.log 2>&1 || { echo "$v FAILED"; tail -3 gpurun_out/bench_$v.log; exit 1; }
  python - "$v" <<'PY'
import json,sys
v=sys.argv[1]
d=json.loads(open(f"gpurun_out/bench_{v}.log").read().strip().splitlines()[-1])
print(f"{v:12s} {d['value']:9.1f} Mrays/s  {d['ms_per_step']:8.2f} ms/frame  launch {d['roofline']['avg_launch_ms']:8.2f} ms  rays/frame {d['config']['rays_per_frame']/1e6:.1f}M  hbm frac {d['roofline']['frac']:.3f}")
PY
done
