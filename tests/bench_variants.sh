#!/bin/bash
# bench_variants.sh NAME...: headline frame with each raytracer_amd/variants/librt_NAME.so (kernel tuning experiments)
for v in "$@"; do
  RT_LIB=raytracer_amd/variants/librt_$v.so timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/var_$v.json 2> gpurun_out/var_$v.err || { echo "$v failed"; tail -n 5 gpurun_out/var_$v.err; exit 1; }
  python - "$v" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/var_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(f"{sys.argv[1]:12s} {d['value']:9.1f} {d['unit']}  {d['ms_per_step']:8.2f} ms/frame")
PY
done
