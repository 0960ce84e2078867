#!/bin/bash
# whole frame and one rank's eighth of it for the library variants named on the command line
mkdir -p gpurun_out
for v in "$@"; do
  if [ "$v" != base ]; then export RT_LIB=$PWD/raytracer_amd/variants/librt_$v.so; else unset RT_LIB; fi
  for sh in "" "--shard 0/8"; do
    timeout -k 10 200 python -u bench.py --steps 8 --warmup 2 --no-cpu-baseline $sh > gpurun_out/sv_$v.log 2>&1 || { tail -5 gpurun_out/sv_$v.log; exit 1; }
    python3 -c "import json; d=json.loads(open('gpurun_out/sv_$v.log').read().strip().splitlines()[-1]); r=d['roofline']; print('$v', '$sh' or 'whole', 'frame ms', round(d['ms_per_step'],2), 'k_rays ms', round(r['kernel_ms_per_frame'],2), 'launches', r['launches_per_frame'])"
  done
done
