#!/bin/bash
# Whitted kernel at 2 / 3 / 4 waves per SIMD: sphere/plane frame and the development-size photon frame
mkdir -p gpurun_out
for v in base w3 w2; do
  if [ "$v" != base ]; then export RT_LIB=$PWD/raytracer_amd/variants/librt_$v.so; else unset RT_LIB; fi
  for w in spheres_1080p_16spp_whitted photon_box_540p_small; do
    timeout -k 10 200 python -u bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/wo_${v}_$w.log 2>&1 || { tail -5 gpurun_out/wo_${v}_$w.log; exit 1; }
    python3 -c "import json; d=json.loads(open('gpurun_out/wo_${v}_$w.log').read().strip().splitlines()[-1]); print('$v', '$w', 'frame ms', round(d['ms_per_step'],1))"
  done
done
