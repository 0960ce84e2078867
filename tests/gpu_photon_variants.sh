#!/bin/bash
# development-size photon workload and batch lookup timing for the library variants named on the command line
mkdir -p gpurun_out
for v in "$@"; do
  if [ "$v" != base ]; then export RT_LIB=$PWD/raytracer_amd/variants/librt_$v.so; else unset RT_LIB; fi
  echo "== $v"
  timeout -k 10 200 python -u bench.py --workload photon_box_540p_small --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/phv_$v.log 2>&1 || { tail -5 gpurun_out/phv_$v.log; exit 1; }
  python3 -c "import json,sys; d=json.loads(open('gpurun_out/phv_$v.log').read().strip().splitlines()[-1]); print('frame ms', round(d['ms_per_step'],1))"
  timeout -k 10 200 python -u tests/bench_photon.py --photons 1000000 --caustic 100000 --gather-only 500000 2>&1 | grep "^gather" || exit 1
done
