#!/bin/bash
# the Whitted kernel's culling votes: every GPU test with the default build, then the Whitted / photon tests with builds
# whose register allocation and code layout differ (the earlier miscompares moved with unrelated code)
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread 2>&1 | tail -3 || exit 1
for v in w5 w3 check; do
  echo "== $v"
  RT_LIB=$PWD/raytracer_amd/variants/librt_$v.so timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py tests/test_full_size_parity.py -m gpu -q --timeout 180 --timeout-method thread -k "whitted or photon or textured" 2>&1 | tail -2
done
for w in spheres_1080p_16spp_whitted photon_box_540p_small; do
  BENCH_ARGS="--workload $w" tests/bench_quick.sh cull_$w
  BENCH_ARGS="--workload $w" tests/bench_quick.sh nocull_$w RT_LIB=$PWD/raytracer_amd/variants/librt_nocull.so
done
