#!/bin/bash
# one GPU call: threshold variants of the ray-list kernel, then the culling-vote variants of the Whitted kernel
tests/bench_quick.sh base || exit 1
for v in leaf8 leaf32 bvh16 bvh48 scene32 refill8 refill40; do tests/bench_quick.sh $v RT_LIB=$PWD/raytracer_amd/variants/librt_$v.so || exit 1; done
echo "== whitted tests, cull variant"
RT_LIB=$PWD/raytracer_amd/variants/librt_cull.so timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 180 --timeout-method thread -k "whitted or photon or textured" 2>&1 | tail -8
echo "== violations (cullcheck)"
for w in spheres_1080p_16spp_whitted photon_box_540p_small; do
  RT_LIB=$PWD/raytracer_amd/variants/librt_cullcheck.so timeout -k 10 300 python bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'], d['ms_per_step'], d.get('phase_cycles'))"
done
BENCH_ARGS="--workload spheres_1080p_16spp_whitted" tests/bench_quick.sh sph_base
BENCH_ARGS="--workload spheres_1080p_16spp_whitted" tests/bench_quick.sh sph_cull RT_LIB=$PWD/raytracer_amd/variants/librt_cull.so
