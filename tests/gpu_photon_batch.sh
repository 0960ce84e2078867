#!/bin/bash
# photon tests, the development-size photon workload and the batch lookup timing; logs under gpurun_out/
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q -k "photon or kd or full_size" --timeout 200 --timeout-method thread > gpurun_out/ph_test.log 2>&1 || { tail -30 gpurun_out/ph_test.log; exit 1; }
tail -2 gpurun_out/ph_test.log
timeout -k 10 300 python -u bench.py --workload photon_box_540p_small --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/ph_small.log 2>&1 || { tail -20 gpurun_out/ph_small.log; exit 1; }
tail -1 gpurun_out/ph_small.log
timeout -k 10 300 python -u tests/bench_photon.py --photons 1000000 --caustic 100000 --gather-only 500000 > gpurun_out/ph_gather.log 2>&1 || { tail -20 gpurun_out/ph_gather.log; exit 1; }
cat gpurun_out/ph_gather.log
