#!/bin/bash
# tests/gpu_shard_balance.sh [WORLD]: what each rank of a WORLD-GPU run of the headline frame does, timed on one GPU (bench.py --shard R/WORLD)
W=${1:-8}
mkdir -p gpurun_out
for r in $(seq 0 $((W - 1))); do
    timeout -k 10 120 python -u bench.py --shard $r/$W --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/shard_$r.log 2>&1 || { tail -5 gpurun_out/shard_$r.log; exit 1; }
    python3 -c "import json; d=json.loads([l for l in open('gpurun_out/shard_$r.log') if l.startswith('{')][-1]); print('shard $r/$W frame ms', round(d['ms_per_step'],2), 'rays', d['config'].get('rays_per_frame'))"
done
