#!/bin/bash
# whole frame and one rank's eighth of it for several hand-over points of the wavefront pipeline (RT_FINISH_BELOW)
mkdir -p gpurun_out
for fb in "$@"; do
  for sh in "" "--shard 0/8"; do
    RT_FINISH_BELOW=$fb timeout -k 10 200 python -u bench.py --steps 8 --warmup 2 --no-cpu-baseline $sh > gpurun_out/fs_$fb.log 2>&1 || { tail -5 gpurun_out/fs_$fb.log; exit 1; }
    python3 -c "import json; d=json.loads(open('gpurun_out/fs_$fb.log').read().strip().splitlines()[-1]); r=d['roofline']; print('finish_below $fb', '$sh' or 'whole', 'frame ms', round(d['ms_per_step'],2), 'k_rays ms', round(r['kernel_ms_per_frame'],2), 'launches', r['launches_per_frame'])"
  done
done
