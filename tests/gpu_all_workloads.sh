#!/bin/bash
# one bench line per workload (no CPU baseline), logs under gpurun_out/wl_*.log
mkdir -p gpurun_out
for w in scene_a_720p_128spp spheres_1080p_16spp_whitted photon_box_540p_small mesh_10m_4k_adaptive; do
  timeout -k 10 400 python -u bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/wl_$w.log 2>&1 || { tail -5 gpurun_out/wl_$w.log; exit 1; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/wl_$w.log').read().strip().splitlines()[-1])
r=d['roofline']
print('$w', 'ms', round(d['ms_per_step'],2), 'Mrays/s', round(d['value'],1), 'rays', d['config'].get('rays_per_frame'), 'kernel', r['kernel'], 'frac', round(r['frac'],3), 'ach', round(r['achieved'],1), r['unit'], 'launches', r.get('launches_per_frame'), 'valu', r.get('valu_fp32',{}).get('frac'))
"
done
