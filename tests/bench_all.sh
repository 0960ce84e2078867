#!/bin/bash
# tests/bench_all.sh TAG: every bench workload once (no CPU baseline), one line each, JSON lines kept in gpurun_out/TAG/
set -e
OUT=gpurun_out/$1; mkdir -p $OUT
for w in mesh_1m_1080p_64spp mesh_64k_540p_8spp mesh_10m_4k_adaptive scene_a_720p_128spp spheres_1080p_16spp_whitted spheres_1080p_16spp_path photon_box_540p_small; do
  timeout -k 10 400 python bench.py --workload $w --no-cpu-baseline > $OUT/$w.json 2> $OUT/$w.err || { echo "$w failed"; tail -n 5 $OUT/$w.err; exit 1; }
  python - "$OUT/$w.json" "$w" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(f"{sys.argv[2]:30s} {d['value']:9.1f} Mrays/s {d['ms_per_step']:9.2f} ms/frame  {r['kernel']} {r['kernel_ms_per_frame']:.2f} ms in {r['launches_per_frame']} launches", flush=True)
PY
done
