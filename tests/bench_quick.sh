#!/bin/bash
# tests/bench_quick.sh LABEL [env assignments...]: one line of the headline bench under the given environment
L=$1; shift
env "$@" timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS} 2> gpurun_out/q_$L.err | tail -1 > gpurun_out/q_$L.json || { echo "$L failed"; tail -5 gpurun_out/q_$L.err; exit 1; }
python - "$L" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/q_{sys.argv[1]}.json").read())
r = d["roofline"]
print(f"{sys.argv[1]:14s} {d['value']:8.1f} {d['unit']} {d['ms_per_step']:8.2f} ms/frame; {r['kernel']} {r['avg_launch_ms']*r.get('launches_per_frame',1):8.2f} ms in {r.get('launches_per_frame')} launches")
PY
