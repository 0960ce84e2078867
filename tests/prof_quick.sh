#!/bin/bash
# tests/prof_quick.sh TAG [bench args]: per-kernel times of a short bench run under rocprofv3, written to gpurun_out/TAG/
set -e
TAG=$1; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/trace.log
python3 - "$OUT" <<'PY'
import csv, sys, json
out = sys.argv[1]
d = json.loads(open(out + "/bench.json").read().strip().splitlines()[-1])
print("ms_per_step", round(d["ms_per_step"], 2))
for r in csv.DictReader(open(out + "/trace/t_kernel_stats.csv")):
    if float(r["Percentage"]) > 0.3:
        print(f'{r["Name"][:60]:60s} calls {r["Calls"]:>5s} total {float(r["TotalDurationNs"])/1e6:9.2f} ms avg {float(r["AverageNs"])/1e6:8.3f} ms {r["Percentage"]:>6s} %')
PY
