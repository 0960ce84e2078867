#!/bin/bash
# tests/prof_l1.sh TAG: how busy the vector memory path (address unit, L1, data return) is in the dominant kernel of the default bench command
set -e
OUT=$PWD/gpurun_out/$1
mkdir -p $OUT
export TMPDIR=/tmp
B1="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline"
timeout -s KILL 200 rocprofv3 --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -o p -- $B1 > $OUT/a.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum TA_FLAT_READ_WAVEFRONTS_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -o p -- $B1 > $OUT/b.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for sub in ("a", "b"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            k = "k_rays<false" if "k_rays<false" in k else "k_wf_shade" if "k_wf_shade" in k else None
            if k: acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in acc.items():
        print(sub, k, {n: f"{x:.4g}" for n, x in v.items()})
PY
