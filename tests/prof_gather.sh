#!/bin/bash
# tests/prof_gather.sh TAG: counters of the batch irradiance kernel (tests/bench_photon.py --gather-only) -> gpurun_out/TAG/
set -e
OUT=$PWD/gpurun_out/$1
mkdir -p $OUT
export TMPDIR=/tmp
B="python3 tests/bench_photon.py --photons 1000000 --caustic 100000 --gather-only 2000000"
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/sq -o p -- $B > $OUT/sq.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_WAVES SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/sq2 -o p -- $B > $OUT/sq2.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $OUT/cache -o p -- $B > $OUT/cache.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- $B > $OUT/fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- $B > $OUT/write.log 2>&1
python3 - <<PY
import csv, glob
sums = {}
for f in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_photon_irradiance" in r["Kernel_Name"]:
            d = sums.setdefault(r["Dispatch_Id"], {})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0) + float(r["Counter_Value"])
for k in sorted(sums, key=int):
    print(k, {a: "%.4g" % b for a, b in sorted(sums[k].items())})
PY
