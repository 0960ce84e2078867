#!/bin/bash
# tests/prof_trace.sh OUTNAME POPULATION [variant]: SQ counter passes of the ray-list kernel on one ray population
set -e
OUT=$PWD/gpurun_out/$1; POP=$2; V=${3:-base}
if [ "$V" != base ]; then export RT_LIB=$PWD/raytracer_amd/variants/librt_$V.so; fi
export TMPDIR=/tmp
B="python3 tests/bench_trace.py --spp 6 --reps 1 --only $POP"
mkdir -p $OUT
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $OUT/a -o p -- $B > $OUT/a.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INSTS_BRANCH --output-format csv -d $OUT/b -o p -- $B > $OUT/b.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $OUT/c -o p -- $B > $OUT/c.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/d -o p -- $B > $OUT/d.log 2>&1
python3 - $OUT <<'PY'
import csv,glob,collections,sys
agg=collections.defaultdict(float)
for f in sorted(glob.glob(sys.argv[1]+'/*/**/*counter_collection.csv',recursive=True)):
    for r in csv.DictReader(open(f)):
        if 'k_rays' in r['Kernel_Name']: agg[r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in sorted(agg.items()): print(f"{k:28s} {v:.4g}")
PY
