#!/bin/bash
# tests/prof_pmc.sh OUTNAME [variant]: kernel-trace stats + three separate PMC passes of one bench run
set -e
OUT=gpurun_out/$1; V=${2:-base}
if [ "$V" != base ]; then export RT_LIB=$PWD/raytracer_amd/variants/librt_$V.so; fi
export TMPDIR=/tmp
B="python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $B > $OUT.trace.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc1 -o p -- $B > $OUT.pmc1.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc2 -o p -- $B > $OUT.pmc2.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/pmc3 -o p -- $B > $OUT.pmc3.log 2>&1
find $OUT -name "*.csv" | head -20
