#!/bin/bash
# tests/prof_workload.sh TAG WORKLOAD [STEPS]: kernel-trace statistics and memory-side traffic of one bench workload,
# written to gpurun_out/TAG/ (copy the summaries worth keeping to profiles/).
set -e
OUT=$PWD/gpurun_out/$1
W=$2
N=${3:-3}
mkdir -p $OUT
export TMPDIR=/tmp
B="python3 bench.py --workload $W --steps $N --warmup 1 --no-cpu-baseline"
timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $B > $OUT/bench_under_trace.json 2> $OUT/trace.log
B1="python3 bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline"
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- $B1 > $OUT/fetch.log 2>&1
timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- $B1 > $OUT/write.log 2>&1
timeout -s KILL 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/sq -o p -- $B1 > $OUT/sq.log 2>&1
find $OUT -name "*.csv" | sort
