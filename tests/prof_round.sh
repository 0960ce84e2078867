#!/bin/bash
# tests/prof_round.sh TAG: the evidence behind bench.py's roofline object, written to gpurun_out/TAG/
#   1. rocprofv3 --kernel-trace --stats of the default bench command
#   2. HBM traffic: FETCH_SIZE and WRITE_SIZE in separate counter passes
set -e
OUT=$PWD/gpurun_out/$1
mkdir -p $OUT
export TMPDIR=/tmp
B="python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $B > $OUT/bench_under_trace.json 2> $OUT/trace.log
B1="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- $B1 > $OUT/fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- $B1 > $OUT/write.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $OUT/cache -o p -- $B1 > $OUT/cache.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $OUT/rdreq -o p -- $B1 > $OUT/rdreq.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum --output-format csv -d $OUT/wrreq -o p -- $B1 > $OUT/wrreq.log 2>&1
find $OUT -name "*.csv" | sort
