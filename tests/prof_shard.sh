#!/bin/bash
# tests/prof_shard.sh R/W [TAG]: kernel timeline of one rank's share of the headline frame (bench.py --shard) under rocprofv3
set -e
S=$1; TAG=${2:-shard}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 200 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --shard $S > $OUT/bench.json 2> $OUT/bench.err
python3 -c "import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print('shard $S: frame ms', round(d['ms_per_step'],2), 'k_rays ms', round(d['roofline']['kernel_ms_per_frame'],2), 'launches', d['roofline']['launches_per_frame'], 'exchange', d.get('frame_exchange_pack_unpack_ms'))"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --shard $S > $OUT/bench_under_trace.json 2> $OUT/trace.log
python3 tests/prof_trace_timeline.py $OUT/trace 60
