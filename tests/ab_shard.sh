#!/bin/bash
# tests/ab_shard.sh R/W LIB...: bench.py --shard R/W (and the whole frame) through each of the given builds of the library
sh=$1; shift
for lib in "$@"; do
    echo "== $lib"
    for a in "--shard $sh" ""; do
        RT_LIB=$PWD/$lib timeout -k 10 200 python -u bench.py $a --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('  $a frame ms', round(d['ms_per_step'],2))" || exit 1
    done
done
