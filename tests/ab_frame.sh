#!/bin/bash
# tests/ab_frame.sh WORKLOAD LIB...: bench.py on one workload through each of the given builds of the library
w=$1; shift
for lib in "$@"; do
    echo "== $lib"
    RT_LIB=$PWD/$lib timeout -k 10 300 python -u bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "^\{|Error|error" || exit 1
done
