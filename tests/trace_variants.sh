#!/bin/bash
# run tests/bench_trace.py against tuning variants of the library: tests/trace_variants.sh name1 name2 ...
for v in "$@"; do
  if [ "$v" = base ]; then unset RT_LIB; else export RT_LIB=$PWD/raytracer_amd/variants/librt_$v.so; fi
  echo "== $v"
  timeout -k 10 200 python -u tests/bench_trace.py --spp 6 $TRACE_ARGS 2>&1 | tail -3 || { echo "$v FAILED"; exit 1; }
done
