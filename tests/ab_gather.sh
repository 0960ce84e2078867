#!/bin/bash
# tests/ab_gather.sh LIB...: the batch photon lookup (tests/bench_photon.py --gather-only) through each of the given builds of the library
for lib in "$@"; do
    echo "== $lib"
    RT_LIB=$PWD/$lib timeout -k 10 120 python -u tests/bench_photon.py --photons 1000000 --caustic 100000 --gather-only 450000 2>&1 | grep "^gather" || exit 1
done
