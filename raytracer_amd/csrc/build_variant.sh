#!/bin/bash
# build_variant.sh NAME [-DFLAG...]  ->  raytracer_amd/variants/librt_NAME.so (kernel tuning experiments)
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
NAME=$1; shift
mkdir -p "$HERE/../variants"
make -C "$HERE" -j8 >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -Wno-unused-function \
  -I"$HERE/../../include" -I"$HERE" "$@" -c "$HERE/rt_api.hip" -o "$HERE/../variants/rt_api_$NAME.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$HERE/../variants/librt_$NAME.so" "$HERE/../variants/rt_api_$NAME.o" "$HERE"/host/*.o
rm -f "$HERE/../variants/rt_api_$NAME.o"
echo built librt_$NAME.so
