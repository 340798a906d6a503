#!/bin/bash
# tools/build_variant.sh <tag> <extra hipcc flags...>: ray_tracing_amd/build_var/librt_<tag>.so with other compile-time settings,
# for A/B runs (RT_AMD_LIB=<that file> python tools/scene_bench.py ...).  The host objects come from the normal build.
set -e
tag=$1; shift
cd "$(dirname "$0")/../ray_tracing_amd"
make -s librt_amd.so
mkdir -p build_var/data && cp -u data/earth_1024x512.rgb build_var/data/
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function \
    -Wno-bitwise-instead-of-logical "$@" -c csrc/rt_amd.hip -o build_var/rt_amd_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build_var/rt_amd_$tag.o build/rt_error.o build/rt_records.o build/rt_order.o \
    build/rt_bvh.o build/rt_host.o build/rt_flatten.o build/rt_multi.o -ldl -o build_var/librt_$tag.so
echo built build_var/librt_$tag.so
