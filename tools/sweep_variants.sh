#!/bin/bash
# runs tools/scene_bench.py once per library variant in ray_tracing_amd/build_var (tools/build_variant.sh)
for so in ray_tracing_amd/build_var/librt_*.so; do
  echo "== $so"
  RT_AMD_LIB=$PWD/$so timeout -k 10 200 python tools/scene_bench.py "$@" 2>&1 | grep Mpaths || exit 1
done
