#!/bin/bash
# A/B of two builds of the library over the per-lane scenes, in ONE GPU call (two boxes differ by about 5 %, so
# numbers from two calls do not compare).  The tree's library is "new"; the other one is a variant built by
# tools/build_variant.sh (for the previous commit: git stash; tools/build_variant.sh prev; git stash pop).
#   tools/gpu.sh ab 'bash tools/ab_scenes.sh prev > gpurun_out/ab.log 2>&1' 900
#   awk '/==/{tag=$2; next} {s[tag" "$1]+=$5; n[tag" "$1]++} END{for(k in s) print k, s[k]/n[k]}' gpurun_out/ab.log | sort -k2
V=${1:-prev}
J="random_spheres:1200:800:64 next_week_final:800:800:128 sphere_field_1m:1920:1080:64 cornell_box_volumes:512:512:256 sphere_field_4k:512:512:32 simple_scene:512:512:64 another_simple:256:256:16"
for i in 1 2; do
echo "== new"; timeout -k 10 300 python tools/scene_bench.py $J || exit 1
echo "== $V"; RT_AMD_LIB=ray_tracing_amd/build_var/librt_$V.so timeout -k 10 300 python tools/scene_bench.py $J || exit 1
done
