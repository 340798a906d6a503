J="random_spheres:1200:800:64 next_week_final:800:800:128 sphere_field_1m:1920:1080:64 cornell_box_volumes:512:512:256 sphere_field_4k:512:512:32 simple_scene:512:512:64 another_simple:256:256:16"
for i in 1 2; do
echo "== new"; timeout -k 10 300 python tools/scene_bench.py $J || exit 1
echo "== prev"; RT_AMD_LIB=ray_tracing_amd/build_var/librt_prev.so timeout -k 10 300 python tools/scene_bench.py $J || exit 1
done
