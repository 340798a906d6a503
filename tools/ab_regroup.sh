set -e
J="random_spheres:1200:800:64 next_week_final:800:800:128 sphere_field_4k:512:512:32"
for r in 6 8 12 16 24; do echo "== block kernel rounds=$r"; timeout -k 10 200 python tools/scene_bench.py --opt LANES_REGROUP=1 --opt REGROUP_ROUNDS=$r $J; done
