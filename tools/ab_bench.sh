#!/bin/bash
# A/B timing of library builds: tools/ab_bench.sh <spp> lib1.so lib2.so ...   (runs on the GPU box)
spp=$1; shift
for lib in "$@"; do
  for rep in 1 2; do
    RT_AMD_LIB=$PWD/$lib timeout -k 10 200 python bench.py --spp $spp --steps 3 --warmup 1 --no-cpu 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', j['value'], 'Mpaths/s kernel_ms', j['roofline']['kernel_ms'])" || exit 1
  done
done
