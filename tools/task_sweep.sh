#!/bin/bash
for spp in ${SPPS:-1024}; do for t in ${TASKS:-131072 262144 524288 1048576}; do
RT_AMD_TARGET_TASKS=$t timeout -k 10 200 python bench.py --spp $spp --steps 2 --warmup 1 --no-cpu 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('spp $spp tasks $t', j['value'], 'Mpaths/s kernel_ms', j['roofline']['kernel_ms'])" || exit 1
done; done
