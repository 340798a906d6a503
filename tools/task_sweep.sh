#!/bin/bash
# sweep of RT_OPT_TARGET_TASKS on the headline scene (tools/scene_bench.py --opt)
for spp in ${SPPS:-1024}; do for t in ${TASKS:-131072 262144 524288 1048576}; do
echo -n "spp $spp tasks $t: "; timeout -k 10 200 python tools/scene_bench.py --opt TARGET_TASKS=$t cornell_box:1024:1024:$spp || exit 1
done; done
