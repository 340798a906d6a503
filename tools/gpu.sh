#!/bin/bash
# tools/gpu.sh <tag> '<command>' [timeout_s] -- one gpurun call from the build container; output tail to /tmp/gpu_<tag>.log
tag=$1; cmd=$2; to=${3:-600}
timeout $((to + 900)) /usr/local/graft/bin/gpurun --timeout $to -- "$cmd" > /tmp/gpu_$tag.log 2>&1
echo "rc=$?" >> /tmp/gpu_$tag.log
