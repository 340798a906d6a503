#!/bin/bash
# tools/prof_scene.sh <tag> <scene:nx:ny:spp>: rocprofv3 kernel trace + SQ counter passes of tools/scene_bench.py on one scene.
tag=$1; job=$2
out=$PWD/gpurun_out/prof_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/tools/scene_bench.py $job"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $B > $out/trace.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $out/pmc1 -- $B > $out/pmc1.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_WAVES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $out/pmc2 -- $B > $out/pmc2.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC --output-format csv -d $out/pmc3 -- $B > $out/pmc3.log 2>&1 || true
echo done
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $out/pmc4 -- $B > $out/pmc4.log 2>&1 || true
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $out/pmc5 -- $B > $out/pmc5.log 2>&1 || true
timeout -s KILL 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum --output-format csv -d $out/pmc6 -- $B > $out/pmc6.log 2>&1 || true
echo done2
