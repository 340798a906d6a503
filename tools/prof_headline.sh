#!/bin/bash
# tools/prof_headline.sh <tag> [config]: rocprofv3 on the DEFAULT bench.py command (Cornell 1024x1024x1024spp), or on
# bench.py --config <c1|c3|c5>: kernel-trace stats, then separate PMC passes (SQ pipeline, HBM read, HBM write).
# Results under gpurun_out/prof_<tag>/.
tag=$1; cfg=${2:-c2}
out=$PWD/gpurun_out/prof_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --config $cfg --steps 2 --warmup 1 --no-cpu --no-extra"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $B > $out/trace.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $out/pmc1 -- $B > $out/pmc1.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_WAVES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $out/pmc2 -- $B > $out/pmc2.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $out/pmc3 -- $B > $out/pmc3.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc4 -- $B > $out/pmc4.log 2>&1 || exit 1
echo done
