#!/bin/bash
# tools/prof_quick.sh <tag> <scene:nx:ny:spp> [scene_bench options]: two SQ counter passes of tools/scene_bench.py on one scene, printed per render kernel
tag=$1; job=$2; shift 2
out=$PWD/gpurun_out/prof_$tag; mkdir -p $out
root=$PWD
cd /tmp && export TMPDIR=/tmp
B="python3 $root/tools/scene_bench.py $* $job"
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU --output-format csv -d $out/pmc1 -- $B > $out/pmc1.log 2>&1 || exit 1
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $out/pmc2 -- $B > $out/pmc2.log 2>&1 || exit 1
grep Mpaths $out/pmc1.log
python3 - $out <<'PY'
import csv, glob, os, sys
d = sys.argv[1]
for p in ("pmc1", "pmc2"):
    agg = {}
    for f in sorted(glob.glob(os.path.join(d, p, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            if "render_" in r["Kernel_Name"]:
                agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()): print(f"  {k:26s} {v[-1]:.5g}")
PY
