#!/bin/bash
# tools/prof_ta.sh <scene:nx:ny:spp> [scene_bench options]: address-unit and L1 counters of one scene (is a per-lane walk bound by the texture-address path?)
job=$1; shift
out=$PWD/gpurun_out/prof_ta; rm -rf $out; mkdir -p $out
root=$PWD
cd /tmp && export TMPDIR=/tmp
B="python3 $root/tools/scene_bench.py $* $job"
timeout -s KILL 150 rocprofv3 --pmc TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TD_TD_BUSY_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d $out/p1 -- $B > $out/p1.log 2>&1 || { tail -5 $out/p1.log; exit 1; }
timeout -s KILL 150 rocprofv3 --pmc TA_BUSY_avr TA_BUSY_max TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum GRBM_GUI_ACTIVE --output-format csv -d $out/p2 -- $B > $out/p2.log 2>&1 || { tail -5 $out/p2.log; exit 1; }
grep Mpaths $out/p1.log
python3 - $out <<'PY'
import csv, glob, os, sys
d = sys.argv[1]
for p in ("p1", "p2"):
    agg = {}
    for f in sorted(glob.glob(os.path.join(d, p, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            if "render_" in r["Kernel_Name"]:
                agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()): print(f"  {k:40s} {v[-1]:.5g}")
PY
