#!/bin/bash
# tools/pmc_variants.sh <job>: SQ instruction counters of tools/scene_bench.py <job> for every library in ray_tracing_amd/build_var
job=$1
out=$PWD/gpurun_out/pmcv; mkdir -p $out
root=$PWD
cd /tmp && export TMPDIR=/tmp
for so in $root/ray_tracing_amd/build_var/librt_*.so; do
  tag=$(basename $so .so)
  export RT_AMD_LIB=$so
  timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $out/$tag -- python3 $root/tools/scene_bench.py $job > $out/$tag.log 2>&1 || exit 1
  echo "== $tag"; grep Mpaths $out/$tag.log
  python3 - $out/$tag <<'PY'
import csv,glob,sys,os
agg={}
for f in glob.glob(os.path.join(sys.argv[1],"**","*counter_collection.csv"),recursive=True):
    for r in csv.DictReader(open(f)):
        if "render_" in r["Kernel_Name"]:
            agg.setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
for k,v in sorted(agg.items()): print(f"  {k:24s} {sum(v)/len(v):.5g}")
PY
done
