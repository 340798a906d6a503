#!/usr/bin/env python3
"""Summarises gpurun_out/prof_<tag>/ (tools/prof.sh) into a small text file for profiles/."""
import csv
import glob
import os
import sys

d = sys.argv[1]
out = []
for f in sorted(glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True)):
    out.append("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
    for r in csv.DictReader(open(f)):
        out.append(f"{r['Name'][:70]:70s} calls {r['Calls']:>4s} avg_ns {float(r['AverageNs']):>14.0f} total_ns {r['TotalDurationNs']:>14s} pct {r['Percentage']}")
for p in ("pmc1", "pmc2", "pmc3", "pmc4", "pmc5", "pmc6"):
    for f in sorted(glob.glob(os.path.join(d, p, "**", "*counter_collection.csv"), recursive=True)):
        agg = {}
        for r in csv.DictReader(open(f)):
            if "render_" not in r["Kernel_Name"]:
                continue
            k = r["Counter_Name"]
            agg.setdefault(k, []).append(float(r["Counter_Value"]))
        out.append(f"== {p}: render_kernel counters, mean per dispatch over {max(len(v) for v in agg.values()) if agg else 0} dispatches ==")
        for k, v in agg.items():
            out.append(f"{k:28s} {sum(v) / len(v):.6g}")
print("\n".join(out))
