#!/usr/bin/env python3
"""profiles/counters.json from the committed rocprofv3 summaries (tools/prof_summary.py output) and the census counts:
per BASELINE config what the counters say about its render kernel, in the form bench.py attaches as roofline.counters.

    python tools/make_counters.py c2=profiles/r03_headline_fast_path.txt c1=... c3=... c5=...

valu_issue_frac   SQ_INSTS_VALU x 2 cycles / (1024 SIMDs x kernel cycles)   (a SIMD retires one wave64 FP32 instruction per
                  2 cycles at best: MI355X_MICROARCH.md; kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs, else duration x 2.4 GHz)
lane_util         SQ_THREAD_CYCLES_VALU / (64 x SQ_INSTS_VALU)
wait_frac         SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
gpu_rays_per_path rays the kernel traces per path (profiles/work_counts.json: the -DRT_COUNT_WALK / -DRT_COUNT_SHADE builds)
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENE = {"c2": "cornell_box", "c1": "random_spheres", "c3": "next_week_final", "c5": "sphere_field_1m"}


def parse(path):
    v, kernel, avg_ns = {}, None, None
    for ln in open(path):
        m = re.match(r"(void .*?render_\w+<[^>]*>).*avg_ns\s+([0-9.]+)", ln)
        if m and avg_ns is None:
            kernel, avg_ns = m.group(1).replace("void (anonymous namespace)::", ""), float(m.group(2))
        m = re.match(r"([A-Z][A-Za-z_0-9]+)\s+([0-9.e+]+)\s*$", ln)
        if m:
            v.setdefault(m.group(1), float(m.group(2)))
    return kernel, avg_ns, v


def main():
    counts = json.load(open(os.path.join(ROOT, "profiles", "work_counts.json")))
    out = {}
    for arg in sys.argv[1:]:
        cfg, path = arg.split("=")
        kernel, avg_ns, v = parse(os.path.join(ROOT, path))
        cycles = v["GRBM_GUI_ACTIVE"] / 8.0 if v.get("GRBM_GUI_ACTIVE") else avg_ns * 2.4  # the counter is summed over the 8 XCDs
        wc = counts.get(SCENE[cfg], {})
        rays = wc.get("gpu_rays_per_path") or (wc.get("own_tree_per_path") or {}).get("rays")
        out[cfg] = {
            "valu_issue_frac": round(v["SQ_INSTS_VALU"] * 2.0 / (1024.0 * cycles), 3),
            "lane_util": round(v["SQ_THREAD_CYCLES_VALU"] / (64.0 * v["SQ_INSTS_VALU"]), 3),
            "wait_frac": round(v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"], 3),
            "gpu_rays_per_path": rays,
            "kernel": kernel, "kernel_ms_profiled": round(avg_ns / 1e6, 3), "source": path,
        }
    json.dump(out, open(os.path.join(ROOT, "profiles", "counters.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
