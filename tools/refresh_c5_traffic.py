#!/usr/bin/env python3
"""profiles/c5_traffic.json from the committed summary of config 5's rocprofv3 passes (profiles/r03_c5_sphere_field_1m_1024spp.txt)
and the census (profiles/work_counts.json): the memory-side bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, in KiB in the
summary: the gfx950 correction of MI355X_MICROARCH.md), the L2 hit rate and the walks' algorithmic bytes."""
import json, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = "profiles/r03_c5_sphere_field_1m_1024spp.txt"
txt = open(os.path.join(ROOT, src)).read()
val = lambda name: float(re.search(rf"^{name}\s+([0-9.e+]+)", txt, re.M).group(1))
kernel, avg_ns = re.search(r"(render_fastlanes_kernel<[^>]*>).*?avg_ns\s+([0-9.]+)", txt).groups()
fetch, write, hit, miss = val("FETCH_SIZE"), val("WRITE_SIZE"), val("TCC_HIT_sum"), val("TCC_MISS_sum")
wc = json.load(open(os.path.join(ROOT, "profiles", "work_counts.json")))["sphere_field_1m"]
paths = 1920 * 1080 * 1024
p = os.path.join(ROOT, "profiles", "c5_traffic.json")
d = json.load(open(p))
d["hbm_bytes_per_launch"] = int(2 * fetch * 1024 + write * 1024)
d["l2_hit"] = round(hit / (hit + miss), 3)
d["algorithmic_bytes_per_launch"] = int(wc["bytes_per_path"] * paths)
d["lane_steps_per_path"] = round(wc["own_tree_per_path"]["node_steps"] + wc["own_tree_per_path"]["leaf_tests"], 1)
d["paths_per_launch"] = paths
ms = float(avg_ns) * 1e-6
d["source"] = (f"{src} (rocprofv3 passes of `bench.py --config c5`): 2 x FETCH_SIZE {fetch:.5e} KiB (gfx950 correction; TCC_MISS_sum {miss:.4e} x 128 B "
               f"is the same number) + WRITE_SIZE {write:.0f} KiB, {kernel}, 1920x1080x1024spp, {ms:.0f} ms per launch: "
               f"{d['hbm_bytes_per_launch'] / ms / 1e9:.2f} TB/s at the memory side of the L2 (Infinity-Cache hits are counted in, and the 64 MB tree fits that "
               f"cache). The algorithmic bytes (nodes and leaves a walk touches x 32 B = {d['algorithmic_bytes_per_launch'] / 1e12:.1f} TB per launch) are mostly served by "
               f"the L1s and the L2 (hit rate {100 * d['l2_hit']:.0f} percent): they stand beside the roofline as roofline.algorithmic. Round 3's shade records and one-round "
               "surface fetch took 17 percent off the memory-side bytes (2.63 -> 2.18 TB per launch), so this fraction FELL while the kernel got faster.")
json.dump(d, open(p, "w"), indent=1)
print(d["hbm_bytes_per_launch"], d["l2_hit"], d["lane_steps_per_path"], f"{ms:.1f} ms")
