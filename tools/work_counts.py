#!/usr/bin/env python3
"""Algorithmic work per path, written to profiles/work_counts.json (bench.py's rooflines read it).

  CPU part (no GPU needed): the instrumented CPU oracle on the reference-shaped tree -- rays, slab / rect / sphere tests
  and scatters per path (what SURVEY.md 3.3 measured on the reference) and the canonical FLOP count per path
  (SURVEY.md 8d costs: slab 15; rect reject 6, +12 on accept; sphere 23; diffuse shade 100; specular 40; ray-gen 20).
  GPU part (--gpu, needs a library built with -DRT_COUNT_WALK: tools/build_variant.sh count -DRT_COUNT_WALK, then
  RT_AMD_LIB=ray_tracing_amd/build_var/librt_count.so): node steps and leaf tests per path of the per-lane walk over the
  library's own tree; bytes_per_path = 32 B per node step + 32 B per leaf test.

usage: work_counts.py [--gpu] scene:nx:ny:spp ...
"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from ray_tracing_amd import _capi, api  # noqa: E402

OUT = os.path.join(ROOT, "profiles", "work_counts.json")


def cpu_counts(name, nx, ny, spp):
    import oracle_lib as O
    hs = api.HostScene(name, nx / ny)
    O.counters(reset=True)
    O.render(hs, nx, ny, 0, spp, 1, 50, O.ORC_RNG_COUNTER, 0, 8)
    c = O.counters()
    p = c["paths"]
    per = {k: v / p for k, v in c.items()}
    hits = per["rays"] - per["end_miss"]
    flop = per["slab_tests"] * 15 + per["rect_tests"] * 6 + per["sphere_tests"] * 23 + hits * 12 + per["scatters"] * 85 + 20
    return {"reference_tree_per_path": {k: round(v, 3) for k, v in per.items() if k != "paths"}, "flop_per_path": round(flop, 1),
            "reference_tree_bytes_per_path": round(per["slab_tests"] * 32 + per["sphere_tests"] * 16 + per["rect_tests"] * 20, 1),
            "cpu_sample": f"{nx}x{ny}x{spp}"}


def gpu_counts(name, nx, ny, spp):
    import torch
    lib = _capi.load()
    if not hasattr(lib, "rt_debug_walk_counts"):
        raise SystemExit("this library was not built with -DRT_COUNT_WALK (see the docstring)")
    lib.rt_debug_walk_counts.argtypes = [C.c_void_p, C.c_int]
    sc = api.Scene(api.HostScene(name, nx / ny), 0)
    if sc.info(_capi.RT_INFO_RECORDS) > 0:
        sc.close()
        return {}
    acc = torch.empty((ny, nx, 4), dtype=torch.float32, device="cuda:0")
    lib.rt_debug_walk_counts(None, 1)
    api.render(sc, acc, nx, ny, 0, spp)
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 16)()
    lib.rt_debug_walk_counts(out, 1)
    nodes, leaves, rays, paths, w_nodes, w_leaves, w_shades, shaded = [int(x) for x in out][:8]
    print(f"{name}: lanes on per wave-level node step {nodes / max(w_nodes, 1):.1f}, per leaf phase {leaves / max(w_leaves, 1):.1f}, "
          f"per shade phase {shaded / max(w_shades, 1):.1f}; per path: {w_nodes * 64 / paths:.1f} node-step lane slots for {nodes / paths:.1f} steps, "
          f"{w_leaves * 64 / paths:.1f} leaf slots for {leaves / paths:.1f} tests, {w_shades * 64 / paths:.1f} shade slots for {shaded / paths:.1f} shadings")
    levels = sc.info(_capi.RT_INFO_BVH_LEVELS)
    sc.close()
    return {"own_tree_per_path": {"node_steps": round(nodes / paths, 3), "leaf_tests": round(leaves / paths, 3), "rays": round(rays / paths, 3)},
            "bytes_per_path": round((nodes * 32 + leaves * 32) / paths, 1), "bvh_levels": levels, "gpu_sample": f"{nx}x{ny}x{spp}"}


def main():
    args = sys.argv[1:]
    gpu = "--gpu" in args
    jobs = [a for a in args if not a.startswith("--")] or ["cornell_box:256:256:16"]
    data = json.load(open(OUT)) if os.path.exists(OUT) else {}
    for job in jobs:
        name, nx, ny, spp = job.split(":")
        nx, ny, spp = int(nx), int(ny), int(spp)
        entry = data.get(name, {})
        entry.update(gpu_counts(name, nx, ny, spp) if gpu else cpu_counts(name, nx, ny, spp))
        data[name] = entry
        print(name, json.dumps(entry))
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main()
