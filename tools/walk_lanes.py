#!/usr/bin/env python3
"""Lanes on per wave-level node step / leaf phase / shade phase of the per-lane kernels (needs a library built with
-DRT_COUNT_WALK: tools/build_variant.sh count -DRT_COUNT_WALK, then RT_AMD_LIB=ray_tracing_amd/build_var/librt_count.so).
usage: walk_lanes.py [--opt NAME=VALUE ...] scene:nx:ny:spp ...    (prints only; tools/work_counts.py --gpu writes profiles/work_counts.json)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ray_tracing_amd import _capi, api
lib = _capi.load()
lib.rt_debug_walk_counts.argtypes = [C.c_void_p, C.c_int]
opts, jobs = [], []
it = iter(sys.argv[1:])
for a in it:
    if a == "--opt":
        k, v = next(it).split("=")
        opts.append((getattr(_capi, "RT_OPT_" + k), int(v)))
    else:
        jobs.append(a)
for job in jobs:
    name, nx, ny, spp = job.split(":"); nx, ny, spp = int(nx), int(ny), int(spp)
    sc = api.Scene(api.HostScene(name, nx / ny), 0)
    for k, v in opts:
        sc.set_option(k, v)
    acc = torch.empty((ny, nx, 4), dtype=torch.float32, device="cuda:0")
    lib.rt_debug_walk_counts(None, 1)
    api.render(sc, acc, nx, ny, 0, spp)
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 16)()
    lib.rt_debug_walk_counts(out, 1)
    nodes, leaves, rays, paths, w_nodes, w_leaves, w_shades, shaded = [int(x) for x in out][:8]
    one, two, three = [int(x) for x in out][8:11]
    print(f"{name} {opts}: lanes on per wave-level node step {nodes / max(w_nodes, 1):.1f}, per leaf phase {leaves / max(w_leaves, 1):.1f}, "
          f"per shade phase {shaded / max(w_shades, 1):.1f}; per path: {w_nodes * 64 / paths:.1f} node-step lane slots for {nodes / paths:.1f} steps, "
          f"{w_leaves * 64 / paths:.1f} leaf slots for {leaves / paths:.1f} tests, {w_shades * 64 / paths:.1f} shade slots for {shaded / paths:.1f} shadings, rays {rays / paths:.2f}; leaf phases with one / two / three classes of leaves "
          f"(spheres and rects | boxes | the rest): {100 * one / max(w_leaves, 1):.0f} / {100 * two / max(w_leaves, 1):.0f} / {100 * three / max(w_leaves, 1):.0f} %", flush=True)
    sc.close()
