#!/usr/bin/env python3
"""Mpaths/s of rt_render on the built-in scenes (kernel time from HIP events).
usage: scene_bench.py [--opt NAME=VALUE ...] [scene:nx:ny:spp ...]     NAME: an RT_OPT_* of include/rt_amd.h without the prefix,
e.g. --opt SHADE_BATCH=24 --opt TARGET_TASKS=524288 (rt_scene_set_option on every scene)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ray_tracing_amd import api, _capi
opts, args = [], []
it = iter(sys.argv[1:])
for a in it:
    if a == "--opt":
        k, v = next(it).split("=")
        opts.append((getattr(_capi, "RT_OPT_" + k), int(v)))
    else:
        args.append(a)
jobs = args or ["cornell_box:512:512:16", "cornell_box_volumes:256:256:8", "random_spheres:300:200:4", "next_week_final:200:200:2"]
for job in jobs:
    name, nx, ny, spp = job.split(":"); nx, ny, spp = int(nx), int(ny), int(spp)
    sc = api.Scene(api.HostScene(name, nx / ny), 0)
    for k, v in opts:
        sc.set_option(k, v)
    acc = torch.empty((ny, nx, 4), dtype=torch.float32, device="cuda:0")
    for _ in range(2):
        api.render(sc, acc, nx, ny, 0, spp)
    torch.cuda.synchronize()
    ms = sc.last_render_ms()
    print(f"{name:22s} {nx}x{ny}x{spp}: {ms:10.3f} ms  {nx*ny*spp/ms/1e3:10.2f} Mpaths/s  path {sc.info(_capi.RT_INFO_LAST_RENDER_PATH)} mean {acc[...,1:].mean().item()/spp:.4f}", flush=True)
    sc.close()
