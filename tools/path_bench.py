#!/usr/bin/env python3
"""Exact against fast render path, kernel time from HIP events.  usage: path_bench.py [scene:nx:ny:spp ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ray_tracing_amd import api, _capi
jobs = sys.argv[1:] or ["cornell_box_volumes:512:512:64", "simple_scene:512:512:64", "another_simple:512:512:64"]
for job in jobs:
    name, nx, ny, spp = job.split(":"); nx, ny, spp = int(nx), int(ny), int(spp)
    for path in (_capi.RT_PATH_EXACT, _capi.RT_PATH_FAST):
        sc = api.Scene(api.HostScene(name, nx / ny), 0)
        try:
            sc.set_path(render=path)
            acc = torch.empty((ny, nx, 4), dtype=torch.float32, device="cuda:0")
            for _ in range(2):
                api.render(sc, acc, nx, ny, 0, spp)
            torch.cuda.synchronize()
            ms = sc.last_render_ms()
            print(f"{name:22s} path {path}: {ms:9.3f} ms {nx*ny*spp/ms/1e3:9.1f} Mpaths/s ran {sc.info(_capi.RT_INFO_LAST_RENDER_PATH)} mean {acc[...,1:].mean().item()/spp:.4f}", flush=True)
        except Exception as e:
            print(name, path, "error", e)
        sc.close()
