#!/usr/bin/env python3
"""Kernel time of rt_render vs max_depth (how the cost splits between ray generation, scan and shading)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ray_tracing_amd import api, _capi
nx = ny = 1024; spp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
sc = api.Scene(api.HostScene("cornell_box", 1.0), 0)
acc = torch.empty((ny, nx, 4), dtype=torch.float32, device="cuda:0")
for path in (_capi.RT_PATH_FAST, _capi.RT_PATH_EXACT):
    sc.set_path(render=path)
    for depth in (0, 1, 2, 4, 50):
        for _ in range(2):
            api.render(sc, acc, nx, ny, 0, spp, max_depth=depth)
        torch.cuda.synchronize()
        ms = sc.last_render_ms()
        print(f"path {path} depth {depth:2d}: {ms:8.3f} ms  {nx*ny*spp/ms/1e3:9.1f} Mpaths/s  mean {acc[...,1:].mean().item()/spp:.5f}", flush=True)
