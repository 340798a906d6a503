#!/usr/bin/env python3
"""Sweep of the per-lane walk's batch thresholds (RT_OPT_SHADE_BATCH, RT_OPT_LEAF_BATCH) on one scene; the scene is built once."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ray_tracing_amd import api, _capi
name, nx, ny, spp = sys.argv[1].split(":"); nx, ny, spp = int(nx), int(ny), int(spp)
sc = api.Scene(api.HostScene(name, nx / ny), 0)
acc = torch.empty((ny, nx, 4), dtype=torch.float32, device="cuda:0")
for shade in (8, 16, 24, 32, 40, 48, 56):
    row = []
    for leaf in (4, 8, 16, 24, 32):
        sc.set_option(_capi.RT_OPT_SHADE_BATCH, shade); sc.set_option(_capi.RT_OPT_LEAF_BATCH, leaf)
        for _ in range(2):
            api.render(sc, acc, nx, ny, 0, spp)
        torch.cuda.synchronize()
        row.append(f"{nx*ny*spp/sc.last_render_ms()/1e3:8.1f}")
    print(f"{name} shade {shade:2d}: leaf 4/8/16/24/32 ->", " ".join(row), flush=True)
