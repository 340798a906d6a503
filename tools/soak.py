#!/usr/bin/env python3
"""Renders the same image many times and compares each with the first (chained kernels: races would show as differences)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ray_tracing_amd import api
jobs = sys.argv[1:] or ["cornell_box:256:256:64:150", "random_spheres:192:128:32:100", "cornell_box_volumes:128:128:32:100", "next_week_final:128:128:16:60"]
for job in jobs:
    name, nx, ny, spp, reps = job.split(":"); nx, ny, spp, reps = int(nx), int(ny), int(spp), int(reps)
    sc = api.Scene(api.HostScene(name, nx / ny), 0)
    acc = torch.empty((ny, nx, 4), dtype=torch.float32, device="cuda:0")
    api.render(sc, acc, nx, ny, 0, spp)
    first = acc.clone()
    worst = 0.0
    for i in range(reps):
        api.render(sc, acc, nx, ny, 0, spp)
        d = ((acc - first).abs() / (first.abs() + 1e-3)).max().item()
        worst = max(worst, d)
        if not torch.all(acc[..., 0] == spp):
            print(name, "sample counts wrong at repetition", i); sys.exit(1)
    print(f"{name:22s} {reps} repetitions, worst relative difference {worst:.2e}", flush=True)
    if worst > 1e-4:
        sys.exit(1)
    sc.close()
