#!/usr/bin/env python3
"""Where the per-lane fast walk and the oracle disagree on the 1M-sphere scene: how far apart the two hits are."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from ray_tracing_amd import _capi, api
name = sys.argv[1] if len(sys.argv) > 1 else "sphere_field_1m"
hs = api.HostScene(name, 1920 / 1080)
sc = api.Scene(hs, 0)
rng = np.random.default_rng(77)
n = 20000
cam = api.camera_rays(sc, rng.uniform(size=(n // 2, 2)).astype(np.float32))
half = 50.0 if name.endswith("1m") else 8.0
o = rng.uniform(-0.8 * half, 0.8 * half, size=(n // 2, 3)).astype(np.float32)
rays = np.concatenate([cam, np.concatenate([o, rng.normal(size=(n // 2, 3)).astype(np.float32), np.zeros((n // 2, 1), np.float32)], axis=1)])
sc.set_path(trace=_capi.RT_PATH_FAST)
fast = api.trace_rays(sc, rays)
cpu = O.trace_rays(hs, rays)
bad = np.flatnonzero(fast["prim_id"] != cpu["prim_id"])
print(len(bad), "of", n, "ids differ;", (bad < n // 2).sum(), "of them camera rays")
dt = fast["t"][bad] - cpu["t"][bad]
rel = dt / np.maximum(cpu["t"][bad], 1e-6)
print("fast miss where cpu hits:", ((fast["prim_id"][bad] < 0) & (cpu["prim_id"][bad] >= 0)).sum(),
      " cpu miss where fast hits:", ((fast["prim_id"][bad] >= 0) & (cpu["prim_id"][bad] < 0)).sum())
both = (fast["prim_id"][bad] >= 0) & (cpu["prim_id"][bad] >= 0)
print("both hit:", both.sum(), " rel dt percentiles (1,25,50,75,99):", np.percentile(rel[both], [1, 25, 50, 75, 99]))
print("fast farther by > 1e-3 rel:", (rel[both] > 1e-3).sum(), " fast nearer by > 1e-3 rel:", (rel[both] < -1e-3).sum())
for i in bad[:8]:
    print(" ray", i, "o", rays[i, :3], "d", rays[i, 3:6], "fast", fast["prim_id"][i], fast["t"][i], "cpu", cpu["prim_id"][i], cpu["t"][i])
same = np.flatnonzero((fast["prim_id"] == cpu["prim_id"]) & (cpu["prim_id"] >= 0))
r2 = np.abs(fast["t"][same] - cpu["t"][same]) / cpu["t"][same]
print("same ids: rel |dt| percentiles (50, 99, max):", np.percentile(r2, [50, 99]), r2.max())
