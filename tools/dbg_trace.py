#!/usr/bin/env python3
"""Prints rays on which the FAST and EXACT trace paths disagree (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ray_tracing_amd import api, _capi
from ray_tracing_amd.api import HostScene, Scene
name = sys.argv[1] if len(sys.argv) > 1 else "cornell_box"
rng = np.random.default_rng(9)
n = 100000
o = rng.uniform(1, 554, size=(n, 3)).astype(np.float32)
d = rng.normal(size=(n, 3)).astype(np.float32)
rays = np.concatenate([o, d, np.zeros((n, 1), np.float32)], axis=1)
sc = Scene(HostScene(name, 1.0), 0)
exact = api.trace_rays(sc, rays)
sc.set_path(trace=_capi.RT_PATH_FAST)
fast = api.trace_rays(sc, rays)
fast2 = api.trace_rays(sc, rays)
print("fast twice equal:", np.array_equal(fast["prim_id"], fast2["prim_id"]), np.array_equal(fast["t"], fast2["t"]))
bad = np.nonzero(exact["prim_id"] != fast["prim_id"])[0]
print(len(bad), "differ; exact ids of those:", np.bincount(exact["prim_id"][bad] + 1), "fast ids:", np.bincount(fast["prim_id"][bad] + 1))
for i in bad[:12]:
    print("ray", rays[i, :6], "\n  exact id", exact["prim_id"][i], "t", exact["t"][i], "n", exact["normal"][i], "p", exact["p"][i],
          "\n  fast  id", fast["prim_id"][i], "t", fast["t"][i], "n", fast["normal"][i], "p", fast["p"][i])
