#!/usr/bin/env python3
"""Prints where rt_debug_eval's one-bounce kinds differ from the oracle (development aid for tests/test_gpu_kat.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import oracle_lib as O
from ray_tracing_amd import _capi, api
from ray_tracing_amd.api import HostScene, Scene
from test_gpu_kat import _bounce_items
np.set_printoptions(linewidth=200, precision=7)
for scene, lo, hi in (("random_spheres", -6.0, 6.0), ("cornell_box_volumes", 1.0, 554.0), ("simple_scene", -3.0, 3.0)):
    hs = HostScene(scene, 1.0); sc = Scene(hs, 0)
    rng = np.random.default_rng(17); n = 3000
    it, keys, pixels = _bounce_items(rng, n, lo, hi)
    want = np.stack([O.kat_bounce(hs, it[i, 0:7], keys[i], pixels[i]) for i in range(n)])
    exact = api.debug_eval(sc, _capi.RT_EVAL_BOUNCE, it)[:, :10]
    fast = api.debug_eval(sc, _capi.RT_EVAL_BOUNCE_FAST, it)[:, :10]
    w = want[:, 6:9]
    dead = (want[:, 2] > 0) & (~np.isfinite(w).all(axis=1) | (w == 0).all(axis=1))
    want[dead, 2:9] = 0
    print(scene, "kinds", np.unique(want[:, 1], return_counts=True))
    bad = np.where(~np.all(exact[:, 0:3] == want[:, 0:3], axis=1))[0]
    print(" exact head mismatches", len(bad))
    for i in bad[:5]: print("  ", i, "exact", exact[i], "\n      want ", want[i])
    bad = np.where(~np.isclose(exact[:, 3:9], want[:, 3:9], rtol=1e-5, atol=1e-6).all(axis=1))[0]
    print(" exact value mismatches", len(bad))
    for i in bad[:5]: print("  ", i, "exact", exact[i], "\n      want ", want[i])
    same = np.all(fast[:, 0:3] == want[:, 0:3], axis=1)
    print(" fast same head", same.mean())
    bad = np.where(~same)[0]
    for i in bad[:5]: print("  ", i, "fast ", fast[i], "\n      want ", want[i])
    media = want[:, 1] == 4
    ok = same & ~media
    bad = np.where(ok & ~(np.isclose(fast[:, 3:6], want[:, 3:6], rtol=1e-3, atol=1e-4).all(axis=1) & np.isclose(fast[:, 6:9], want[:, 6:9], rtol=2e-3, atol=1e-5).all(axis=1)))[0]
    print(" fast value mismatches", len(bad))
    for i in bad[:8]: print("  ", i, "fast ", fast[i], "\n      want ", want[i])
    sc.close()
