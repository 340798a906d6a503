#!/usr/bin/env python3
"""Development: exact vs fast closest hits on rays inside the sphere grid of another_simple (media statistics)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ray_tracing_amd import api, _capi
scene = sys.argv[1] if len(sys.argv) > 1 else "another_simple"
hs = api.HostScene(scene, 1.0); sc = api.Scene(hs, 0)
rng = np.random.default_rng(3)
n = 400000
o = np.stack([rng.uniform(-5, 5, n), rng.uniform(16, 18, n), rng.uniform(-5, 5, n)], 1).astype(np.float32)
d = rng.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.concatenate([o, d, np.zeros((n, 1), np.float32)], 1)
d_ = hs.desc.contents
iso = np.array([d_.materials[i].kind == _capi.RT_MAT_ISOTROPIC for i in range(d_.n_materials)])
res = {}
for name, path in (("exact", _capi.RT_PATH_EXACT), ("fast", _capi.RT_PATH_FAST)):
    sc.set_path(trace=path)
    h = api.trace_rays(sc, rays, seed=11)
    med = (h["material"] >= 0) & iso[np.maximum(h["material"], 0)]
    res[name] = (h, med)
    print(name, "hit", (h["prim_id"] >= 0).mean(), "medium", med.mean(), "mean t medium", h["t"][med].mean(), "mean t solid", h["t"][(h["prim_id"] >= 0) & ~med].mean())
e, f = res["exact"], res["fast"]
both_solid = ~e[1] & ~f[1]
print("solid ids equal", (e[0]["prim_id"][both_solid] == f[0]["prim_id"][both_solid]).mean())
dm = e[1] != f[1]
print("medium/solid disagreement", dm.mean())
i = np.flatnonzero(dm)[:8]
for k in i:
    print(k, rays[k], "exact", e[0]["prim_id"][k], e[0]["t"][k], e[0]["material"][k], "fast", f[0]["prim_id"][k], f[0]["t"][k], f[0]["material"][k])
# analytic check: rays that start inside sphere (i, j): P(medium hit) = 1 - exp(-density * (L - 0.001)), L = distance to the exit
ci = np.clip(np.floor((o[:, 0] + 5) / 2), 0, 4); cj = np.clip(np.floor((o[:, 2] + 5) / 2), 0, 4)
c = np.stack([ci * 2 - 4, np.full(n, 17.0), cj * 2 - 4], 1)
oc = o - c
inside = (oc ** 2).sum(1) < 1.0
b = (oc * d).sum(1); cc = (oc ** 2).sum(1) - 1.0
L = -b + np.sqrt(np.maximum(b * b - cc, 0))
dens = 5 - cj
p = 1 - np.exp(-dens * np.maximum(L - 0.001, 0))
for j in range(5):
    m = inside & (cj == j)
    print(f"density {5 - j}: analytic {p[m].mean():.4f}  exact {e[1][m].mean():.4f}  fast {f[1][m].mean():.4f}   (n={m.sum()})")
m = ~inside
print(f"outside: exact {e[1][m].mean():.4f} fast {f[1][m].mean():.4f}")
