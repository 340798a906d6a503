import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import oracle_lib as O
from ray_tracing_amd import _capi, api
from ray_tracing_amd.api import HostScene, Scene
gold = np.load(os.path.join(O.GOLD, "cornell_box.npz"))
for key, nx, ny in (("hits_128x128", 128, 128), ("hits_200x120", 200, 120)):
    tab = gold[key].reshape(-1, 21)
    sc = Scene(HostScene("cornell_box", nx / ny), 0)
    sc.set_path(trace=_capi.RT_PATH_FAST)
    rays = tab[:, 2:9].copy()
    h = api.trace_rays(sc, rays)
    ref_id = tab[:, 10].astype(np.int32)
    bad = np.flatnonzero(h["prim_id"] != ref_id)
    m = (tab[:, 9] > 0.5) & (h["prim_id"] == ref_id)
    tb = (h["t"][m].view(np.uint32) != tab[m, 11].astype(np.float32).view(np.uint32))
    print(key, "id mismatches", len(bad), "t-bit mismatches among same ids", tb.sum(), "of", m.sum())
    for i in bad[:10]:
        print("  ray", i, "st", tab[i, 0:2], "ref id", ref_id[i], "t", tab[i, 11], "gpu id", h["prim_id"][i], "t", h["t"][i])
    # which prims have t-bit mismatches
    ids = ref_id[m][tb]
    print("  t-mismatch by prim id:", dict(zip(*np.unique(ids, return_counts=True))))
    sc.close()
