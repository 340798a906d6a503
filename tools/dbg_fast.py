import sys, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from ray_tracing_amd import _capi, api
from ray_tracing_amd.api import HostScene, Scene
g = np.load('tests/golden/cornell_box.npz')
tab = g['hits_128x128'].reshape(-1,20)
hs = HostScene('cornell_box',1.0); sc = Scene(hs,0)
print('records', sc.info(3))
sc.set_path(trace=_capi.RT_PATH_FAST)
hits = api.trace_rays(sc, tab[:,2:9].copy())
want = tab[:,10].astype(np.int32)
bad = hits['prim_id'] != want
print('mismatch', bad.sum(), 'of', len(want))
import collections
print(collections.Counter(zip(want[bad].tolist(), hits['prim_id'][bad].tolist())).most_common(12))
i = np.nonzero(bad)[0][:3]
for k in i: print(k, tab[k,2:9], 'want', want[k], tab[k,11], 'got', hits[k])
