"""rt_trace_rays_noalloc (include/rt_amd.h): the closest-hit query with its device staging in a static device array
instead of hipMalloc'ed buffers.  Same kernels as rt_trace_rays, so the bar is equality bit for bit.  The file name
sorts last on purpose: this test must pass after every other test has used the GPU, and alone in a fresh process."""
import numpy as np
import pytest

from ray_tracing_amd import _capi, api
from ray_tracing_amd.api import HostScene, Scene

pytestmark = pytest.mark.gpu


def _rays(n, seed, lo, hi):
    rng = np.random.default_rng(seed)
    o = rng.uniform(lo, hi, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    return np.concatenate([o, d, rng.uniform(size=(n, 1)).astype(np.float32)], axis=1)


def _same(a, b):
    return a.tobytes() == b.tobytes()


@pytest.mark.parametrize("scene,lo,hi", [("cornell_box", 1.0, 554.0), ("cornell_box_volumes", 1.0, 554.0), ("random_spheres", -8.0, 8.0)])
def test_noalloc_trace_equals_rt_trace_rays_on_two_calls_in_a_row(scene, lo, hi):
    hs = HostScene(scene, 1.0)
    sc = Scene(hs, 0)
    for path in (_capi.RT_PATH_EXACT, _capi.RT_PATH_FAST):
        sc.set_path(trace=path)
        big, small = _rays(50000 + 37, 3, lo, hi), _rays(1000 + 5, 4, lo, hi)
        want_big, want_small = api.trace_rays(sc, big, seed=5), api.trace_rays(sc, small, seed=6)
        # two calls in a row on the same static array, the second with a smaller input: what the first call left
        # behind the second one's rays and hits must not show
        got_big = api.trace_rays(sc, big, seed=5, noalloc=True)
        got_small = api.trace_rays(sc, small, seed=6, noalloc=True)
        assert (want_big["prim_id"] >= 0).mean() > 0.3
        assert _same(got_big, want_big), f"{scene} path {path}: {np.sum(got_big['prim_id'] != want_big['prim_id'])} ids differ"
        assert _same(got_small, want_small), f"{scene} path {path}: second (smaller) call differs"
        # and a second scene on the same device shares the array
        other = Scene(hs, 0)
        other.set_path(trace=path)
        assert _same(api.trace_rays(other, small, seed=6, noalloc=True), want_small)
        other.close()
    sc.close()


def test_noalloc_trace_limits_and_edge_cases():
    sc = Scene(HostScene("cornell_box", 1.0), 0)
    assert len(api.trace_rays(sc, np.zeros((0, 7), np.float32), noalloc=True)) == 0
    full = _rays(_capi.RT_TRACE_NOALLOC_MAX_RAYS, 8, 1.0, 554.0)
    assert _same(api.trace_rays(sc, full, noalloc=True), api.trace_rays(sc, full))
    with pytest.raises(api.RtError, match="static staging area holds 65536"):
        api.trace_rays(sc, _rays(_capi.RT_TRACE_NOALLOC_MAX_RAYS + 1, 9, 1.0, 554.0), noalloc=True)
    one = np.array([[278, 278, -800, 0, 0, 1, 0]], dtype=np.float32)
    assert _same(api.trace_rays(sc, one, noalloc=True), api.trace_rays(sc, one))
    sc.close()
