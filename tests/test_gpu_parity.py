"""GPU parity tests: the HIP path (through the C-ABI) against the oracle and the reference's golden vectors."""
import os

import numpy as np
import pytest

import oracle_lib as O
from ray_tracing_amd import _capi, api
from ray_tracing_amd.api import HostScene, Scene

PATHS = {"exact": _capi.RT_PATH_EXACT, "fast": _capi.RT_PATH_FAST}

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def gpu_render(scene, nx, ny, s0, s1, **kw):
    import torch
    acc = torch.empty((ny, nx, 4), dtype=torch.float32, device="cuda:0")
    api.render(scene, acc, nx, ny, s0, s1, **kw)
    torch.cuda.synchronize()
    return acc.cpu().numpy()


@pytest.mark.parametrize("key,nx,ny", [("hits_128x128", 128, 128), ("hits_200x120", 200, 120)])
def test_primary_hits_bit_exact_vs_reference(golden, key, nx, ny):
    """north_star check 1: primary-ray closest-hit primitive ids (and t, p, normal) bit-exact
    with the reference's hitable::hit on the same camera."""
    tab = golden[key].reshape(-1, 21)
    hs = HostScene("cornell_box", nx / ny)
    sc = Scene(hs, 0)
    rays = api.camera_rays(sc, tab[:, 0:2])
    assert np.array_equal(bits(rays[:, :6]), bits(tab[:, 2:8]))
    rays[:, 6] = tab[:, 8]
    hits = api.trace_rays(sc, rays)
    assert np.array_equal(hits["prim_id"], tab[:, 10].astype(np.int32))
    m = tab[:, 9] > 0.5
    assert np.array_equal(bits(hits["t"][m]), bits(tab[m, 11]))
    assert np.array_equal(bits(hits["p"][m]), bits(tab[m, 14:17]))
    assert np.array_equal(bits(hits["normal"][m]), bits(tab[m, 17:20]))
    sc.close()


def test_secondary_rays_bit_exact_vs_oracle(cornell_host, cornell_gpu):
    """Incoherent rays from inside the box (what bounces look like), every lane on its own course."""
    rng = np.random.default_rng(5)
    n = 50000
    o = rng.uniform(1, 554, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d[::7, 0] = 0.0  # axis-parallel directions: 1/d = inf in the slab test
    d[::11, 1] = -0.0
    rays = np.concatenate([o, d, np.zeros((n, 1), np.float32)], axis=1)
    for (tmin, tmax) in [(0.001, O.FLT_MAX), (0.001, 300.0)]:
        g = api.trace_rays(cornell_gpu, rays, tmin, tmax)
        c = O.trace_rays(cornell_host, rays, tmin, tmax)
        assert np.array_equal(g["prim_id"], c["prim_id"])
        assert np.array_equal(g["material"], c["material"])
        m = c["prim_id"] >= 0
        assert np.array_equal(bits(g["t"][m]), bits(c["t"][m]))
        assert np.array_equal(bits(g["p"][m]), bits(c["p"][m]))
        assert np.array_equal(bits(g["normal"][m]), bits(c["normal"][m]))


@pytest.mark.parametrize("key,nx,ny", [("hits_128x128", 128, 128), ("hits_200x120", 200, 120)])
def test_fast_path_primary_ids_vs_reference(golden, key, nx, ny):
    """north_star check 1 on the path rt_render runs on this scene (RT_PATH_FAST: the register-resident record table;
    rt_trace_rays then runs the very scan the render kernels run).  Its hit decisions repeat the reference's arithmetic
    -- t = (k - o) / d correctly rounded, in-plane point as product then sum, the sphere's roots operation by
    operation -- so ids AND t must equal the reference's primary-hit table bit for bit.  (p and the normal are built
    after the scan with FMA: close, not equal.)"""
    tab = golden[key].reshape(-1, 21)
    hs = HostScene("cornell_box", nx / ny)
    sc = Scene(hs, 0)
    assert sc.info(_capi.RT_INFO_FAST_PATH_AVAILABLE) == 1 and 0 < sc.info(_capi.RT_INFO_RECORDS) <= 64
    sc.set_path(trace=_capi.RT_PATH_FAST)
    rays = tab[:, 2:9].copy()
    hits = api.trace_rays(sc, rays)
    assert np.array_equal(hits["prim_id"], tab[:, 10].astype(np.int32)), \
        f"{(hits['prim_id'] != tab[:, 10].astype(np.int32)).sum()} ids differ from the reference"
    m = tab[:, 9] > 0.5
    assert len(np.unique(hits["prim_id"][m])) >= 8  # walls, lamp, sphere and faces of the tall box are all in the table
    assert np.array_equal(bits(hits["t"][m]), bits(tab[m, 11]))
    assert np.allclose(hits["p"][m], tab[m, 14:17], rtol=1e-5, atol=2e-3)
    assert np.allclose(hits["normal"][m], tab[m, 17:20], atol=1e-5)
    sc.close()


def test_record_table_path_refuses_a_range_that_starts_at_or_below_zero(cornell_host):
    """The record-table scan orders hits by the bits of t (rtf::Best), which order like t only for t > 0: with t_min <= 0 it
    would drop hits at negative t that the exact path and the reference report.  Such a range is refused on that path, on both
    closest-hit entry points, and answered -- like the oracle, bit for bit -- on the exact one."""
    rng = np.random.default_rng(31)
    n = 4000
    o = rng.uniform(1, 554, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    rays = np.concatenate([o, d, np.zeros((n, 1), np.float32)], axis=1)
    sc = Scene(cornell_host, 0)
    for t_min in (0.0, -1000.0):
        exact = api.trace_rays(sc, rays, t_min=t_min)
        cpu = O.trace_rays(cornell_host, rays, t_min=t_min)
        assert np.array_equal(exact["prim_id"], cpu["prim_id"]) and np.array_equal(bits(exact["t"]), bits(cpu["t"]))
    assert (exact["t"] < 0).mean() > 0.2  # the range really reaches behind the origins
    sc.set_path(trace=_capi.RT_PATH_FAST)
    for t_min in (0.0, -1000.0, float("nan")):
        for noalloc in (False, True):
            with pytest.raises(api.RtError, match="needs t_min > 0"):
                api.trace_rays(sc, rays, t_min=t_min, noalloc=noalloc)
    ok = api.trace_rays(sc, rays, t_min=1e-6)  # any positive start is fine
    sc.set_path(trace=_capi.RT_PATH_EXACT)
    assert (ok["prim_id"] == api.trace_rays(sc, rays, t_min=1e-6)["prim_id"]).mean() > 0.999 and (ok["prim_id"] >= 0).mean() > 0.8
    sc.close()


def test_fast_path_agrees_with_exact_path_on_incoherent_rays(cornell_host):
    """Bounce-like rays: the fast path's ids and t against the bit-exact path (itself equal to the oracle,
    test_secondary_rays_bit_exact_vs_oracle) and against the oracle directly.  The rects' and the sphere's t are
    the reference's bit for bit; what may differ is a ray through an edge of the tall box (one slab test there
    instead of six rect tests: no crack between faces) and which of two primitives at exactly the same t wins."""
    rng = np.random.default_rng(9)
    n = 100000
    o = rng.uniform(1, 554, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d[::13, 0] = 0.0  # axis-parallel: 1/d is infinite
    d[::17, 2] = -0.0
    rays = np.concatenate([o, d, np.zeros((n, 1), np.float32)], axis=1)
    sc = Scene(cornell_host, 0)
    exact = api.trace_rays(sc, rays)
    sc.set_path(trace=_capi.RT_PATH_FAST)
    fast = api.trace_rays(sc, rays)
    sc.close()
    cpu = O.trace_rays(cornell_host, rays)
    assert np.array_equal(exact["prim_id"], cpu["prim_id"])
    same = exact["prim_id"] == fast["prim_id"]
    assert same.mean() >= 0.99999, f"{(~same).sum()} of {n} ids differ"
    m = same & (exact["prim_id"] >= 0)
    assert np.array_equal(bits(fast["t"][m]), bits(cpu["t"][m]))
    assert np.array_equal(fast["material"][m], exact["material"][m])
    assert np.allclose(fast["normal"][m], exact["normal"][m], atol=1e-3)


def test_fast_path_resolves_exact_ties_like_the_reference(cornell_host):
    """Rays in the planes x = y and x + y = 555 through the room's centre line meet two walls at EXACTLY the same t along
    the room's edges (the same subtraction and division on both axes).  bvh_node::hit gives such a tie to the primitive
    its depth-first walk meets later (src/bvh.h:132-139); the fast path's scan visits records in another order
    (by chain and axis) and settles ties by rank (rt_records.h), so the ids must still agree with the oracle's."""
    rng = np.random.default_rng(23)
    n = 40000
    c = np.float32(277.5)
    oz = rng.uniform(-800, 500, size=n).astype(np.float32)
    o = np.stack([np.full(n, c), np.full(n, c), oz], axis=1)
    a = rng.uniform(0.05, 1.0, size=n).astype(np.float32)
    sx = np.where(rng.integers(0, 2, size=n) == 0, -1.0, 1.0).astype(np.float32)
    sy = np.where(rng.integers(0, 2, size=n) == 0, -1.0, 1.0).astype(np.float32)
    d = np.stack([sx * a, sy * a, rng.uniform(-1, 1, size=n).astype(np.float32)], axis=1)
    rays = np.concatenate([o, d, np.zeros((n, 1), np.float32)], axis=1).astype(np.float32)
    sc = Scene(cornell_host, 0)
    sc.set_path(trace=_capi.RT_PATH_FAST)
    fast = api.trace_rays(sc, rays)
    sc.close()
    cpu = O.trace_rays(cornell_host, rays)
    assert np.array_equal(fast["prim_id"], cpu["prim_id"]), f"{(fast['prim_id'] != cpu['prim_id']).sum()} ids differ"
    m = cpu["prim_id"] >= 0
    assert np.array_equal(bits(fast["t"][m]), bits(cpu["t"][m]))
    # the construction does produce edge hits: both walls' planes are reached at the same t on a good part of the rays
    t_x = (np.where(sx < 0, np.float32(0.0), np.float32(555.0)) - c) / d[:, 0]
    on_edge = m & (cpu["t"] == t_x.astype(np.float32))
    assert on_edge.sum() > 1000


def test_trace_edge_cases(cornell_gpu, cornell_host):
    assert len(api.trace_rays(cornell_gpu, np.zeros((0, 7), np.float32))) == 0
    one = np.array([[278, 278, -800, 0, 0, 1, 0]], dtype=np.float32)
    g, c = api.trace_rays(cornell_gpu, one), O.trace_rays(cornell_host, one)
    assert g["prim_id"][0] == c["prim_id"][0] >= 0 and bits(g["t"])[0] == bits(c["t"])[0]
    # a ragged count (not a multiple of the wave or block size)
    rays = np.tile(one, (1000 + 37, 1))
    assert np.all(api.trace_rays(cornell_gpu, rays)["prim_id"] == c["prim_id"][0])
    away = np.array([[278, 278, -800, 0, 0, -1, 0]], dtype=np.float32)
    assert api.trace_rays(cornell_gpu, away)["prim_id"][0] == -1


@pytest.mark.parametrize("path", ["exact", "fast"])
@pytest.mark.parametrize("nx,ny,ns,nss,depth", [(64, 64, 8, 1, 50), (40, 24, 4, 2, 5), (33, 17, 3, 1, 1), (16, 16, 2, 1, 0)])
def test_render_matches_oracle_on_the_same_stream(nx, ny, ns, nss, depth, path):
    """Same scene, same per-path random stream (include/rt_rng.h), same draw order: the GPU image must
    equal the CPU oracle's pixel by pixel up to libm/rounding differences.  A path can diverge when an
    ulp flips a comparison, so a small fraction of pixels may differ by more; none may differ in the mean.
    Tolerance: >= 97 % of pixels within 1e-3 relative (+1e-4 absolute), image mean within 1 %."""
    hs = HostScene("cornell_box", nx / ny)
    sc = Scene(hs, 0)
    sc.set_path(render=PATHS[path])
    g = gpu_render(sc, nx, ny, 0, ns, sub_samples=nss, max_depth=depth, seed=3)
    assert sc.info(_capi.RT_INFO_LAST_RENDER_PATH) == PATHS[path]
    c, _, _ = O.render(hs, nx, ny, 0, ns, nss, depth, O.ORC_RNG_COUNTER, 3, 8)
    assert np.all(g[..., 0] == ns) and np.all(c[..., 0] == ns)
    close = np.isclose(g[..., 1:], c[..., 1:], rtol=1e-3, atol=1e-4).all(axis=-1)
    assert close.mean() >= 0.97, f"only {close.mean():.4f} of pixels agree"
    gm, cm = g[..., 1:].mean(axis=(0, 1)), c[..., 1:].mean(axis=(0, 1))
    assert np.allclose(gm, cm, rtol=1e-2, atol=1e-5), (gm, cm)
    sc.close()


def test_block_kernel_and_wave_kernel_trace_the_same_paths(cornell_gpu):
    """The record-table path has two kernels (one wave per 8x8 tile; one block per 16x16 tile that regroups its paths by
    class).  Paths are keyed by (pixel, sample) and both kernels run the same per-path code, so their images agree up to
    float summation order."""
    nx, ny, ns = 80, 48, 8
    block = gpu_render(cornell_gpu, nx, ny, 0, ns, sub_samples=2, max_depth=50, seed=6)
    cornell_gpu.set_option(_capi.RT_OPT_BLOCK_KERNEL, 0)
    wave = gpu_render(cornell_gpu, nx, ny, 0, ns, sub_samples=2, max_depth=50, seed=6)
    cornell_gpu.set_option(_capi.RT_OPT_BLOCK_KERNEL, 1)
    cornell_gpu.set_option(_capi.RT_OPT_BLOCK_WAVES, 8)  # the eight-wave block of large launches (32x16 tiles), forced at this size
    block8 = gpu_render(cornell_gpu, nx, ny, 0, ns, sub_samples=2, max_depth=50, seed=6)
    cornell_gpu.set_option(_capi.RT_OPT_BLOCK_WAVES, 0)
    assert np.all(block[..., 0] == ns) and np.all(wave[..., 0] == ns) and np.all(block8[..., 0] == ns)
    for other in (wave, block8):
        close = np.isclose(block[..., 1:], other[..., 1:], rtol=1e-4, atol=1e-5).all(axis=-1)
        assert close.mean() >= 0.995, f"only {close.mean():.4f} of pixels agree"
        assert np.allclose(block[..., 1:].mean(axis=(0, 1)), other[..., 1:].mean(axis=(0, 1)), rtol=1e-3)


@pytest.mark.parametrize("scene,nx,ny,ns", [("cornell_box", 100, 60, 40), ("cornell_box", 16, 16, 3), ("random_spheres", 72, 40, 24),
                                             ("cornell_box_volumes", 40, 40, 24)])
def test_chained_tasks_trace_the_same_paths(scene, nx, ny, ns):
    """The fast kernels take task after task from a counter (a wave or block whose pool is empty flushes its tile and goes
    on; the stragglers of the previous task add to the image themselves).  Paths are keyed by (pixel, sample), so the image
    equals that of the one-task-per-wave form up to float summation order -- whatever the interleaving."""
    hs = HostScene(scene, nx / ny)
    sc = Scene(hs, 0)
    sc.set_path(render=_capi.RT_PATH_FAST)
    chained = gpu_render(sc, nx, ny, 0, ns, sub_samples=1, max_depth=50, seed=11)
    again = gpu_render(sc, nx, ny, 0, ns, sub_samples=1, max_depth=50, seed=11)
    sc.set_option(_capi.RT_OPT_CHAIN_TASKS, 0)
    single = gpu_render(sc, nx, ny, 0, ns, sub_samples=1, max_depth=50, seed=11)
    sc.close()
    assert np.all(chained[..., 0] == ns) and np.all(single[..., 0] == ns)
    assert np.allclose(chained, single, rtol=2e-5, atol=1e-5)
    assert np.allclose(chained, again, rtol=2e-5, atol=1e-5)


@pytest.mark.parametrize("scene,nx,ny,ns", [("random_spheres", 72, 40, 24), ("cornell_box_volumes", 40, 40, 24), ("next_week_final", 48, 48, 8),
                                             ("sphere_field_4k", 40, 24, 8)])
def test_regrouping_per_lane_kernel_traces_the_same_paths(scene, nx, ny, ns):
    """RT_OPT_LANES_REGROUP = 1: render_fastlanes_block_kernel, where the four waves of a block trade paths (and the cursors of the
    walks under way) through LDS every few node steps.  A path's draws and the order of its leaf visits do not depend on the lane
    that carries it, so the image equals the wave kernel's up to float summation order -- media included -- at any cadence,
    chained or not."""
    sc = Scene(HostScene(scene, nx / ny), 0)
    wave = gpu_render(sc, nx, ny, 0, ns, seed=13)
    assert sc.info(_capi.RT_INFO_RECORDS) == -1 and sc.info(_capi.RT_INFO_LAST_RENDER_PATH) == _capi.RT_PATH_FAST
    sc.set_option(_capi.RT_OPT_LANES_REGROUP, 1)
    for rounds, chain in ((0, 1), (1, 1), (3, 0)):
        sc.set_option(_capi.RT_OPT_REGROUP_ROUNDS, rounds)
        sc.set_option(_capi.RT_OPT_CHAIN_TASKS, chain)
        block = gpu_render(sc, nx, ny, 0, ns, seed=13)
        assert np.all(block[..., 0] == ns)
        assert np.allclose(block, wave, rtol=2e-5, atol=1e-5), (rounds, chain, np.abs(block - wave).max())
    sc.close()


def test_generic_scan_with_box_records_matches_oracle():
    """Scenes of more than 16 primitives keep their box records in the record table and are scanned by the generic loop
    (lanes that miss a box skip its subtree).  No built-in scene of that size fits the table, so the Cornell box is built
    with its boxes kept (RT_CREATE_KEEP_BOX_RECORDS) and must give the oracle's image on the same random stream, and the
    oracle's closest hits (ids and t bit for bit: the generic scan decides hits with the same arithmetic as the run scan)."""
    nx, ny, ns = 48, 40, 6
    hs = HostScene("cornell_box", nx / ny)
    sc = Scene(hs, 0, create_flags=_capi.RT_CREATE_KEEP_BOX_RECORDS)
    plain = Scene(hs, 0)
    assert sc.info(_capi.RT_INFO_RECORDS) > plain.info(_capi.RT_INFO_RECORDS)  # the 7 boxes of the tree are in the table
    sc.set_path(render=_capi.RT_PATH_FAST)
    g = gpu_render(sc, nx, ny, 0, ns, max_depth=50, seed=3)
    c, _, _ = O.render(hs, nx, ny, 0, ns, 1, 50, O.ORC_RNG_COUNTER, 3, 8)
    close = np.isclose(g[..., 1:], c[..., 1:], rtol=1e-3, atol=1e-4).all(axis=-1)
    assert close.mean() >= 0.97, f"only {close.mean():.4f} of pixels agree"
    assert np.allclose(g[..., 1:].mean(axis=(0, 1)), c[..., 1:].mean(axis=(0, 1)), rtol=1e-2, atol=1e-5)
    rng = np.random.default_rng(31)
    n = 30000
    rays = np.concatenate([rng.uniform(1, 554, size=(n, 3)), rng.normal(size=(n, 3)), np.zeros((n, 1))], axis=1).astype(np.float32)
    sc.set_path(trace=_capi.RT_PATH_FAST)
    fast, cpu = api.trace_rays(sc, rays), O.trace_rays(hs, rays)
    same = fast["prim_id"] == cpu["prim_id"]
    assert same.mean() >= 0.9999, f"{(~same).sum()} ids differ"
    m = same & (cpu["prim_id"] >= 0)
    assert np.array_equal(bits(fast["t"][m]), bits(cpu["t"][m]))
    sc.close(), plain.close()


def test_render_is_deterministic_and_sample_ranges_add_up(cornell_gpu):
    """Disjoint sample ranges (what different GPUs render) sum to the full render: the paths are
    keyed by (pixel, sample), not by who traces them.  Float sums differ only in order."""
    import torch
    nx = ny = 48
    full = gpu_render(cornell_gpu, nx, ny, 0, 16, seed=11)
    again = gpu_render(cornell_gpu, nx, ny, 0, 16, seed=11)
    assert np.allclose(full, again, rtol=1e-5, atol=1e-6)
    acc = torch.zeros((ny, nx, 4), dtype=torch.float32, device="cuda:0")
    for (a, b) in [(0, 5), (5, 6), (6, 16)]:
        api.render(cornell_gpu, acc, nx, ny, a, b, seed=11, accumulate=True)
    torch.cuda.synchronize()
    parts = acc.cpu().numpy()
    assert np.all(parts[..., 0] == 16)
    assert np.allclose(full, parts, rtol=1e-4, atol=1e-5)
    other = gpu_render(cornell_gpu, nx, ny, 0, 16, seed=12)
    assert not np.allclose(full, other, rtol=1e-3)
    empty = gpu_render(cornell_gpu, nx, ny, 4, 4, seed=11)
    assert not empty.any()


def test_resolve_matches_oracle(cornell_gpu):
    import torch
    nx, ny = 40, 24
    acc = torch.empty((ny, nx, 4), dtype=torch.float32, device="cuda:0")
    api.render(cornell_gpu, acc, nx, ny, 0, 8, seed=1)
    img = api.resolve(acc)
    torch.cuda.synchronize()
    want = O.resolve(acc.cpu().numpy())
    assert np.array_equal(img.cpu().numpy().view(np.uint32), want)
    # special values: a NaN/inf/negative accumulator must quantise like the CPU loop does
    a = torch.tensor([[[4.0, 0.0, 2.0, 8.0], [1.0, float("inf"), -1.0, 0.25]]], device="cuda:0")
    assert np.array_equal(api.resolve(a).cpu().numpy().view(np.uint32), O.resolve(a.cpu().numpy()))


def test_statistical_parity_with_reference_image():
    """north_star check 2 at test scale: per-channel image mean within 3 standard errors of the
    reference's converged render, PSNR of the 8-bit images stated.  The reference image is the
    4096-spp golden fixture when present (256x256, -t 1)."""
    path = os.path.join(O.GOLD, "cornell_box_256x256_4096spp_r50.npy")
    if not os.path.exists(path):
        pytest.skip("converged reference fixture not generated yet")
    ref = np.load(path)  # (ny, nx, 3) RGB means
    ny, nx = ref.shape[:2]
    hs = HostScene("cornell_box", nx / ny)
    sc = Scene(hs, 0)
    batches = [gpu_render(sc, nx, ny, k * 512, (k + 1) * 512, seed=21)[..., 1:][..., ::-1] / 512.0 for k in range(8)]
    sc.close()
    gpu = np.mean(batches, axis=0)  # 4096 spp, RGB
    gmean = np.array([b.mean(axis=(0, 1)) for b in batches])  # (8, 3) image means per batch
    se_gpu = gmean.std(axis=0, ddof=1) / np.sqrt(len(batches))
    # the reference render has the same per-sample variance at equal spp
    se = np.sqrt(2.0) * se_gpu
    diff = np.abs(gpu.mean(axis=(0, 1)) - ref.mean(axis=(0, 1)))
    assert np.all(diff <= 3.0 * se + 1e-6), (diff, se)

    def to8(x):
        return np.floor(np.clip(np.sqrt(np.maximum(x, 0)), 0, 1) * 255.99)
    mse = np.mean((to8(gpu) - to8(ref)) ** 2)
    psnr = 10 * np.log10(255.0 ** 2 / mse)
    print(f"PSNR gpu(4096 spp) vs reference(4096 spp) = {psnr:.2f} dB; mean diff {diff}, 3*SE {3 * se}")
    assert psnr >= 30.0


def _to8(x):
    return np.floor(np.clip(np.sqrt(np.maximum(x, 0)), 0, 1) * 255.99)


def _psnr(a, b):
    return 10 * np.log10(255.0 ** 2 / np.mean((_to8(a) - _to8(b)) ** 2))


def _gpu_rgb(sc, n, spp, seed):
    return gpu_render(sc, n, n, 0, spp, seed=seed)[..., 1:][..., ::-1] / spp


def test_cornell_4k_rank_share_of_baseline_config_4():
    """BASELINE config 4 at its full image size: 3840 x 2160, the last four samples of the 16 384 (what the tail of rank 7's
    range looks like).  Every pixel gets its four samples, nothing is NaN or negative, the image mean is the small render's
    (aspect ratio and camera equal), and the same range rendered in two halves adds up to it."""
    nx, ny, s0, s1 = 3840, 2160, 16380, 16384
    sc = Scene(HostScene("cornell_box", nx / ny), 0)
    full = gpu_render(sc, nx, ny, s0, s1, seed=0)
    assert sc.info(_capi.RT_INFO_LAST_RENDER_PATH) == _capi.RT_PATH_FAST
    assert np.all(full[..., 0] == s1 - s0) and np.isfinite(full).all() and (full[..., 1:] >= 0).all()
    halves = gpu_render(sc, nx, ny, s0, s0 + 2, seed=0) + gpu_render(sc, nx, ny, s0 + 2, s1, seed=0)
    assert np.allclose(full, halves, rtol=1e-4, atol=1e-4)
    small = gpu_render(sc, 384, 216, 0, 256, seed=3)
    sc.close()
    m_full, m_small = full[..., 1:].mean(axis=(0, 1)) / (s1 - s0), small[..., 1:].mean(axis=(0, 1)) / 256
    assert np.allclose(m_full, m_small, rtol=0.03), (m_full, m_small)


def test_identical_rays_at_different_indices_draw_different_numbers():
    """include/rt_rng.h on the device: a path's stream is keyed by (key, pixel), and for rt_trace_rays the pixel word is the
    ray's index.  4096 copies of ONE ray through the smoke boxes of cornell_box_volumes (constant_medium::hit draws its
    free-flight distance, src/volume.h:43) must scatter at 4096 different distances, the same ones as the oracle on the same
    stream, on both GPU paths; another seed gives other distances."""
    hs = HostScene("cornell_box_volumes", 1.0)
    sc = Scene(hs, 0)
    one = np.array([[278, 278, -800, -0.05, -0.12, 1, 0]], dtype=np.float32)
    rays = np.tile(one, (4096, 1))
    cpu = O.trace_rays(hs, rays, seed=5)
    d = hs.desc.contents
    iso = np.array([d.materials[i].kind == _capi.RT_MAT_ISOTROPIC for i in range(d.n_materials)])
    in_medium = (cpu["material"] >= 0) & iso[np.maximum(cpu["material"], 0)]
    assert in_medium.mean() > 0.3  # the ray does cross a smoke box, and a good part of the copies scatter inside it
    for path in (_capi.RT_PATH_EXACT, _capi.RT_PATH_FAST):
        sc.set_path(trace=path)
        g = api.trace_rays(sc, rays, seed=5)
        gm = (g["material"] >= 0) & iso[np.maximum(g["material"], 0)]
        assert (gm == in_medium).mean() > 0.995
        both = gm & in_medium
        assert np.allclose(g["t"][both], cpu["t"][both], rtol=1e-4)
        assert len(np.unique(g["t"][gm])) > 0.99 * gm.sum()  # no two indices share a stream
        other = api.trace_rays(sc, rays, seed=6)
        om = (other["material"] >= 0) & iso[np.maximum(other["material"], 0)]
        assert not np.allclose(other["t"][both & om], g["t"][both & om], rtol=1e-4)
    sc.close()


@pytest.mark.parametrize("n,spps", [(64, (256, 1024, 4096, 65536)), (256, (256, 4096, 65536))])
def test_pixel_noise_falls_as_one_over_spp_up_to_65536(n, spps):
    """A path's stream is keyed by 64 bits, (sample, seed) and the pixel (include/rt_rng.h; round 2: 32 bits in all, so at
    this test's 256 x 256 x 131 072 = 8.6e9 paths every key was shared by two paths of different pixels).  Shared or
    correlated streams would show as a noise floor: the mean squared difference of two renders over disjoint sample
    ranges (same seed), times spp, must stay what it is at a few hundred samples.  (Caustic pixels make that product
    noisy, about +-25 % from size to size; a floor at the level of N samples would multiply it by 65536 / N.)"""
    sc = Scene(HostScene("cornell_box", 1.0), 0)
    v = {}
    for spp in spps:
        a = gpu_render(sc, n, n, 0, spp, seed=77)[..., 1:] / spp
        b = gpu_render(sc, n, n, spp, 2 * spp, seed=77)[..., 1:] / spp
        v[spp] = float(((a - b) ** 2).mean()) * spp
    sc.close()
    print({k: f"{x:.3f}" for k, x in v.items()})
    low = np.mean([v[k] for k in spps[:-1]])
    assert 0.4 * low <= v[65536] <= 1.6 * low, v


def test_whole_image_statistics_equal_the_reference_at_1024_spp():
    """north_star check 2, pixel-level form.  If the GPU estimator and the reference's are the same random
    variable per pixel, the reference's 1024-spp image must sit exactly as far from the truth as a GPU 1024-spp
    image does.  Truth = a 65 536-spp GPU render.  Tolerance: the two PSNRs agree within 0.35 dB (one figure
    would move by 3 dB for a factor 2 in per-pixel variance, and any local bias only ever lowers the first)."""
    path = os.path.join(O.GOLD, "cornell_box_256x256_1024spp_r50.npy")
    if not os.path.exists(path):
        pytest.skip("1024-spp reference fixture not generated")
    ref = np.load(path)
    n = ref.shape[0]
    sc = Scene(HostScene("cornell_box", 1.0), 0)
    truth, own = _gpu_rgb(sc, n, 65536, 41), _gpu_rgb(sc, n, 1024, 42)
    sc.close()
    p_ref, p_own = _psnr(truth, ref), _psnr(truth, own)
    print(f"PSNR vs 65536-spp GPU render: reference 1024 spp {p_ref:.2f} dB, GPU 1024 spp {p_own:.2f} dB")
    assert abs(p_ref - p_own) <= 0.35


def test_psnr_against_the_converged_reference_and_why_40_db_is_out_of_reach():
    """BASELINE.json asks for PSNR >= 40 dB against the reference at 4096 spp.  The reference cannot supply an
    image that good: its RNG is libstdc++'s minstd_rand0 (src/main.cpp:62), period 2^31-2 draws, which a
    256x256 render exhausts after about 1300 spp; from then on it replays the same numbers.  Evidence: the first
    and the second 4096 samples of the reference's 8192-spp run agree to > 45 dB where two independent 4096-spp
    renders agree to 37.5 dB.  Its "converged" image therefore keeps the noise of about 1300 spp, and the best any
    renderer can score against it is about 37.4 dB, which both GPU paths reach with a 65 536-spp render."""
    p4 = os.path.join(O.GOLD, "cornell_box_256x256_4096spp_r50.npy")
    p8 = os.path.join(O.GOLD, "cornell_box_256x256_8192spp_r50.npy")
    if not (os.path.exists(p4) and os.path.exists(p8)):
        pytest.skip("converged reference fixtures not generated")
    ref4, ref8 = np.load(p4), np.load(p8)
    second_half = 2 * ref8 - ref4
    replay = _psnr(ref4, second_half)
    sc = Scene(HostScene("cornell_box", 1.0), 0)
    a, b = _gpu_rgb(sc, 256, 4096, 51), _gpu_rgb(sc, 256, 4096, 52)
    truth = _gpu_rgb(sc, 256, 65536, 53)
    sc.close()
    independent = _psnr(a, b)
    best = _psnr(truth, ref8)
    print(f"reference first vs second 4096 samples {replay:.2f} dB; two independent GPU 4096-spp renders {independent:.2f} dB; "
          f"GPU 65536 spp vs reference 8192 spp {best:.2f} dB")
    assert replay > independent + 6.0   # the reference's second half is (mostly) a replay of its first
    assert best >= independent - 1.0    # as close to the reference as an independent unbiased 4096-spp render would be


@pytest.mark.parametrize("path", ["fast", "exact"])
def test_psnr_40_db_against_the_reference_inside_its_rng_period(path):
    """north_star check 2 as stated: PSNR >= 40 dB against the reference at 4096 spp.  The fixture is the reference's own
    96x96 render (oracle/make_golden.py --image96): 96*96*4096 paths draw about 0.95e9 numbers, inside the period of its
    minstd_rand0 (2^31-2), so its 4096 samples are 4096 independent samples -- unlike the 256x256 fixture, which replays
    after about 1300 spp (next test).  The GPU image is a 262 144-spp render, so what is measured is the noise of the
    reference's 4096 samples plus any bias of the GPU estimator; 8-bit images after the reference's sqrt gamma."""
    ref = np.load(os.path.join(O.GOLD, "cornell_box_96x96_4096spp_r50.npy"))
    n = ref.shape[0]
    sc = Scene(HostScene("cornell_box", 1.0), 0)
    sc.set_path(render=PATHS[path])
    batches = [_gpu_rgb(sc, n, 16384, 60 + k) for k in range(16)]
    gpu = np.mean(batches, axis=0)
    assert sc.info(_capi.RT_INFO_LAST_RENDER_PATH) == PATHS[path]
    sc.close()
    psnr = _psnr(gpu, ref)
    print(f"PSNR {path} GPU 262144 spp vs reference 4096 spp at 96x96: {psnr:.2f} dB")
    assert psnr >= 40.0
    # per-channel image means within 3 standard errors of the reference's 8192-spp image: the standard error of a
    # 16 384-spp image mean from the spread of the 16 batches, scaled to 8192 samples (the GPU's own error is 1/32 of that)
    ref8 = np.load(os.path.join(O.GOLD, "cornell_box_96x96_8192spp_r50.npy"))
    se8192 = np.std([b.mean(axis=(0, 1)) for b in batches], axis=0, ddof=1) * np.sqrt(2.0)
    diff = np.abs(gpu.mean(axis=(0, 1)) - ref8.mean(axis=(0, 1)))
    print(f"mean diff {diff}, 3 SE {3 * se8192}")
    assert np.all(diff <= 3.0 * se8192 * np.sqrt(1.0 + 1.0 / 32.0)), (diff, se8192)


def test_reference_96x96_fixture_does_not_replay():
    """The two halves of the reference's 8192-spp 96x96 run are independent (no replay of the random stream inside
    1.9e9 draws): they agree no better than two independent GPU 4096-spp renders do."""
    ref4 = np.load(os.path.join(O.GOLD, "cornell_box_96x96_4096spp_r50.npy"))
    ref8 = np.load(os.path.join(O.GOLD, "cornell_box_96x96_8192spp_r50.npy"))
    halves = _psnr(ref4, 2 * ref8 - ref4)
    sc = Scene(HostScene("cornell_box", 1.0), 0)
    independent = _psnr(_gpu_rgb(sc, 96, 4096, 71), _gpu_rgb(sc, 96, 4096, 72))
    sc.close()
    print(f"reference halves {halves:.2f} dB, two independent GPU 4096-spp renders {independent:.2f} dB")
    assert abs(halves - independent) < 1.0


# ---- scenes beyond the Cornell box (ray_tracing_amd/host/rt_scenes_ext.h): media, moving spheres, textures, lens,
# ---- nested BVHs under instance chains.  They run on the exact path (the record table is for small static scenes).
EXT = ["cornell_box_volumes", "random_spheres", "next_week_final", "sphere_field_4k", "simple_scene", "another_simple",
       "two_perlin_spheres"]


@pytest.mark.parametrize("scene", EXT)
def test_ext_scene_trace_equals_oracle_and_reference(scene):
    tab = np.load(os.path.join(O.GOLD, f"{scene}.npz"))["hits_64x48"].reshape(-1, 21)
    hs = HostScene(scene, 64 / 48)
    sc = Scene(hs, 0)
    # too large (or too rich) for the record table; the fast path for them is the per-lane one (rt_fastlanes.hpp)
    assert sc.info(_capi.RT_INFO_FAST_PATH_AVAILABLE) == 1
    assert (sc.info(_capi.RT_INFO_RECORDS) > 0) == (scene == "two_perlin_spheres")  # the only one small and plain enough
    rays = tab[:, 2:9].copy()
    g, c = api.trace_rays(sc, rays, seed=5), O.trace_rays(hs, rays, seed=5)
    sc.close()
    d = hs.desc.contents
    iso = np.array([d.materials[i].kind == _capi.RT_MAT_ISOTROPIC for i in range(d.n_materials)])
    gm = (g["material"] >= 0) & iso[np.maximum(g["material"], 0)]
    cm = (c["material"] >= 0) & iso[np.maximum(c["material"], 0)]
    # same random stream on both sides; a medium's t goes through logf (device vs glibc: last-bit differences), which can
    # flip which of two near-equal candidates wins on a handful of rays
    same = g["prim_id"] == c["prim_id"]
    assert same.mean() >= 0.998, f"{(~same).sum()} ids differ"
    solid = same & ~gm & ~cm & (c["prim_id"] >= 0)
    assert np.array_equal(bits(g["t"][solid]), bits(c["t"][solid]))
    assert np.array_equal(bits(g["p"][solid]), bits(c["p"][solid]))
    assert np.array_equal(bits(g["normal"][solid]), bits(c["normal"][solid]))
    med = same & gm & cm
    assert np.allclose(g["t"][med], c["t"][med], rtol=1e-5)
    # and against the reference's own table wherever no medium is involved on either side
    ref_solid = (tab[:, 20] < 0.5) & ~gm
    assert np.array_equal(g["prim_id"][ref_solid], tab[ref_solid, 10].astype(np.int32))
    m = ref_solid & (tab[:, 9] > 0.5)
    assert np.array_equal(bits(g["t"][m]), bits(tab[m, 11]))
    # (u, v) are produced on the device only where a texture reads them (the earth sphere): atan2f / asinf, not bit-exact
    uv = solid & (g["u"] != 0)
    assert np.allclose(g["u"][uv], c["u"][uv], atol=2e-6) and np.allclose(g["v"][uv], c["v"][uv], atol=2e-6)
    if scene == "next_week_final":
        assert uv.sum() > 10


@pytest.mark.parametrize("scene,nx,ny,ns", [("cornell_box_volumes", 32, 24, 8), ("random_spheres", 48, 32, 4), ("next_week_final", 32, 24, 4), ("sphere_field_4k", 32, 24, 4),
                                                ("simple_scene", 32, 24, 4), ("another_simple", 32, 24, 4), ("two_perlin_spheres", 32, 24, 4)])
def test_ext_scene_render_matches_oracle_on_the_same_stream(scene, nx, ny, ns):
    """Same tolerance as on the Cornell box; these scenes add logf / sinf / atan2f (media, perlin, image uv) on the way,
    so a few more paths diverge: >= 93 % of pixels within 1e-3, image mean within 3 %."""
    hs = HostScene(scene, nx / ny)
    sc = Scene(hs, 0)
    sc.set_path(render=_capi.RT_PATH_EXACT)
    g = gpu_render(sc, nx, ny, 0, ns, seed=9)
    assert sc.info(_capi.RT_INFO_LAST_RENDER_PATH) == _capi.RT_PATH_EXACT
    sc.close()
    c, _, _ = O.render(hs, nx, ny, 0, ns, 1, 50, O.ORC_RNG_COUNTER, 9, 8)
    close = np.isclose(g[..., 1:], c[..., 1:], rtol=1e-3, atol=1e-4).all(axis=-1)
    assert close.mean() >= 0.93, f"only {close.mean():.4f} of pixels agree"
    gm, cmn = g[..., 1:].mean(axis=(0, 1)), c[..., 1:].mean(axis=(0, 1))
    assert np.allclose(gm, cmn, rtol=3e-2, atol=1e-4), (gm, cmn)


def _fast_hits_vs_oracle(hs, fast, cpu, min_same=0.999):
    """The per-lane fast path against the CPU oracle DIRECTLY (not through the exact HIP path): ids equal on >= 99.9 % of
    the rays that end on a solid surface on both sides, t within 1e-3 relative there."""
    d = hs.desc.contents
    iso = np.array([d.materials[i].kind == _capi.RT_MAT_ISOTROPIC for i in range(d.n_materials)])
    fm = (fast["material"] >= 0) & iso[np.maximum(fast["material"], 0)]
    cm = (cpu["material"] >= 0) & iso[np.maximum(cpu["material"], 0)]
    solid = ~fm & ~cm
    same = fast["prim_id"][solid] == cpu["prim_id"][solid]
    assert same.mean() >= min_same, f"{(~same).sum()} of {solid.sum()} ids differ from the oracle's"
    m = np.flatnonzero(solid)[same & (cpu["prim_id"][solid] >= 0)]
    assert len(m) > 1000
    assert np.allclose(fast["t"][m], cpu["t"][m], rtol=1e-3, atol=1e-3)
    assert np.array_equal(fast["material"][m], cpu["material"][m])
    assert abs(fm.mean() - cm.mean()) < 0.02


@pytest.mark.parametrize("scene", EXT)
def test_ext_scene_fast_trace_close_to_exact(scene):
    """The per-lane fast path (ordered, culled walk, hardware-rate arithmetic) finds the same closest hits as the exact
    path up to rounding.  Media draw random numbers, so their hits are compared only through which rays they involve."""
    tab = np.load(os.path.join(O.GOLD, f"{scene}.npz"))["hits_64x48"].reshape(-1, 21)
    hs = HostScene(scene, 64 / 48)
    sc = Scene(hs, 0)
    rng = np.random.default_rng(17)
    n = 20000
    lo, hi = (0.0, 555.0) if scene in ("cornell_box_volumes", "next_week_final") else (-8.0, 8.0)
    o = rng.uniform(lo, hi, size=(n, 3)).astype(np.float32)
    if scene in ("random_spheres", "simple_scene", "two_perlin_spheres"):
        o[:, 1] = np.abs(o[:, 1]) * 0.5 + 0.05  # above the ground sphere
    if scene == "another_simple":
        o[:, 1] = 17.0 + o[:, 1] * 0.5
    if scene == "next_week_final":
        # above the ground boxes (height <= 101): neighbouring boxes share walls, and a ray that starts INSIDE one meets two
        # coincident faces at the same t, where the two walks are free to pick different winners (ties: rt_fastlanes.hpp)
        o[:, 1] = 110.0 + o[:, 1] * 0.8
    dvec = rng.normal(size=(n, 3)).astype(np.float32)
    rays = np.concatenate([tab[:, 2:9].astype(np.float32), np.concatenate([o, dvec, rng.uniform(size=(n, 1)).astype(np.float32)], axis=1)])
    exact = api.trace_rays(sc, rays, seed=5)
    sc.set_path(trace=_capi.RT_PATH_FAST)
    fast = api.trace_rays(sc, rays, seed=5)
    sc.close()
    cpu = O.trace_rays(hs, rays, seed=5)
    _fast_hits_vs_oracle(hs, fast, cpu)
    d = hs.desc.contents
    iso = np.array([d.materials[i].kind == _capi.RT_MAT_ISOTROPIC for i in range(d.n_materials)])
    em = (exact["material"] >= 0) & iso[np.maximum(exact["material"], 0)]
    fm = (fast["material"] >= 0) & iso[np.maximum(fast["material"], 0)]
    solid = ~em & ~fm
    assert solid.mean() > 0.3
    same = exact["prim_id"][solid] == fast["prim_id"][solid]
    assert same.mean() >= 0.999, f"{(~same).sum()} of {solid.sum()} ids differ"
    m = np.flatnonzero(solid)[same & (exact["prim_id"][solid] >= 0)]
    assert np.allclose(fast["t"][m], exact["t"][m], rtol=1e-3, atol=1e-3)
    assert np.array_equal(fast["material"][m], exact["material"][m])
    # float32 loses digits on the radius-1000 ground sphere and on grazing hits of small spheres seen from afar
    # (b*b - a*c cancels), differently on the two paths; a normal is that error divided by the radius
    dn = np.abs(fast["normal"][m] - exact["normal"][m]).max(axis=1)
    assert (dn <= 2e-3).mean() >= 0.98 and dn.max() <= 0.05, (dn.max(), (dn > 2e-3).sum())
    dp = np.abs(fast["p"][m] - exact["p"][m]).max(axis=1)
    assert (dp <= 0.05).mean() >= 0.995 and dp.max() <= 5.0, (dp.max(), (dp > 0.05).sum())
    uv = m[exact["u"][m] != 0]
    assert np.allclose(fast["u"][uv], exact["u"][uv], atol=1e-3) and np.allclose(fast["v"][uv], exact["v"][uv], atol=1e-3)
    # the fraction of rays that end inside a medium is a property of the scene, not of the path
    assert abs(em.mean() - fm.mean()) < 0.02


@pytest.mark.parametrize("scene,nx,ny,spp", [("cornell_box_volumes", 64, 64, 64), ("random_spheres", 96, 64, 64), ("next_week_final", 64, 64, 32), ("sphere_field_4k", 64, 64, 64),
                                                 ("simple_scene", 64, 64, 64), ("another_simple", 64, 64, 64), ("two_perlin_spheres", 64, 64, 64)])
def test_ext_scene_fast_render_statistics(scene, nx, ny, spp):
    """The per-lane fast path (what AUTO picks for scenes that do not fit the record table).  Its image is an independent unbiased estimate of the same
    integral as the exact path's: per-channel image means agree within 3 standard errors (estimated from 8 batches per path)."""
    sc = Scene(HostScene(scene, nx / ny), 0)
    means = {}
    for name, path in (("fast", _capi.RT_PATH_FAST), ("exact", _capi.RT_PATH_EXACT)):
        sc.set_path(render=path)
        b = np.array([gpu_render(sc, nx, ny, k * spp, (k + 1) * spp, seed=31 if name == "fast" else 32)[..., 1:].mean(axis=(0, 1)) / spp
                      for k in range(8)])
        assert sc.info(_capi.RT_INFO_LAST_RENDER_PATH) == (_capi.RT_PATH_FAST if name == "fast" else _capi.RT_PATH_EXACT)
        means[name] = (b.mean(axis=0), b.std(axis=0, ddof=1) / np.sqrt(len(b)))
    sc.close()
    diff = np.abs(means["fast"][0] - means["exact"][0])
    se = np.sqrt(means["fast"][1] ** 2 + means["exact"][1] ** 2)
    assert np.all(diff <= 3.0 * se + 1e-5), (scene, diff, se, means)


@pytest.mark.parametrize("scene,nx,ny,ns", [("random_spheres", 48, 32, 4), ("sphere_field_4k", 32, 24, 4), ("simple_scene", 32, 24, 4),
                                             ("two_perlin_spheres", 32, 24, 4)])
def test_ext_scene_fast_render_matches_oracle_on_the_same_stream(scene, nx, ny, ns):
    """The FAST path (per-lane walk of the library's own tree, or the record table for two_perlin_spheres) against the CPU
    oracle on the same per-path random stream, for the scenes without media (a medium draws per visit, and the fast walk
    skips media behind the best hit, so its stream differs by design there: those scenes are compared statistically).
    Same draws in the same order along a path, so images agree pixel by pixel until an ulp flips a comparison somewhere
    along a path: >= 90 % of pixels within 2e-3, image mean within 3 %.  (The sphere field: >= 80 %.  Its paths are a dozen
    bounces between curved mirrors and glass balls, where a last-bit difference in one hit point doubles at every bounce.)"""
    hs = HostScene(scene, nx / ny)
    sc = Scene(hs, 0)
    sc.set_path(render=_capi.RT_PATH_FAST)
    g = gpu_render(sc, nx, ny, 0, ns, seed=9)
    assert sc.info(_capi.RT_INFO_LAST_RENDER_PATH) == _capi.RT_PATH_FAST
    sc.close()
    c, _, _ = O.render(hs, nx, ny, 0, ns, 1, 50, O.ORC_RNG_COUNTER, 9, 8)
    assert np.all(g[..., 0] == ns)
    close = np.isclose(g[..., 1:], c[..., 1:], rtol=2e-3, atol=2e-4).all(axis=-1)
    assert close.mean() >= (0.80 if scene == "sphere_field_4k" else 0.90), f"only {close.mean():.4f} of pixels agree"
    gm, cmn = g[..., 1:].mean(axis=(0, 1)), c[..., 1:].mean(axis=(0, 1))
    # (the sphere field's fifth of diverged paths are independent samples on the two sides: 3072 paths, standard error 2.4 %)
    assert np.allclose(gm, cmn, rtol=8e-2 if scene == "sphere_field_4k" else 3e-2, atol=1e-4), (gm, cmn)


def test_sphere_field_1m_baseline_config_5():
    """BASELINE.json configs[4]: the procedural 1M-sphere scene.  Its tree is deeper than 20 levels, so the walk runs the
    four-steps-per-look instantiation; closest hits of the fast walk, of the exact walk (the reference's own 2M-entry tree,
    ordered and culled) and of the CPU oracle must agree; a full-size 1920x1080 frame must come out whole (every pixel
    sampled, finite, not black); fast and exact renders are the same estimator (means within 3 standard errors)."""
    nx, ny = 1920, 1080
    hs = HostScene("sphere_field_1m", nx / ny)
    sc = Scene(hs, 0)
    assert sc.info(_capi.RT_INFO_RECORDS) == -1 and sc.info(_capi.RT_INFO_BVH_LEVELS) > 20
    rng = np.random.default_rng(77)
    n = 20000
    cam = api.camera_rays(sc, rng.uniform(size=(n // 2, 2)).astype(np.float32))
    o = rng.uniform(-40, 40, size=(n // 2, 3)).astype(np.float32)
    o[:, 1] = np.abs(o[:, 1]) * 0.5 + 1.0
    rays = np.concatenate([cam, np.concatenate([o, rng.normal(size=(n // 2, 3)).astype(np.float32), np.zeros((n // 2, 1), np.float32)], axis=1)])
    exact = api.trace_rays(sc, rays)
    sc.set_path(trace=_capi.RT_PATH_FAST)
    fast = api.trace_rays(sc, rays)
    cpu = O.trace_rays(hs, rays)
    assert (cpu["prim_id"] >= 0).mean() > 0.5
    assert np.array_equal(exact["prim_id"], cpu["prim_id"])
    mm = cpu["prim_id"] >= 0
    assert np.array_equal(bits(exact["t"][mm]), bits(cpu["t"][mm]))
    _fast_hits_vs_oracle(hs, fast, cpu)
    # one full-size frame at 1 spp through the path AUTO picks
    frame = gpu_render(sc, nx, ny, 0, 1, seed=5)
    assert sc.info(_capi.RT_INFO_LAST_RENDER_PATH) == _capi.RT_PATH_FAST
    assert np.all(frame[..., 0] == 1) and np.isfinite(frame).all() and (frame[..., 1:] >= 0).all()
    assert frame[..., 1:].mean() > 0.01
    # the same estimator on both paths, at a small size
    sx, sy, spp = 96, 54, 16
    means = {}
    for name, path in (("fast", _capi.RT_PATH_FAST), ("exact", _capi.RT_PATH_EXACT)):
        sc.set_path(render=path)
        b = np.array([gpu_render(sc, sx, sy, k * spp, (k + 1) * spp, seed=31 if name == "fast" else 32)[..., 1:].mean(axis=(0, 1)) / spp
                      for k in range(8)])
        assert sc.info(_capi.RT_INFO_LAST_RENDER_PATH) == path
        means[name] = (b.mean(axis=0), b.std(axis=0, ddof=1) / np.sqrt(len(b)))
    sc.close()
    diff = np.abs(means["fast"][0] - means["exact"][0])
    se = np.sqrt(means["fast"][1] ** 2 + means["exact"][1] ** 2)
    assert np.all(diff <= 3.0 * se + 1e-5), (diff, se, means)


@pytest.mark.parametrize("scene,nx,ny", [("random_spheres", 1200, 800), ("next_week_final", 800, 800)])
def test_baseline_configs_1_and_3_at_their_full_image_size(scene, nx, ny):
    """BASELINE.json configs[0] and configs[2] at their full image sizes (1200 x 800 book-1 spheres, 800 x 800 Next-Week final),
    4 spp, through the path AUTO picks (the per-lane walk of the library's own tree): every pixel gets its samples, nothing is
    NaN or negative, the two halves of the sample range add up to the whole, the image is not black, and its mean is that of
    a small render of the same camera (a sixteenth of the pixels, more samples)."""
    sc = Scene(HostScene(scene, nx / ny), 0)
    spp = 4
    full = gpu_render(sc, nx, ny, 0, spp, seed=2)
    assert sc.info(_capi.RT_INFO_LAST_RENDER_PATH) == _capi.RT_PATH_FAST and sc.info(_capi.RT_INFO_RECORDS) == -1
    assert full.shape == (ny, nx, 4) and np.all(full[..., 0] == spp) and np.isfinite(full).all() and (full[..., 1:] >= 0).all()
    halves = gpu_render(sc, nx, ny, 0, 2, seed=2) + gpu_render(sc, nx, ny, 2, spp, seed=2)
    assert np.allclose(full, halves, rtol=1e-4, atol=1e-4)
    assert (full[..., 1:].sum(axis=-1) > 0).mean() > 0.2  # not black (the Next-Week room is lit by one small lamp: a third of the pixels at 4 spp)
    small = gpu_render(sc, nx // 4, ny // 4, 0, 64, seed=9)
    sc.close()
    m_full, m_small = full[..., 1:].mean(axis=(0, 1)) / spp, small[..., 1:].mean(axis=(0, 1)) / 64
    assert np.allclose(m_full, m_small, rtol=0.05), (m_full, m_small)


def test_headline_size_takes_the_eight_wave_block_and_equals_the_four_wave_image():
    """BASELINE.json configs[1] at its full image size (Cornell 1024 x 1024), 128 spp: a launch of 2^27 paths, which AUTO runs
    on the block kernel with blocks of eight waves (32 x 16 tiles).  Every pixel gets its samples, nothing is NaN or negative,
    and the image is that of the four-wave blocks (RT_OPT_BLOCK_WAVES = 4) up to float summation order -- paths are keyed by
    (pixel, sample), whatever the block that traces them."""
    nx = ny = 1024
    spp = 128
    sc = Scene(HostScene("cornell_box", 1.0), 0)
    eight = gpu_render(sc, nx, ny, 0, spp, seed=4)
    assert sc.info(_capi.RT_INFO_LAST_RENDER_PATH) == _capi.RT_PATH_FAST
    sc.set_option(_capi.RT_OPT_BLOCK_WAVES, 4)
    four = gpu_render(sc, nx, ny, 0, spp, seed=4)
    sc.close()
    assert np.all(eight[..., 0] == spp) and np.isfinite(eight).all() and (eight[..., 1:] >= 0).all()
    close = np.isclose(eight[..., 1:], four[..., 1:], rtol=2e-4, atol=1e-3).all(axis=-1)
    assert close.mean() >= 0.999, f"only {close.mean():.5f} of pixels agree"
    assert np.allclose(eight[..., 1:].mean(axis=(0, 1)), four[..., 1:].mean(axis=(0, 1)), rtol=1e-4)


@pytest.mark.parametrize("scene,lo,hi", [("cornell_box_volumes", 0.0, 555.0), ("next_week_final", 0.0, 555.0)])
def test_fast_path_medium_free_flight_has_the_oracles_distribution(scene, lo, hi):
    """constant_medium::hit (src/volume.h:23-53) on the per-lane fast path, where it is NOT the exact path's code: the boundary's
    two hits come from a slab test (smoke boxes) or two sphere tests, the free-flight distance from the hardware log, a lone child
    of a bvh_node draws twice, and a medium behind the best hit is not visited.  The hit distance is random, so it is compared as a
    distribution: for every medium of the scene, the t of the rays that end in it on the fast path against the oracle's (another
    random stream), two-sample Kolmogorov-Smirnov; and the share of the rays each medium takes."""
    from scipy.stats import ks_2samp
    hs = HostScene(scene, 1.0)
    sc = Scene(hs, 0)
    rng = np.random.default_rng(23)
    n = 120000
    o = rng.uniform(lo, hi, size=(n, 3)).astype(np.float32)
    if scene == "next_week_final":
        o[:, 1] = 110.0 + o[:, 1] * 0.8
        o[: n // 2] = (np.array([360.0, 150.0, 145.0]) + rng.normal(size=(n // 2, 3)) * 45.0).astype(np.float32)  # in and around the blue sphere
    d = rng.normal(size=(n, 3)).astype(np.float32)
    rays = np.concatenate([o, d, rng.uniform(size=(n, 1)).astype(np.float32)], axis=1)
    sc.set_path(trace=_capi.RT_PATH_FAST)
    fast = api.trace_rays(sc, rays, seed=41)
    sc.close()
    cpu = O.trace_rays(hs, rays, seed=77)
    desc = hs.desc.contents
    iso = np.array([desc.materials[i].kind == _capi.RT_MAT_ISOTROPIC for i in range(desc.n_materials)])
    fm = (fast["material"] >= 0) & iso[np.maximum(fast["material"], 0)]
    cm = (cpu["material"] >= 0) & iso[np.maximum(cpu["material"], 0)]
    assert cm.mean() > 0.05, cm.mean()
    checked = 0
    for pid in np.unique(cpu["prim_id"][cm]):
        a, b = fast["t"][fm & (fast["prim_id"] == pid)], cpu["t"][cm & (cpu["prim_id"] == pid)]
        share_f, share_c = len(a) / n, len(b) / n
        assert abs(share_f - share_c) <= 4.0 * np.sqrt(max(share_c, 1e-4) / n) + 5e-4, (pid, share_f, share_c)
        if len(b) >= 400:
            p = ks_2samp(a.astype(np.float64), b.astype(np.float64)).pvalue
            assert p > 1e-4, f"{scene} medium {pid}: t differs in distribution (KS p = {p:.2e}, {len(a)} vs {len(b)} rays)"
            checked += 1
    assert checked >= 1
