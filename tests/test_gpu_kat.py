"""Known-answer tests of the sampling / shading FUNCTIONS on the device (rt_debug_eval, include/rt_amd.h): the reference-written
vectors of tests/golden/kat.npz (the ones tests/test_oracle_pinned.py pins the oracle with) and the oracle itself, against
both device families -- rt:: (RT_PATH_EXACT) and rtf:: (RT_PATH_FAST).  Whole renders only say THAT something is off; these say
which function.  Bars: the exact family is bit-equal to the reference where no libm call is involved and within a few ulp
where one is (sinf / cosf / atan2f / asinf come from another libm on the device); the fast family within rtol 1e-5 .. 1e-4
(hardware reciprocal, square root, sine; FMA contraction).

    src/plane.h:76-98, src/sphere.h:68-103 (pdf_value, random)   src/pdf.h:4-74   src/vec3.h:173-224 (onb, cosine direction)
    src/utils.h:4-68 (schlick, reflect, refract, rejection samplers, sphere uv)   src/perlin.h:54-136   src/texture.h:9-81
    src/material.h (the five scatter functions) and src/main.cpp:91-127 (one bounce of color())
"""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O
from ray_tracing_amd import _capi, api
from ray_tracing_amd.api import HostScene, Scene

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def same_bits(a, b):
    """bit-equal, any NaN equal to any NaN (the sign and payload of a NaN are not arithmetic)"""
    a, b = np.ascontiguousarray(a, dtype=np.float32), np.ascontiguousarray(b, dtype=np.float32)
    return bool(np.all((bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))))


def ulps(a, b):
    """distance in float32 steps (same-sign finite values)"""
    a, b = bits(a).astype(np.int64), bits(b).astype(np.int64)
    return np.abs(a - b)


def items(n):
    return np.zeros((n, 16), dtype=np.float32)


@pytest.fixture(scope="module")
def kat():
    return np.load(os.path.join(O.GOLD, "kat.npz"))


@pytest.fixture(scope="module")
def cornell():
    hs = HostScene("cornell_box", 1.0)
    sc = Scene(hs, 0)
    yield hs, sc
    sc.close()


def test_light_pdf_values_on_the_device(kat, cornell):
    """xz_rect::pdf_value, sphere::pdf_value, cosine_pdf::value: exact family bit-equal to the reference's numbers, fast family
    (for the normalised direction: the value does not depend on the direction's length) within 2e-5."""
    _, sc = cornell
    k = kat["kat_pdf"]
    it = items(len(k))
    it[:, 0] = k[:, 0]
    it[:, 1:4], it[:, 4:7], it[:, 7:12], it[:, 12:15] = k[:, 1:4], k[:, 4:7], k[:, 7:12], k[:, 13:16]
    out = api.debug_eval(sc, _capi.RT_EVAL_LIGHT_PDF, it)
    assert same_bits(out[:, 0], k[:, 12]), f"{(bits(out[:, 0]) != bits(k[:, 12])).sum()} pdf values differ"
    assert same_bits(out[:, 1], k[:, 16])
    assert (k[:, 12] > 0).sum() > 100
    # fast family: the hit / no-hit decision may differ on a ray that grazes the light's edge; none of these does
    assert ((out[:, 2] > 0) == (k[:, 12] > 0)).mean() >= 0.995
    both = (out[:, 2] > 0) & (k[:, 12] > 0) & np.isfinite(k[:, 12])
    assert np.allclose(out[both, 2], k[both, 12], rtol=2e-5)
    assert np.allclose(out[:, 3], k[:, 16], rtol=2e-5, atol=1e-6)


def test_onb_unit_vector_and_sphere_uv_on_the_device(kat, cornell):
    _, sc = cornell
    k = kat["kat_onb_uv"]
    it = items(len(k))
    it[:, 0:3] = k[:, 0:3]
    out = api.debug_eval(sc, _capi.RT_EVAL_ONB, it)
    assert np.array_equal(bits(out[:, 0:12]), bits(k[:, 3:15]))  # u, v, w, unit(n): divisions and square roots only
    # get_sphere_uv: atan2f / asinf of the device's libm, then a sum that cancels near v = 0: an absolute bar
    assert np.abs(out[:, 12:14] - k[:, 15:17]).max() <= 2e-7
    it[:, 3:6] = k[:, 0:3][::-1]
    fast = api.debug_eval(sc, _capi.RT_EVAL_ONB_FAST, it)
    assert np.allclose(fast[:, 0:12], k[:, 3:15], rtol=1e-5, atol=2e-6)
    u, v, w = k[:, 3:6], k[:, 6:9], k[:, 9:12]
    a = it[:, 3:6]
    local = a[:, 0:1] * u + a[:, 1:2] * v + a[:, 2:3] * w  # onb::local, src/vec3.h:203-205
    assert np.allclose(fast[:, 12:15], local, rtol=1e-5, atol=1e-5 * np.abs(a).max())


def test_reflect_refract_schlick_on_the_device(kat, cornell):
    _, sc = cornell
    k = kat["kat_optics"]
    it = items(len(k))
    it[:, 0:9] = k[:, 0:9]
    out = api.debug_eval(sc, _capi.RT_EVAL_OPTICS, it)
    assert np.array_equal(bits(out[:, 0:7]), bits(k[:, 9:16]))  # reflect, refract: bit-equal
    assert (k[:, 12] > 0).sum() > 50 and (k[:, 12] == 0).sum() > 20  # refraction and total internal reflection both occur
    assert ulps(out[:, 7], k[:, 16]).max() <= 2  # schlick: the reference calls powf(x, 5), the kernels multiply


def test_perlin_noise_and_turbulence_on_the_device(kat, cornell):
    _, sc = cornell
    k = kat["kat_perlin"]
    it = items(len(k))
    it[:, 0:3] = k[:, 0:3]
    out = api.debug_eval(sc, _capi.RT_EVAL_PERLIN, it)
    assert np.array_equal(bits(out[:, 0]), bits(k[:, 3])) and np.array_equal(bits(out[:, 1]), bits(k[:, 4]))
    assert np.allclose(out[:, 2], k[:, 3], rtol=1e-5, atol=2e-6) and np.allclose(out[:, 3], k[:, 4], rtol=1e-5, atol=4e-6)
    # the fast family's second form (eight corners written out: the record-table kernel for textured scenes)
    assert np.allclose(out[:, 4], k[:, 3], rtol=1e-5, atol=2e-6) and np.allclose(out[:, 5], k[:, 4], rtol=1e-5, atol=4e-6)


def test_textures_on_the_device(kat, cornell):
    """image_texture / checker_texture / noise_texture::value on the harness's synthetic table (7 x 5 image, a checker of black
    and white, two noise scales): a Cornell scene description with its texture table swapped for that one."""
    hs, _ = cornell
    W, H = 7, 5
    px = ((np.arange(W * H * 3) * 37 + 11) & 255).astype(np.uint8)
    T = _capi.rt_texture
    tex = [T(_capi.RT_TEX_IMAGE, -1, -1, W, H, 0, 0, (C.c_float * 3)(0, 0, 0), 0.0),
           T(_capi.RT_TEX_CONSTANT, -1, -1, 0, 0, 0, 0, (C.c_float * 3)(0, 0, 0), 0.0),
           T(_capi.RT_TEX_CONSTANT, -1, -1, 0, 0, 0, 0, (C.c_float * 3)(1, 1, 1), 0.0),
           T(_capi.RT_TEX_CHECKER, 1, 2, 0, 0, 0, 0, (C.c_float * 3)(0, 0, 0), 0.0),
           T(_capi.RT_TEX_NOISE, -1, -1, 0, 0, 0, 0, (C.c_float * 3)(0, 0, 0), 0.1),
           T(_capi.RT_TEX_NOISE, -1, -1, 0, 0, 0, 0, (C.c_float * 3)(0, 0, 0), 2.0)]
    d = _capi.rt_scene_desc.from_buffer_copy(hs.desc.contents)
    assert d.n_textures <= len(tex)  # the Cornell materials' texture indices stay valid
    arr = (T * len(tex))(*tex)
    buf = (C.c_uint8 * len(px)).from_buffer_copy(px.tobytes())
    d.textures, d.n_textures, d.texels, d.n_texel_bytes = arr, len(tex), buf, len(px)
    lib = _capi.load()
    h = C.c_void_p()
    assert lib.rt_scene_create(C.byref(d), 0, C.byref(h)) == _capi.RT_OK, lib.rt_last_error()

    class Raw:
        _s = h
    try:
        k = kat["kat_textures"]
        it = items(len(k))
        it[:, 0], it[:, 1], it[:, 2] = 0, k[:, 0], k[:, 1]
        out = api.debug_eval(Raw, _capi.RT_EVAL_TEXTURE, it)
        assert np.array_equal(bits(out[:, 0:3]), bits(k[:, 2:5]))  # image lookup: bit-equal
        assert np.allclose(out[:, 3:6], k[:, 2:5], rtol=1e-6)
        it = items(len(k))
        it[:, 0], it[:, 3:6] = 3, k[:, 5:8]
        out = api.debug_eval(Raw, _capi.RT_EVAL_TEXTURE, it)
        assert (out[:, 0] == k[:, 8]).mean() >= 0.995 and (out[:, 3] == k[:, 8]).mean() >= 0.99  # checker: only the sign of a sine product
        p = kat["kat_perlin"]
        it = items(len(p))
        it[:, 0], it[:, 3:6] = np.where(p[:, 5] < 1.0, 4, 5), p[:, 0:3]
        out = api.debug_eval(Raw, _capi.RT_EVAL_TEXTURE, it)
        assert np.allclose(out[:, 0], p[:, 6], rtol=0, atol=4e-6)  # noise: 0.5 (1 + sinf(...)), the device's sinf
        assert np.allclose(out[:, 3], p[:, 6], rtol=0, atol=2e-4)  # hardware sine of an argument of up to a few hundred radians
        # the wave-level form the per-lane render kernel uses for a few lanes (56 lanes, one term each): the same terms in another order
        assert np.allclose(out[:, 6], out[:, 3], rtol=0, atol=2e-5) and np.allclose(out[:, 6], p[:, 6], rtol=0, atol=2e-4)
    finally:
        lib.rt_scene_destroy(h)


@pytest.mark.parametrize("scene", ["cornell_box", "simple_scene"])
def test_samplers_on_the_device_equal_the_oracle_on_the_same_stream(scene):
    """hitable_list::random over the important list (a rect and a sphere in the Cornell box, a sphere lamp in simple_scene),
    random_cosine_direction (with its factor 2), the rejection samplers: device against oracle, each from draw 0 of the same
    counter stream."""
    hs = HostScene(scene, 1.0)
    sc = Scene(hs, 0)
    rng = np.random.default_rng(5)
    n = 2000
    it = items(n)
    keys = rng.integers(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32)
    pixels = rng.integers(0, 2**22, size=n, dtype=np.uint64).astype(np.uint32)
    it[:, 0], it[:, 1] = keys.view(np.float32), pixels.view(np.float32)
    it[:, 2:5] = rng.uniform(1, 554, size=(n, 3)) if scene == "cornell_box" else rng.uniform(-4, 4, size=(n, 3))
    out = api.debug_eval(sc, _capi.RT_EVAL_SAMPLERS, it)
    want = np.stack([O.kat_samplers(hs, keys[i], pixels[i], it[i, 2:5]) for i in range(n)])
    scale = np.nanmax(np.abs(want[:, 0:3]))  # (a point inside a sphere light has no cone towards it: NaN on both sides)
    assert np.array_equal(np.isnan(out[:, 0:3]), np.isnan(want[:, 0:3]))
    assert np.allclose(out[:, 0:3], want[:, 0:3], rtol=1e-5, atol=1e-6 * scale, equal_nan=True)  # rect: bit-equal sums; sphere: sincosf
    assert np.allclose(out[:, 3:6], want[:, 3:6], rtol=1e-5, atol=2e-6)
    assert np.array_equal(bits(out[:, 6:11]), bits(want[:, 6:11]))  # rejection samplers: draws and products only
    assert np.allclose(out[:, 11:14], want[:, 6:9], atol=1e-6) and np.allclose(out[:, 14:16], want[:, 9:11], atol=1e-6)
    sc.close()


def _bounce_items(rng, n, lo, hi):
    it = items(n)
    it[:, 0:3] = rng.uniform(lo, hi, size=(n, 3))
    it[:, 3:6] = rng.normal(size=(n, 3))
    it[:, 6] = rng.uniform(size=n)
    keys = rng.integers(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32)
    pixels = rng.integers(0, 2**22, size=n, dtype=np.uint64).astype(np.uint32)
    it[:, 7], it[:, 8] = keys.view(np.float32), pixels.view(np.float32)
    return it, keys, pixels


@pytest.mark.parametrize("scene,lo,hi", [("cornell_box", 1.0, 554.0), ("random_spheres", -6.0, 6.0), ("cornell_box_volumes", 1.0, 554.0),
                                         ("simple_scene", -3.0, 3.0)])
def test_one_bounce_of_color_on_the_device_equals_the_oracle(scene, lo, hi):
    """world->hit() + ONE bounce of color() per material: lambertian (mixture pdf over the important list), metal (with and
    without fuzz), dielectric, diffuse_light, isotropic -- both device families against the oracle on the same stream.
    Exact family: the material met and whether the path goes on are equal; direction and weight within 1e-5 (metal, dielectric
    and emission bit-equal: no libm in them).  Fast family: the same hits (>= 99.5 % of the rays: ties and edges), direction
    and weight within 1e-3 where both scatter the same way, t within 1e-3 (the bar of the hit-by-hit tests: a 1000-unit ground
    sphere met from a few units away loses digits in its discriminant)."""
    hs = HostScene(scene, 1.0)
    sc = Scene(hs, 0)
    rng = np.random.default_rng(17)
    n = 3000
    it, keys, pixels = _bounce_items(rng, n, lo, hi)
    want = np.stack([O.kat_bounce(hs, it[i, 0:7], keys[i], pixels[i]) for i in range(n)])
    exact = api.debug_eval(sc, _capi.RT_EVAL_BOUNCE, it)[:, :10]
    fast = api.debug_eval(sc, _capi.RT_EVAL_BOUNCE_FAST, it)[:, :10]
    kinds = set(int(k) for k in want[want[:, 0] > 0, 1])
    assert len(kinds) >= 3, kinds
    media = want[:, 1] == _capi.RT_MAT_ISOTROPIC
    # the kernels end a path whose weight is zero or NaN (every later factor multiplies it; de_nan, src/vec3.h:184-191): there the
    # oracle reports a scattered ray of no weight, the device none
    w = want[:, 6:9]
    dead = (want[:, 2] > 0) & (~np.isfinite(w).all(axis=1) | (w == 0).all(axis=1))
    want = want.copy()
    want[dead, 2:9] = 0
    # exact family
    assert np.array_equal(exact[:, 0:3], want[:, 0:3])
    solid = ~media
    assert np.array_equal(bits(exact[solid, 9]), bits(want[solid, 9]))  # t of a solid hit (a medium's t calls logf)
    nolibm = np.isin(want[:, 1], [_capi.RT_MAT_METAL, _capi.RT_MAT_DIELECTRIC, _capi.RT_MAT_DIFFUSE_LIGHT]) & (want[:, 0] > 0)
    if scene == "random_spheres":  # a metal with fuzz draws from the unit sphere: draws and products only, still no libm
        assert nolibm.sum() > 150
    assert nolibm.sum() > 10 and same_bits(exact[nolibm, 3:9], want[nolibm, 3:9])
    assert np.allclose(exact[:, 3:9], want[:, 3:9], rtol=1e-5, atol=1e-6)
    # fast family
    fast = fast.copy()
    same = np.all(fast[:, 0:3] == want[:, 0:3], axis=1)
    assert same.mean() >= 0.995, same.mean()
    ok = same & ~media  # inside a medium the free-flight distance uses the hardware log: compared statistically elsewhere
    assert np.allclose(fast[ok, 3:6], want[ok, 3:6], rtol=1e-3, atol=1e-4)
    # (the weight: a checker's square is the sign of a product of hardware sines, which far from the origin may fall on the
    # other side of a boundary: simple_scene, 1 ray of 3000)
    assert np.isclose(fast[ok, 6:9], want[ok, 6:9], rtol=2e-3, atol=1e-5).all(axis=1).mean() >= 0.998
    assert np.isclose(fast[ok, 9], want[ok, 9], rtol=1e-3).mean() >= 0.995 and np.allclose(fast[ok, 9], want[ok, 9], rtol=2e-2)  # grazing sphere hits
    sc.close()


def test_debug_eval_rejects_bad_arguments(cornell):
    _, sc = cornell
    with pytest.raises(api.RtError, match="unknown kind"):
        api.debug_eval(sc, 99, items(1))
    assert len(api.debug_eval(sc, _capi.RT_EVAL_OPTICS, items(0))) == 0
