"""Python harness over the C-ABI (include/rt_amd.h, include/rt_host.h).

Nothing here computes anything: HostScene asks the C++ host layer for a built-in
scene (scene function + bvh_node build + flatten), Scene uploads it with
rt_scene_create, and render()/trace_rays()/... pass pointers through.  torch is
used only to own device memory and streams.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import rt_hit, rt_render_params


class RtError(RuntimeError):
    pass


def _check(status, what):
    if status != _capi.RT_OK:
        msg = _capi.load().rt_last_error().decode(errors="replace")
        raise RtError(f"{what} failed with status {status}: {msg}")


def scene_names():
    return _capi.load().rt_host_scene_names().decode().split(",")


class HostScene:
    """A flattened built-in scene, host memory only (no GPU needed)."""

    def __init__(self, name="cornell_box", aspect=1.0):
        lib = _capi.load()
        self._h = C.c_void_p()
        _check(lib.rt_host_scene_build(name.encode(), C.c_float(aspect), C.byref(self._h)), "rt_host_scene_build")
        self.name, self.aspect = name, aspect
        self.desc = lib.rt_host_scene_desc(self._h)  # POINTER(rt_scene_desc)
        self.ran01_state = int(lib.rt_host_ran01_state())

    def __del__(self):
        if getattr(self, "_h", None) and _capi is not None and getattr(_capi, "_lib", None) is not None:
            _capi._lib.rt_host_scene_free(self._h)  # (at interpreter exit the module may already be gone)
            self._h = None


class Scene:
    """Device-resident scene (rt_scene_create)."""

    def __init__(self, host_scene, device=0, create_flags=0):
        lib = _capi.load()
        self.host = host_scene if isinstance(host_scene, HostScene) else HostScene(host_scene)
        self.device = device
        self._s = C.c_void_p()
        _check(lib.rt_scene_create_flags(self.host.desc, device, create_flags, C.byref(self._s)), "rt_scene_create_flags")

    def set_option(self, option, value):
        """rt_scene_set_option for the tuning / test switches (RT_OPT_BLOCK_KERNEL, RT_OPT_CHAIN_TASKS, ...)."""
        _check(_capi.load().rt_scene_set_option(self._s, option, value), "rt_scene_set_option")

    def set_path(self, render=None, trace=None):
        """rt_scene_set_option: choose RT_PATH_AUTO / RT_PATH_EXACT / RT_PATH_FAST for rt_render / rt_trace_rays."""
        lib = _capi.load()
        if render is not None:
            _check(lib.rt_scene_set_option(self._s, _capi.RT_OPT_RENDER_PATH, render), "rt_scene_set_option")
        if trace is not None:
            _check(lib.rt_scene_set_option(self._s, _capi.RT_OPT_TRACE_PATH, trace), "rt_scene_set_option")

    def info(self, what):
        v = C.c_int()
        _check(_capi.load().rt_scene_get_info(self._s, what, C.byref(v)), "rt_scene_get_info")
        return v.value

    def last_render_ms(self):
        ms = C.c_float()
        _check(_capi.load().rt_last_render_ms(self._s, C.byref(ms)), "rt_last_render_ms")
        return ms.value

    def close(self):
        if getattr(self, "_s", None) and _capi is not None and getattr(_capi, "_lib", None) is not None:
            _capi._lib.rt_scene_destroy(self._s)
            self._s = None

    __del__ = close


def make_params(nx, ny, sample_begin, sample_end, sub_samples=1, max_depth=50, seed=0, accumulate=False):
    return rt_render_params(C.sizeof(rt_render_params), nx, ny, sample_begin, sample_end, sub_samples, max_depth,
                            seed, _capi.RT_RENDER_ACCUMULATE if accumulate else 0)


def render(scene, accum, nx, ny, sample_begin, sample_end, sub_samples=1, max_depth=50, seed=0, accumulate=False,
           stream=None):
    """rt_render into `accum`, a contiguous float32 CUDA tensor of shape (ny, nx, 4).  Asynchronous."""
    import torch
    assert accum.is_cuda and accum.dtype == torch.float32 and accum.is_contiguous() and accum.numel() == nx * ny * 4
    p = make_params(nx, ny, sample_begin, sample_end, sub_samples, max_depth, seed, accumulate)
    s = torch.cuda.current_stream(accum.device).cuda_stream if stream is None else stream
    _check(_capi.load().rt_render(scene._s, C.byref(p), C.c_void_p(accum.data_ptr()), C.c_void_p(s)), "rt_render")
    return accum


def resolve(accum, out=None, stream=None):
    """rt_resolve: (ny, nx, 4) float sums -> (ny, nx) packed 0xAABBGGRR (int32 tensor holding the uint32 bits)."""
    import torch
    ny, nx = accum.shape[0], accum.shape[1]
    if out is None:
        out = torch.empty((ny, nx), dtype=torch.int32, device=accum.device)
    s = torch.cuda.current_stream(accum.device).cuda_stream if stream is None else stream
    _check(_capi.load().rt_resolve(C.c_void_p(accum.data_ptr()), nx, ny, C.c_void_p(out.data_ptr()), C.c_void_p(s)),
           "rt_resolve")
    return out


def render_to_host(scene, nx, ny, sample_begin, sample_end, sub_samples=1, max_depth=50, seed=0, want_rgba8=True):
    accum = np.zeros((ny, nx, 4), dtype=np.float32)
    rgba = np.zeros((ny, nx), dtype=np.uint32) if want_rgba8 else None
    p = make_params(nx, ny, sample_begin, sample_end, sub_samples, max_depth, seed)
    _check(_capi.load().rt_render_to_host(scene._s, C.byref(p), accum.ctypes.data_as(C.c_void_p),
                                          rgba.ctypes.data_as(C.c_void_p) if want_rgba8 else None), "rt_render_to_host")
    return accum, rgba


def camera_rays(scene, st, lens=None, time01=None):
    st = np.ascontiguousarray(st, dtype=np.float32).reshape(-1, 2)
    n = st.shape[0]
    rays = np.zeros((n, 7), dtype=np.float32)
    lens_p = None if lens is None else np.ascontiguousarray(lens, dtype=np.float32).ctypes.data_as(C.c_void_p)
    time_p = None if time01 is None else np.ascontiguousarray(time01, dtype=np.float32).ctypes.data_as(C.c_void_p)
    _check(_capi.load().rt_camera_rays(scene._s, n, st.ctypes.data_as(C.c_void_p), lens_p, time_p,
                                       rays.ctypes.data_as(C.c_void_p)), "rt_camera_rays")
    return rays


HIT_DTYPE = np.dtype([("prim_id", "<i4"), ("material", "<i4"), ("t", "<f4"), ("u", "<f4"), ("v", "<f4"),
                      ("p", "<f4", 3), ("normal", "<f4", 3), ("reserved", "<f4")])
assert HIT_DTYPE.itemsize == C.sizeof(rt_hit)


def trace_rays(scene, rays, t_min=0.001, t_max=3.4028234663852886e38, seed=0, noalloc=False):
    """rt_trace_rays, or rt_trace_rays_noalloc (static device staging, at most RT_TRACE_NOALLOC_MAX_RAYS rays)."""
    rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 7)
    hits = np.zeros(rays.shape[0], dtype=HIT_DTYPE)
    lib = _capi.load()
    fn, name = (lib.rt_trace_rays_noalloc, "rt_trace_rays_noalloc") if noalloc else (lib.rt_trace_rays, "rt_trace_rays")
    _check(fn(scene._s, rays.shape[0], rays.ctypes.data_as(C.c_void_p), C.c_float(t_min), C.c_float(t_max), seed,
              hits.ctypes.data_as(C.c_void_p)), name)
    return hits


def debug_eval(scene, kind, items):
    """rt_debug_eval: `items` is n x 16 float32 (unused columns 0); returns n x 16 float32."""
    items = np.ascontiguousarray(items, dtype=np.float32).reshape(-1, 16)
    out = np.zeros_like(items)
    _check(_capi.load().rt_debug_eval(scene._s, kind, items.shape[0], items.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)),
           "rt_debug_eval")
    return out
