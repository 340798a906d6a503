"""ray_tracing_amd -- MI355X-native path-tracing core behind droune2001/ray_tracing's scene API.

The product is C++/HIP (ray_tracing_amd/host, ray_tracing_amd/csrc, include/rt_amd.h).
This Python package is the thin harness around the C-ABI that tests, bench.py and the
multi-GPU driver use: device buffers and RCCL come from torch, everything else is the
shared library.
"""
from . import _capi
from .api import Scene, RtError, render, trace_rays, camera_rays, resolve, scene_names  # noqa: F401
