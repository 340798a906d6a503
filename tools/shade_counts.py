#!/usr/bin/env python3
"""Who shades in render_fast_block_kernel: lanes per wave-round by class (needs a library built with -DRT_COUNT_SHADE:
tools/build_variant.sh shadecount -DRT_COUNT_SHADE, then RT_AMD_LIB=ray_tracing_amd/build_var/librt_shadecount.so).
usage: shade_counts.py [scene:nx:ny:spp]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ray_tracing_amd import _capi, api
lib = _capi.load()
lib.rt_debug_shade_counts.argtypes = [C.c_void_p, C.c_int]
name, nx, ny, spp = (sys.argv[1] if len(sys.argv) > 1 else "cornell_box:1024:1024:64").split(":"); nx, ny, spp = int(nx), int(ny), int(spp)
sc = api.Scene(api.HostScene(name, nx / ny), 0)
acc = torch.empty((ny, nx, 4), dtype=torch.float32, device="cuda:0")
lib.rt_debug_shade_counts(None, 1)
api.render(sc, acc, nx, ny, 0, spp)
torch.cuda.synchronize()
out = (C.c_ulonglong * 16)()
lib.rt_debug_shade_counts(out, 1)
c = [int(x) for x in out]
paths = nx * ny * spp
print(f"{name} {nx}x{ny}x{spp}: {paths} paths, {c[0]} wave-rounds, {c[1]} of them scan ({c[1] / c[0]:.3f}), lanes alive per scanning wave {c[2] / max(c[1], 1):.1f}")
print(f"  rays per path {c[2] / paths:.3f}")
for label, w, l in (("scatter_specular", 3, 4), ("scatter_lambert", 5, 6), ("  rect-light branch", 7, 8), ("  axis-sampling branch", 9, 10)):
    print(f"  {label:24s} wave-rounds {c[w]:12d} ({c[w] / c[0]:.3f} of all)  lanes {c[l]:14d}  lanes per executing wave {c[l] / max(c[w], 1):5.1f}")
