#!/usr/bin/env python3
"""Queue kernel (RT_OPT_BLOCK_KERNEL = 2) against the block kernel: same paths, so the same image up to summation order; then timings."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from ray_tracing_amd import api, _capi
def render(sc, nx, ny, s0, s1, **kw):
    acc = torch.empty((ny, nx, 4), dtype=torch.float32, device="cuda:0")
    api.render(sc, acc, nx, ny, s0, s1, **kw); torch.cuda.synchronize()
    return acc.cpu().numpy()
sizes = [(16, 16, 3, 1), (80, 48, 8, 2), (100, 60, 40, 1), (33, 17, 5, 1)]
for nx, ny, ns, nss in sizes:
    sc = api.Scene(api.HostScene("cornell_box", nx / ny), 0)
    sc.set_option(_capi.RT_OPT_BLOCK_KERNEL, 1)
    a = render(sc, nx, ny, 0, ns, sub_samples=nss, seed=6)
    sc.set_option(_capi.RT_OPT_BLOCK_KERNEL, 0)
    b = render(sc, nx, ny, 0, ns, sub_samples=nss, seed=6)
    ok_a = np.all(b[..., 0] == ns)
    close = np.isclose(a[..., 1:], b[..., 1:], rtol=1e-4, atol=1e-5).all(axis=-1).mean()
    print(f"{nx}x{ny}x{ns}x{nss}: A ok {ok_a}, {close:.4f} of pixels equal, means {a[...,1:].mean():.5f} {b[...,1:].mean():.5f}", flush=True)
    sc.close()
if len(sys.argv) > 1:
    spp = int(sys.argv[1])
    sc = api.Scene(api.HostScene("cornell_box", 1.0), 0)
    acc = torch.empty((1024, 1024, 4), dtype=torch.float32, device="cuda:0")
    for mode in (1, 0, 1, 0):
        sc.set_option(_capi.RT_OPT_BLOCK_KERNEL, mode)
        for _ in range(2):
            api.render(sc, acc, 1024, 1024, 0, spp)
        torch.cuda.synchronize()
        ms = sc.last_render_ms()
        print(f"mode {mode}: {ms:.3f} ms  {1024*1024*spp/ms/1e3:.1f} Mpaths/s", flush=True)
