#!/usr/bin/env python3
"""Where does the PSNR against the reference's converged image saturate?  (run on the GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from ray_tracing_amd import api, _capi
G = os.path.join(ROOT, "tests", "golden")
ref4 = np.load(os.path.join(G, "cornell_box_256x256_4096spp_r50.npy"))
ref8 = np.load(os.path.join(G, "cornell_box_256x256_8192spp_r50.npy"))
refB = 2 * ref8 - ref4  # the second 4096 samples of the reference run
def to8(x): return np.floor(np.clip(np.sqrt(np.maximum(x, 0)), 0, 1) * 255.99)
def psnr(a, b): return 10 * np.log10(255.0 ** 2 / np.mean((to8(a) - to8(b)) ** 2))
def lin_rmse(a, b): return float(np.sqrt(np.mean((a - b) ** 2)))
sc = api.Scene(api.HostScene("cornell_box", 1.0), 0)
def gpu(spp, seed, path, s0=0):
    sc.set_path(render=path)
    acc = torch.empty((256, 256, 4), dtype=torch.float32, device="cuda:0")
    api.render(sc, acc, 256, 256, s0, s0 + spp, seed=seed); torch.cuda.synchronize()
    return acc.cpu().numpy()[..., 1:][..., ::-1] / spp
F, E = _capi.RT_PATH_FAST, _capi.RT_PATH_EXACT
f4a, f4b = gpu(4096, 1, F), gpu(4096, 2, F)
f64, e64 = gpu(65536, 3, F), gpu(65536, 4, E)
f64b = gpu(65536, 5, F)
print("ref first 4096 vs ref second 4096   :", round(psnr(ref4, refB), 2), "dB  lin rmse", lin_rmse(ref4, refB))
print("gpu 4096 (seed 1) vs gpu 4096 (seed 2):", round(psnr(f4a, f4b), 2), "dB  lin rmse", lin_rmse(f4a, f4b))
print("gpu fast 65536 vs gpu exact 65536    :", round(psnr(f64, e64), 2), "dB  lin rmse", lin_rmse(f64, e64))
print("gpu fast 65536 vs gpu fast 65536'    :", round(psnr(f64, f64b), 2), "dB  lin rmse", lin_rmse(f64, f64b))
print("gpu fast 65536 vs ref 8192           :", round(psnr(f64, ref8), 2), "dB  lin rmse", lin_rmse(f64, ref8))
print("gpu exact 65536 vs ref 8192          :", round(psnr(e64, ref8), 2), "dB  lin rmse", lin_rmse(e64, ref8))
print("gpu fast 65536 vs ref 4096           :", round(psnr(f64, ref4), 2), "dB")
print("gpu fast 65536 vs ref second 4096    :", round(psnr(f64, refB), 2), "dB")
# where is the difference?  8x8 blocks of 32x32 pixels, signed mean difference / its standard error estimate
d = (f64 - ref8).mean(axis=2)
noise = ((ref4 - refB).mean(axis=2)) / 2.0  # per-pixel noise sample of the 8192-spp reference
blocks = d.reshape(8, 32, 8, 32).mean(axis=(1, 3)); nb = noise.reshape(8, 32, 8, 32).std(axis=(1, 3)) / 32.0
np.set_printoptions(precision=1, suppress=True, linewidth=200)
print("block mean difference in units of its standard error:\n", blocks / nb)
print("pixels > 0.95 linear in ref8:", (ref8.max(axis=2) > 0.95).mean(), " mse share of them:",
      ((to8(f64) - to8(ref8)) ** 2)[ref8.max(axis=2) > 0.95].sum() / ((to8(f64) - to8(ref8)) ** 2).sum())
