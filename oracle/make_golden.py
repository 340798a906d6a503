#!/usr/bin/env python3
"""Writes tests/golden/ from the REAL reference (oracle/_ref/ref_harness).

Runs in the build container only (needs /root/reference to build the harness:
`make -C oracle ref`).  Every number in tests/golden/ is an output of the
reference's own code; see oracle/ref_harness_tail.cpp for what each mode calls.
The fixtures are inputs and expected outputs (data), not source.

    python oracle/make_golden.py            # everything except the slow converged image
    python oracle/make_golden.py --scenes cornell_box --image   # + Cornell 256x256 at 1024/4096/8192 spp (about 25 minutes)
    python oracle/make_golden.py --converged     # converged images WITH second moments of the scenes of BASELINE configs 1, 3, 5 and the
        # smoke-box Cornell variant (CONVERGED below; 4 processes, about 12 minutes), each inside the period of minstd_rand0:
        # the harness counts the draws exactly and this script refuses a fixture that used more than 80 % of the period
    python oracle/make_golden.py --scenes cornell_box --image96 --only-images   # Cornell 96x96 at 4096/8192 spp (3 minutes):
        # the converged image that stays INSIDE the period of the reference's minstd_rand0 (2^31-2 draws; a path takes
        # about 25, so 96*96*8192 paths draw 1.9e9), which a 256x256 render leaves after about 1300 spp
"""
import argparse
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "..", "tests", "golden")
HARNESS = os.path.join(HERE, "_ref", "ref_harness")


def run(*args):
    cmd = [HARNESS] + [str(a) for a in args]
    print("+", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)


# scene -> (nx, ny, spp): what `--converged` renders with the reference's own compute_one_sample_task / color().
# spp is as large as the generator's period allows (draws per sample image measured by the harness: random_spheres 70k,
# next_week_final 193k, cornell_box_volumes 89k, sphere_field_4k 67k), so the standard errors of the fixture are small.
CONVERGED = {
    "random_spheres": (96, 64, 16384),
    "next_week_final": (64, 64, 8192),
    "cornell_box_volumes": (64, 64, 16384),
    "sphere_field_4k": (64, 64, 8192),
}


def converged():
    import re
    with tempfile.TemporaryDirectory() as tmp:
        procs = {}
        for scene, (nx, ny, spp) in CONVERGED.items():
            cmd = [HARNESS, "imagesq", scene, str(nx), str(ny), str(spp), "50", f"{tmp}/{scene}.npy"]
            print("+", " ".join(cmd), flush=True)
            procs[scene] = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        for scene, p in procs.items():
            out = p.communicate()[0]
            if p.returncode:
                raise SystemExit(f"{scene}: harness failed")
            draws, period = map(int, re.search(r"draws (\d+) period (\d+)", out).groups())
            nx, ny, spp = CONVERGED[scene]
            print(f"{scene}: {draws} draws = {draws / period:.3f} of the period", flush=True)
            if draws > 0.8 * period:
                raise SystemExit(f"{scene}: leaves the generator's period; lower spp")
            a = np.load(f"{tmp}/{scene}.npy")
            np.savez_compressed(os.path.join(GOLD, f"{scene}_{nx}x{ny}_{spp}spp_r50_moments.npz"), moments=a,
                                draws=np.int64(draws), spp=np.int32(spp))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--converged", action="store_true", help="only the converged images with second moments (CONVERGED)")
    ap.add_argument("--image", action="store_true", help="also render the 4096-spp converged image")
    ap.add_argument("--image96", action="store_true", help="also render the within-period 96x96 converged images")
    ap.add_argument("--only-images", action="store_true", help="leave the .npz fixtures and kat.npz as they are")
    ap.add_argument("--scenes", default="cornell_box,cornell_box_volumes,random_spheres,next_week_final")
    a = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    if a.converged:
        return converged()
    with tempfile.TemporaryDirectory() as tmp:
        for scene in a.scenes.split(","):
            out = {}
            if scene == "cornell_box":
                # per-sample float buffers of a `-t 1` run: (ns, ny, nx, 4) = [1,b,g,r]
                run("samples", scene, 64, 64, 4, 1, 50, f"{tmp}/s.npy")
                out["samples_64x64_s4_S1_r50"] = np.load(f"{tmp}/s.npy")
                run("samples", scene, 48, 32, 2, 2, 5, f"{tmp}/s.npy")
                out["samples_48x32_s2_S2_r5"] = np.load(f"{tmp}/s.npy")
                # primary rays through pixel centres: see ref_harness_tail.cpp mode_hits for the 21 columns
                run("hits", scene, 128, 128, f"{tmp}/h.npy")
                out["hits_128x128"] = np.load(f"{tmp}/h.npy")
                run("hits", scene, 200, 120, f"{tmp}/h.npy")
                out["hits_200x120"] = np.load(f"{tmp}/h.npy")
                run("tree", scene, os.path.join(GOLD, f"{scene}_tree.txt"))
            else:
                # the scenes of ray_tracing_amd/host/rt_scenes_ext.h, compiled against the reference's headers
                run("samples", scene, 32, 24, 2, 1, 50, f"{tmp}/s.npy")
                out["samples_32x24_s2_S1_r50"] = np.load(f"{tmp}/s.npy")
                run("hits", scene, 64, 48, f"{tmp}/h.npy")
                out["hits_64x48"] = np.load(f"{tmp}/h.npy")
            if a.only_images:
                out = None
            if out is not None:
                np.savez_compressed(os.path.join(GOLD, f"{scene}.npz"), **out)
            if a.image96:
                for spp in (4096, 8192):
                    run("image", scene, 96, 96, spp, 50, f"{tmp}/i.npy")
                    np.save(os.path.join(GOLD, f"{scene}_96x96_{spp}spp_r50.npy"), np.load(f"{tmp}/i.npy"))
            if a.image:
                for spp in (1024, 4096, 8192):  # 8192 also shows where the reference's minstd stream starts to replay
                    run("image", scene, 256, 256, spp, 50, f"{tmp}/i.npy")
                    np.save(os.path.join(GOLD, f"{scene}_256x256_{spp}spp_r50.npy"), np.load(f"{tmp}/i.npy"))
        if not a.only_images:
            os.makedirs(f"{tmp}/kat")
            run("kat", f"{tmp}/kat")
            kat = {f[:-4]: np.load(f"{tmp}/kat/{f}") for f in sorted(os.listdir(f"{tmp}/kat"))}
            np.savez_compressed(os.path.join(GOLD, "kat.npz"), **kat)
    print("wrote", sorted(os.listdir(GOLD)))


if __name__ == "__main__":
    sys.exit(main())
