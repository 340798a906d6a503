#!/usr/bin/env python3
"""bench.py -- headline benchmark: Cornell box 1024x1024, 1024 spp, depth 50 (BASELINE.json configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" is one complete image: every rank renders its disjoint range of the 1024 samples into its own
float4 accumulator (rt_render), the accumulators are summed onto rank 0 with one RCCL reduce (N > 1), and
rank 0 resolves to 8 bits (rt_resolve).  The scene is uploaded once before the timed region (inputs
resident in HBM).  The total work is fixed, so the scaling is strong.  Prints ONE JSON line on rank 0.

value       = nx*ny*spp / seconds per step / 1e6   [Mpaths/s, whole job]
roofline    = the render kernel against the FP32 vector peak (MFMA is unused on purpose: no dense
              contraction on this path): algorithmic FLOP per launch (1.40 kFLOP per path x
              paths per launch) / the kernel's duration from HIP events on the launch stream
cpu_baseline= the compiled reference (oracle/_ref/rt_ref, kind "reference") or, if that binary is not in
              the tree, the oracle port, timed on this host on a bounded sample of the same workload
"""
import argparse
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

NX = NY = 1024
SPP = 1024
DEPTH = 50
FLOP_PER_PATH = 1.40e3  # DESIGN.md "Algorithmic work": canonical FLOP per Cornell path
FP32_VECTOR_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters


def cpu_baseline():
    """Bounded CPU run of the same workload at reduced size: Cornell 256x256, depth 50."""
    ref = os.path.join(ROOT, "oracle", "_ref", "rt_ref")
    cores = min(16, os.cpu_count() or 1)
    if os.path.exists(ref):
        runs = []
        for threads, spp in ((1, 128), (cores, 512)):  # about 18 s of CPU work on the GPU box
            out = subprocess.run([ref, "-w", "256", "-h", "256", "-s", str(spp), "-r", str(DEPTH), "-t", str(threads),
                                  "-o", "/tmp/rt_ref_bench.png"], capture_output=True, text=True, cwd="/tmp", timeout=600)
            m = re.search(r"Time:\s*([0-9.]+)\s*ms", out.stdout)
            if not m:
                raise RuntimeError("reference binary printed no time: " + out.stdout[-200:] + out.stderr[-200:])
            runs.append((threads, 256 * 256 * spp / (float(m.group(1)) * 1e-3) / 1e6, spp))
        best = max(runs, key=lambda r: r[1])
        return {"value": round(best[1], 4), "unit": "Mpaths/s", "cores": best[0], "kind": "reference",
                "sample": "oracle/_ref/rt_ref (the reference's main.cpp, g++ -O2) Cornell 256x256 depth 50, its own "
                          "'Time:' window; " + "; ".join(f"-t {t} -s {s}: {v:.3f} Mpaths/s" for t, v, s in runs)}
    import oracle_lib as O
    from ray_tracing_amd.api import HostScene
    hs = HostScene("cornell_box", 1.0)
    t0 = time.perf_counter()
    O.render(hs, 256, 256, 0, 64, 1, DEPTH, O.ORC_RNG_COUNTER, 0, cores)
    dt = time.perf_counter() - t0
    return {"value": round(256 * 256 * 64 / dt / 1e6, 4), "unit": "Mpaths/s", "cores": cores, "kind": "port",
            "sample": f"oracle/librt_oracle.so Cornell 256x256 64 spp depth 50, {cores} threads, per-path RNG"}


def traffic_from_profile():
    """HBM bytes per launch of the render kernel on the headline workload, from the committed rocprofv3 PMC passes
    (FETCH_SIZE doubled as the gfx950 guide prescribes + WRITE_SIZE, both reported in KiB); None if not profiled."""
    path = os.path.join(ROOT, "profiles", "headline_traffic.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f).get("hbm_bytes_per_launch")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=SPP, help=argparse.SUPPRESS)  # for quick experiments only
    ap.add_argument("--size", type=int, default=NX, help=argparse.SUPPRESS)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    from ray_tracing_amd import api, distributed as D

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    nx = ny = a.size
    spp = a.spp
    s0, s1 = D.sample_range(rank, world, spp)  # disjoint sample ranges
    scene = api.Scene(api.HostScene("cornell_box", nx / ny), local)
    accum = torch.empty((ny, nx, 4), dtype=torch.float32, device=dev)
    image = torch.empty((ny, nx), dtype=torch.int32, device=dev)

    def step():
        api.render(scene, accum, nx, ny, s0, s1, max_depth=DEPTH, seed=0)
        D.reduce_accumulators(accum, 0)
        if rank == 0:
            api.resolve(accum, image)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    # kernel duration from the HIP events rt_render records on its launch stream (last step; steps are identical)
    k_ms = scene.last_render_ms()
    t = torch.tensor([dt, k_ms], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt, k_ms = float(t[0]), float(t[1])

    if rank == 0:
        from ray_tracing_amd import _capi
        # the Cornell box fits the record table and has no textures: rt_render runs it on render_fast_block_kernel
        kernel = ("render_fast_block_kernel" if scene.info(_capi.RT_INFO_LAST_RENDER_PATH) == _capi.RT_PATH_FAST else "render_kernel_uniform")
        sec_per_step = dt / a.steps
        paths = nx * ny * spp
        value = paths / sec_per_step / 1e6
        paths_per_launch = nx * ny * (s1 - s0)
        achieved = paths_per_launch * FLOP_PER_PATH / (k_ms * 1e-3) / 1e12
        out = {
            "metric": "Mpaths/sec (node), Cornell box 1024x1024 1024spp depth 50",
            "value": round(value, 2), "unit": "Mpaths/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(sec_per_step * 1e3, 3), "seconds_to_image": round(sec_per_step, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"cornell_box {nx}x{ny} {spp}spp depth {DEPTH} (BASELINE.json configs[1])",
                       "partition": f"disjoint spp ranges over {world} GPU(s) + one RCCL sum-reduce of float4[{nx}*{ny}]"},
            "roofline": {"bound": "valu_fp32", "achieved": round(achieved, 4), "peak": FP32_VECTOR_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / FP32_VECTOR_PEAK_TFLOPS, 5), "traffic": traffic_from_profile(),
                         "kernel": kernel, "kernel_ms": round(k_ms, 3),
                         "note": "no dense contraction on this path: MFMA unused, FP32 vector peak = FP32 matrix peak"},
        }
        if world == 1 and not a.no_cpu:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
