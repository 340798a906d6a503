#!/usr/bin/env python3
"""bench.py -- headline benchmark: Cornell box 1024x1024, 1024 spp, depth 50 (BASELINE.json configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c1|c3|c5]

With --gpus N > 1 and no WORLD_SIZE in the environment this process only LAUNCHES: it starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>` as a child (before
torch or HIP is touched here; never an exec), relays the child's output and returns its exit code.  Started by
torch.distributed.run itself (WORLD_SIZE set) it is one rank.

One "step" is one complete image: every rank renders its disjoint range of the samples into its own float4
accumulator (rt_render), the accumulators are summed onto rank 0 with one RCCL reduce (N > 1 only), and rank 0
resolves to 8 bits (rt_resolve).  The scene is uploaded once before the timed region (inputs resident in HBM).
The total work is fixed, so the scaling is strong.  Prints ONE JSON line on rank 0.

value       = nx*ny*spp / seconds per step / 1e6   [Mpaths/s, whole job]
roofline    = the render kernel against its bound.  Cornell box, random spheres, Next-Week final: FP32 vector peak
              (MFMA is unused on purpose: no dense contraction on this path): algorithmic FLOP per launch (DESIGN.md:
              kFLOP per path x paths per launch) / the kernel's duration from HIP events on the launch stream.
              1M-sphere field: HBM (algorithmic tree + leaf bytes per path, DESIGN.md, x paths per launch / duration).
cpu_baseline= the compiled reference (oracle/_ref/rt_ref, kind "reference") or, if that binary is not in
              the tree, the oracle port, timed on this host on a bounded sample of the same workload
"""
import argparse
import json
import os
import re
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

DEPTH = 50
FP32_VECTOR_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
HBM_PEAK_GBS = 8000.0            # same guide: HBM3E, about 8 TB/s

# workloads = BASELINE.json configs.  flop_per_path / bytes_per_path: algorithmic work per path from the instrumented
# oracle (tools/work_counts.py; DESIGN.md section 7 lists the counts they are built from).
CONFIGS = {
    "c2": dict(scene="cornell_box", nx=1024, ny=1024, spp=1024, bound="valu_fp32", flop_per_path=1.40e3,
               name="BASELINE.json configs[1]", cpu_scene_flag=None),
    "c1": dict(scene="random_spheres", nx=1200, ny=800, spp=64, bound="valu_fp32", flop_per_path=None,
               name="BASELINE.json configs[0]", cpu_scene_flag=None),
    "c3": dict(scene="next_week_final", nx=800, ny=800, spp=4096, bound="valu_fp32", flop_per_path=None,
               name="BASELINE.json configs[2]", cpu_scene_flag=None),
    "c5": dict(scene="sphere_field_1m", nx=1920, ny=1080, spp=1024, bound="hbm", bytes_per_path=None,
               name="BASELINE.json configs[4]", cpu_scene_flag=None),
    # (quoted on 8 GPUs; on fewer it is the same image, 6 s per step on one.  No FLOP count: at 16:9 the camera sees past
    # the room's sides, a third of the paths end on their first ray, so the square image's 1.4 kFLOP per path does not hold
    # and a per-path rate is not comparable with config 2's)
    "c4": dict(scene="cornell_box", nx=3840, ny=2160, spp=16384, bound="valu_fp32", flop_per_path=None, no_counts=True,
               name="BASELINE.json configs[3]", cpu_scene_flag=None),
}


def load_work_counts():
    """profiles/work_counts.json (written by tools/work_counts.py): per scene the algorithmic FLOP and bytes per path."""
    path = os.path.join(ROOT, "profiles", "work_counts.json")
    if not os.path.exists(path):
        return {}
    with open(path) as f:
        return json.load(f)


def cpu_info():
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = os.cpu_count() or 1
    return model, os.cpu_count() or 1, usable


def cpu_baseline():
    """Bounded CPU run of the headline workload at reduced size: Cornell 256x256, depth 50, the compiled reference at
    -t 1, a thread count in between and -t = the cores this process may use (at most 16: the GPU box's share)."""
    ref = os.path.join(ROOT, "oracle", "_ref", "rt_ref")
    model, host_cpus, usable = cpu_info()
    cores = max(1, min(16, usable))
    mid = max(1, cores // 4)
    if os.path.exists(ref):
        runs = []
        plan = [(1, 64)] + ([(mid, 128)] if 1 < mid < cores else []) + ([(cores, 256)] if cores > 1 else [])
        for threads, spp in plan:  # about 15 s of CPU work on the GPU box
            out = subprocess.run([ref, "-w", "256", "-h", "256", "-s", str(spp), "-r", str(DEPTH), "-t", str(threads),
                                  "-o", "/tmp/rt_ref_bench.png"], capture_output=True, text=True, cwd="/tmp", timeout=600)
            m = re.search(r"Time:\s*([0-9.]+)\s*ms", out.stdout)
            if not m:
                raise RuntimeError("reference binary printed no time: " + out.stdout[-200:] + out.stderr[-200:])
            runs.append((threads, 256 * 256 * spp / (float(m.group(1)) * 1e-3) / 1e6, spp))
        best = max(runs, key=lambda r: r[1])
        return {"value": round(best[1], 4), "unit": "Mpaths/s", "cores": best[0], "kind": "reference",
                "cpu_model": model, "host_cpus": host_cpus, "usable_cpus": usable,
                "sample": "oracle/_ref/rt_ref (the reference's main.cpp, g++ -O2) Cornell 256x256 depth 50, its own "
                          "'Time:' window; " + "; ".join(f"-t {t} -s {s}: {v:.3f} Mpaths/s" for t, v, s in runs)
                          + "; value = the best of them (the reference's threads share one unlocked RNG: SURVEY.md section 6)"}
    import oracle_lib as O
    from ray_tracing_amd.api import HostScene
    hs = HostScene("cornell_box", 1.0)
    t0 = time.perf_counter()
    O.render(hs, 256, 256, 0, 64, 1, DEPTH, O.ORC_RNG_COUNTER, 0, cores)
    dt = time.perf_counter() - t0
    return {"value": round(256 * 256 * 64 / dt / 1e6, 4), "unit": "Mpaths/s", "cores": cores, "kind": "port",
            "cpu_model": model, "host_cpus": host_cpus, "usable_cpus": usable,
            "sample": f"oracle/librt_oracle.so Cornell 256x256 64 spp depth 50, {cores} threads, per-path RNG"}


def traffic_from_profile(cfg):
    """HBM bytes per launch of the render kernel, from the committed rocprofv3 PMC passes (FETCH_SIZE doubled as the
    gfx950 guide prescribes + WRITE_SIZE, both reported in KiB); None if that workload was not profiled."""
    path = os.path.join(ROOT, "profiles", "headline_traffic.json" if cfg == "c2" else f"{cfg}_traffic.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f)


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv):
    """The parent of an N-rank run: one child `torch.distributed.run`, output relayed, exit code returned.
    Nothing here imports torch or touches HIP, and nothing is exec'ed."""
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = os.environ.get("MASTER_PORT") or str(free_port())
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", addr,
           "--master-port", port, os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # RCCL on this pool needs dmabuf IPC
    child = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    sys.stdout.write(child.stdout)
    sys.stdout.flush()
    return child.returncode


def parse_args(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2", help="workload (default: the headline, c2)")
    ap.add_argument("--spp", type=int, default=None, help=argparse.SUPPRESS)  # for quick experiments only
    ap.add_argument("--size", type=int, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--rehearse-rccl", action="store_true", help=argparse.SUPPRESS)  # see main()
    return ap.parse_args(argv)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    a = parse_args(argv)
    if a.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(a.gpus, argv)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")

    import torch
    import torch.distributed as dist
    from ray_tracing_amd import distributed as D

    # RT_BENCH_TEST_HOOK=module:function is for the CPU tests of the launch / partition / reduce logic (tests/ supplies a
    # renderer; no GPU there).  The product has no CPU renderer: without the hook everything below runs on the GPU.
    hook = os.environ.get("RT_BENCH_TEST_HOOK")
    backend = "gloo" if hook else "nccl"
    if hook:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    # --rehearse-rccl: a one-rank process group and the reduce issued on it (started by torch.distributed.run with
    # --nproc-per-node 1): the RCCL initialisation and collective of the N > 1 path executed on a one-GPU box
    grouped = world > 1 or (a.rehearse_rccl and "RANK" in os.environ)
    if grouped:
        dist.init_process_group(backend, **({} if hook else {"device_id": dev}))
    n_ranks_seen = dist.get_world_size() if grouped else 1

    cfg = dict(CONFIGS[a.config])
    nx = ny = None
    if a.size:
        nx, ny = a.size, max(1, a.size * cfg["ny"] // cfg["nx"])
    else:
        nx, ny = cfg["nx"], cfg["ny"]
    spp = a.spp or cfg["spp"]
    s0, s1 = D.sample_range(rank, world, spp)  # disjoint sample ranges

    if hook:
        mod, fn = hook.split(":")
        render, resolve, sync, last_ms, kernel_name = getattr(__import__(mod), fn)(cfg["scene"], nx, ny, DEPTH)
    else:
        from ray_tracing_amd import _capi, api
        scene = api.Scene(api.HostScene(cfg["scene"], nx / ny), local)
        accum_buf = torch.empty((ny, nx, 4), dtype=torch.float32, device=dev)
        image = torch.empty((ny, nx), dtype=torch.int32, device=dev)

        def render(b, e):
            return api.render(scene, accum_buf, nx, ny, b, e, max_depth=DEPTH, seed=0)

        def resolve(acc):
            return api.resolve(acc, image)

        sync = torch.cuda.synchronize
        last_ms = scene.last_render_ms

        def kernel_name():
            if scene.info(_capi.RT_INFO_LAST_RENDER_PATH) != _capi.RT_PATH_FAST:
                return "render_kernel_uniform / render_kernel_lanes (exact path)"
            # record table without textures -> the block kernel; no record table -> the per-lane walk
            return "render_fast_block_kernel" if scene.info(_capi.RT_INFO_RECORDS) > 0 and cfg["scene"] == "cornell_box" else \
                ("render_fast_kernel" if scene.info(_capi.RT_INFO_RECORDS) > 0 else "render_fastlanes_kernel")

    def step():
        acc = render(s0, s1)
        D.reduce_accumulators(acc, 0, even_alone=grouped)
        if rank == 0:
            resolve(acc)

    def fence():
        sync()
        if grouped:
            dist.barrier()
        sync()

    for _ in range(a.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    # kernel duration from the HIP events rt_render records on its launch stream (last step; steps are identical)
    k_ms = last_ms()
    t = torch.tensor([dt, k_ms], dtype=torch.float64, device=dev)
    if grouped:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt, k_ms = float(t[0]), float(t[1])

    if rank == 0:
        sec_per_step = dt / a.steps
        paths = nx * ny * spp
        value = paths / sec_per_step / 1e6
        paths_per_launch = nx * ny * (s1 - s0)
        counts = {} if cfg.get("no_counts") else load_work_counts().get(cfg["scene"], {})
        if cfg["bound"] == "hbm":
            per = cfg.get("bytes_per_path") or counts.get("bytes_per_path")
            achieved = paths_per_launch * per / (k_ms * 1e-3) / 1e9 if per else None
            peak, unit, note = HBM_PEAK_GBS, "GB/s", ("algorithmic bytes: tree nodes visited x 32 B + leaves tested x 32 B per path; "
                                                "most are served by the L1s / L2 (l2_hit), so achieved may exceed the HBM peak: "
                                                "traffic = what reached the memory side of the L2 (profiles/c5_traffic.json)")
        else:
            per = cfg.get("flop_per_path") or counts.get("flop_per_path")
            achieved = paths_per_launch * per / (k_ms * 1e-3) / 1e12 if per else None
            peak, unit = FP32_VECTOR_PEAK_TFLOPS, "TFLOP/s"
            note = "no dense contraction on this path: MFMA unused, FP32 vector peak = FP32 matrix peak"
        partition = (f"disjoint spp ranges over {world} GPUs + one RCCL sum-reduce of float4[{nx}*{ny}] onto rank 0" if world > 1 else
                     "one GPU renders every sample; no reduce at N=1" + (" (rehearsal: a one-rank RCCL reduce per step)" if grouped else ""))
        out = {
            "metric": "Mpaths/sec (node), Cornell box 1024x1024 1024spp depth 50" if a.config == "c2" else
                      f"Mpaths/sec (node), {cfg['scene']} {cfg['nx']}x{cfg['ny']} {cfg['spp']}spp depth {DEPTH}",
            "value": round(value, 2), "unit": "Mpaths/s", "n_gpus": world, "n_ranks_seen": n_ranks_seen, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(sec_per_step * 1e3, 3), "seconds_to_image": round(sec_per_step, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic" if not hook else "synthetic (CPU test hook: launch/partition/reduce logic only, not a measurement)",
            "config": {"workload": f"{cfg['scene']} {nx}x{ny} {spp}spp depth {DEPTH} ({cfg['name']})", "partition": partition},
            "roofline": {"bound": cfg["bound"], "achieved": None if achieved is None else round(achieved, 4), "peak": peak,
                         "unit": unit, "frac": None if achieved is None else round(achieved / peak, 5),
                         "traffic": (traffic_from_profile(a.config) or {}).get("hbm_bytes_per_launch"),
                         "l2_hit": (traffic_from_profile(a.config) or {}).get("l2_hit"), "kernel": kernel_name(), "kernel_ms": round(k_ms, 3),
                         "per_path": per, "note": note},
        }
        tr = traffic_from_profile(a.config) or {}
        if tr.get("lane_steps_per_path") and tr.get("gather_limit_lane_steps_per_s"):
            # config 5: the per-lane walk against the measured throughput of dependent 32-byte gathers (tools/ubench/gather_chain.hip)
            rate = paths_per_launch * tr["lane_steps_per_path"] / (k_ms * 1e-3)
            out["roofline"]["gather"] = {"achieved": round(rate, 1), "limit": tr["gather_limit_lane_steps_per_s"], "unit": "lane-steps/s (32 B each)",
                                         "frac": round(rate / tr["gather_limit_lane_steps_per_s"], 4)}
        if world == 1 and not a.no_cpu and not hook:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
