#!/usr/bin/env python3
"""bench.py -- headline benchmark: Cornell box 1024x1024, 1024 spp, depth 50 (BASELINE.json configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c1|c3|c5] [--no-extra] [--no-cpu]

With --gpus N > 1 and no WORLD_SIZE in the environment this process only LAUNCHES: it starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>` as a child (before
torch or HIP is touched here; never an exec), relays the child's output and returns its exit code.  Started by
torch.distributed.run itself (WORLD_SIZE set) it is one rank.

One "step" is one complete image: every rank renders its disjoint range of the samples into its own float4
accumulator (rt_render), the accumulators are summed onto rank 0 with one RCCL reduce (N > 1 only), rank 0
resolves to 8 bits (rt_resolve) and copies the 8-bit image to (pinned) host memory: SURVEY.md 8d's seconds-to-image ends
with the resolved image on the host.  The scene is uploaded once before the timed region (inputs resident in HBM).
The total work is fixed, so the scaling is strong.  Prints ONE JSON line on rank 0.

value       = nx*ny*spp / seconds per step / 1e6   [Mpaths/s, whole job]
roofline    = the render kernel against its bound.  Cornell box, random spheres, Next-Week final: FP32 vector peak
              (MFMA is unused on purpose: no dense contraction on this path): algorithmic FLOP per launch (DESIGN.md:
              kFLOP per path x paths per launch) / the kernel's duration from HIP events on the launch stream.
              1M-sphere field: HBM.  `achieved` there is what the counters saw reach the memory side of the L2 (bytes per
              path from the committed rocprofv3 passes x paths per launch / duration): the walks' algorithmic bytes
              (nodes and leaves touched x 32 B) are mostly served by the L1s and the L2 and stand beside it as
              roofline.algorithmic; a roofline fraction is never above 1.
              roofline.counters: what the committed profile of that workload says about the same kernel (vector issue,
              lanes on, waiting, the rays per path the GPU really traces): profiles/counters.json.
other_configs = (N = 1, unless --no-extra) BASELINE configs 1, 3 and 5 timed in the same process after the headline's
              steps, 1 warm-up + at least 2 steps each (as many as fill a third of a second), the same way: their numbers in
              the driver's own record.
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


def counters_from_profile(cfg):
    """profiles/counters.json: per workload what the committed rocprofv3 passes and the census builds say about its render
    kernel -- valu_issue_frac (vector instructions x 2 cycles over SIMDs x kernel cycles), lane_util (thread cycles / 64 x
    vector instructions), wait_frac (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES), gpu_rays_per_path (rays the kernel traces per
    path: it ends a path at weight zero, the reference carries it on) -- with the profile each number came from."""
    path = os.path.join(ROOT, "profiles", "counters.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f).get(cfg)


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
    ap.add_argument("--no-extra", action="store_true", help="skip configs 1, 3, 5 after the headline (other_configs)")
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

    def measure(config, steps, warmup, spp_override=None, size_override=None):
        """K timed images of one workload; returns rank 0's report (None on the other ranks)."""
        cfg = dict(CONFIGS[config])
        if size_override:
            nx, ny = size_override, max(1, size_override * cfg["ny"] // cfg["nx"])
        else:
            nx, ny = cfg["nx"], cfg["ny"]
        spp = spp_override or cfg["spp"]
        s0, s1 = D.sample_range(rank, world, spp)  # disjoint sample ranges

        if hook:
            mod, fn = hook.split(":")
            render, resolve, sync, last_ms, kernel_name = getattr(__import__(mod), fn)(cfg["scene"], nx, ny, DEPTH)
            scene = None
        else:
            from ray_tracing_amd import _capi, api
            scene = api.Scene(api.HostScene(cfg["scene"], nx / ny), local)
            accum_buf = torch.empty((ny, nx, 4), dtype=torch.float32, device=dev)
            image = torch.empty((ny, nx), dtype=torch.int32, device=dev)
            host_image = torch.empty((ny, nx), dtype=torch.int32, pin_memory=True)  # where the image ends up

            def render(b, e):
                return api.render(scene, accum_buf, nx, ny, b, e, max_depth=DEPTH, seed=0)

            def resolve(acc):
                api.resolve(acc, image)
                host_image.copy_(image, non_blocking=True)  # in stream order after the resolve; the fence below waits for it
                return host_image

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

        for _ in range(warmup):
            step()
        fence()
        if steps is None:
            # other_configs: enough images to time about a third of a second (at least 2, at most 50): a launch of a few
            # milliseconds is otherwise timed while the clocks are still coming up after the host-side scene build
            steps = max(2, min(50, int(0.35 / max(last_ms() * 1e-3, 1e-4))))
            for _ in range(min(steps, 5)):
                step()
            fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        dt = time.perf_counter() - t0
        # kernel duration from the HIP events rt_render records on its launch stream (last step; steps are identical)
        k_ms = last_ms()
        t = torch.tensor([dt, k_ms], dtype=torch.float64, device=dev)
        if grouped:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, k_ms = float(t[0]), float(t[1])
        kname = kernel_name()
        if scene is not None:
            scene.close()
        if rank != 0:
            return None

        sec_per_step = dt / steps
        paths = nx * ny * spp
        value = paths / sec_per_step / 1e6
        paths_per_launch = nx * ny * (s1 - s0)
        counts = {} if cfg.get("no_counts") else load_work_counts().get(cfg["scene"], {})
        tr = traffic_from_profile(config) or {}
        roof = {"bound": cfg["bound"]}
        if cfg["bound"] == "hbm":
            # what the counters saw reach the memory side of the L2, per path of the profiled launch, x this launch's paths
            per_path_hbm = tr["hbm_bytes_per_launch"] / tr["paths_per_launch"] if tr.get("paths_per_launch") else None
            per_alg = cfg.get("bytes_per_path") or counts.get("bytes_per_path")
            achieved = paths_per_launch * per_path_hbm / (k_ms * 1e-3) / 1e9 if per_path_hbm else None
            peak, unit, per = HBM_PEAK_GBS, "GB/s", per_path_hbm
            traffic = None if per_path_hbm is None else int(paths_per_launch * per_path_hbm)
            note = ("achieved = HBM-side bytes per path from the committed rocprofv3 passes (2 x FETCH_SIZE + WRITE_SIZE, gfx950 correction; "
                    "profiles/c5_traffic.json) x paths per launch / kernel time; the walks' algorithmic bytes are mostly L1 / L2 hits (l2_hit)")
            if per_alg:
                rate = paths_per_launch * per_alg / (k_ms * 1e-3) / 1e9
                roof["algorithmic"] = {"bytes_per_path": per_alg, "bytes_per_launch": int(paths_per_launch * per_alg), "rate": round(rate, 1),
                                       "unit": "GB/s", "note": "tree nodes visited x 32 B + leaves tested x 32 B; served by L1 / L2 / Infinity "
                                       "Cache as well as HBM, so this rate is not bounded by the HBM peak"}
        else:
            per = cfg.get("flop_per_path") or counts.get("flop_per_path")
            achieved = paths_per_launch * per / (k_ms * 1e-3) / 1e12 if per else None
            peak, unit = FP32_VECTOR_PEAK_TFLOPS, "TFLOP/s"
            traffic = tr.get("hbm_bytes_per_launch")
            note = ("no dense contraction on this path: MFMA unused, FP32 vector peak = FP32 matrix peak; per_path = the REFERENCE's "
                    "algorithmic work (SURVEY.md 8d costs x its own operation counts), which prices rays the kernel does not trace: "
                    "see counters.gpu_rays_per_path against the reference's rays per path")
        roof.update({"achieved": None if achieved is None else round(achieved, 4), "peak": peak, "unit": unit,
                     "frac": None if achieved is None else round(achieved / peak, 5), "traffic": traffic, "l2_hit": tr.get("l2_hit"),
                     "kernel": kname, "kernel_ms": round(k_ms, 3), "per_path": None if per is None else round(per, 1), "note": note})
        ctr = counters_from_profile(config)
        if ctr:
            roof["counters"] = ctr
        ref_rays = (counts.get("reference_tree_per_path") or {}).get("rays")
        if ref_rays:
            roof["reference_rays_per_path"] = ref_rays
        if tr.get("lane_steps_per_path") and tr.get("gather_rate_lane_steps_per_s"):
            # config 5: the per-lane walk beside the measured rate of bare chains of dependent 32-byte gathers at the walk's
            # locality (tools/ubench/gather_chain.hip).  A comparison, not a limit: the ubench's access pattern is harsher than the
            # walk's (profiles/r02_gather_chain_ubench.txt), so no fraction is printed
            rate = paths_per_launch * tr["lane_steps_per_path"] / (k_ms * 1e-3)
            roof["gather"] = {"achieved": round(rate, 1), "ubench": tr["gather_rate_lane_steps_per_s"], "unit": "lane-steps/s (32 B each)"}
        return {"cfg": cfg, "nx": nx, "ny": ny, "spp": spp, "steps": steps, "value": value, "sec_per_step": sec_per_step, "roofline": roof}

    m = measure(a.config, a.steps, a.warmup, a.spp, a.size)
    if rank == 0:
        cfg, nx, ny, spp = m["cfg"], m["nx"], m["ny"], m["spp"]
        partition = (f"disjoint spp ranges over {world} GPUs + one RCCL sum-reduce of float4[{nx}*{ny}] onto rank 0" if world > 1 else
                     "one GPU renders every sample; no reduce at N=1" + (" (rehearsal: a one-rank RCCL reduce per step)" if grouped else ""))
        out = {
            "metric": "Mpaths/sec (node), Cornell box 1024x1024 1024spp depth 50" if a.config == "c2" else
                      f"Mpaths/sec (node), {cfg['scene']} {cfg['nx']}x{cfg['ny']} {cfg['spp']}spp depth {DEPTH}",
            "value": round(m["value"], 2), "unit": "Mpaths/s", "n_gpus": world, "n_ranks_seen": n_ranks_seen, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(m["sec_per_step"] * 1e3, 3), "seconds_to_image": round(m["sec_per_step"], 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic" if not hook else "synthetic (CPU test hook: launch/partition/reduce logic only, not a measurement)",
            "config": {"workload": f"{cfg['scene']} {nx}x{ny} {spp}spp depth {DEPTH} ({cfg['name']})", "partition": partition,
                       "seconds_to_image": "render + reduce + resolve + copy of the 8-bit image to pinned host memory"},
            "roofline": m["roofline"],
        }
    # BASELINE configs 1, 3 and 5 in the same process (one GPU only; the headline's fields above are not touched)
    if world == 1 and a.config == "c2" and not a.no_extra and not hook and not a.spp and not a.size:
        others = []
        for name in ("c1", "c3", "c5"):
            o = measure(name, None, 1)
            c = o["cfg"]
            others.append({"workload": f"{c['scene']} {o['nx']}x{o['ny']} {o['spp']}spp depth {DEPTH} ({c['name']})",
                           "value": round(o["value"], 2), "unit": "Mpaths/s", "steps": o["steps"], "warmup": 1,
                           "ms_per_step": round(o["sec_per_step"] * 1e3, 3), "kernel": o["roofline"]["kernel"],
                           "kernel_ms": o["roofline"]["kernel_ms"], "roofline": o["roofline"]})
        out["other_configs"] = others
    if rank == 0:
        if world == 1 and not a.no_cpu and not hook:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
