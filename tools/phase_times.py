#!/usr/bin/env python3
"""Where the waves of render_fastlanes_kernel spend their time: wave cycles in looks + refill / surface / shade / node steps / leaf tests
(needs a library built with -DRT_TIME_PHASES: tools/build_variant.sh phases -DRT_TIME_PHASES, then
RT_AMD_LIB=ray_tracing_amd/build_var/librt_phases.so).  usage: phase_times.py [--opt NAME=VALUE ...] scene:nx:ny:spp ..."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ray_tracing_amd import _capi, api
lib = _capi.load()
lib.rt_debug_phase_cycles.argtypes = [C.c_void_p, C.c_int]
opts, jobs = [], []
it = iter(sys.argv[1:])
for a in it:
    if a == "--opt":
        k, v = next(it).split("=")
        opts.append((getattr(_capi, "RT_OPT_" + k), int(v)))
    else:
        jobs.append(a)
for job in jobs:
    name, nx, ny, spp = job.split(":"); nx, ny, spp = int(nx), int(ny), int(spp)
    sc = api.Scene(api.HostScene(name, nx / ny), 0)
    for k, v in opts:
        sc.set_option(k, v)
    acc = torch.empty((ny, nx, 4), dtype=torch.float32, device="cuda:0")
    lib.rt_debug_phase_cycles(None, 1)
    api.render(sc, acc, nx, ny, 0, spp)
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 16)()
    lib.rt_debug_phase_cycles(out, 1)
    t = [int(x) for x in out][:12]
    tot = sum(t) or 1
    t[2] += sum(t[5:12])
    leaf = t[2] or 1
    print(f"{name} {nx}x{ny}x{spp}: looks+refill+hand-over {100 * t[0] / tot:.1f} %, surface {100 * t[3] / tot:.1f} %, shade {100 * t[4] / tot:.1f} %, "
          f"node steps {100 * t[1] / tot:.1f} %, leaf tests {100 * t[2] / tot:.1f} % of the waves' time ({sc.last_render_ms():.2f} ms)", flush=True)
    print("    leaf phase by branch (the stamps weigh: read the split, not the share): " + ", ".join(f"{n} {100 * t[k] / leaf:.1f} %" for k, n in
          ((5, "fetch + decode + chain"), (6, "sphere"), (7, "rect"), (8, "moving sphere"), (9, "box"), (10, "medium"), (11, "join"))) +
          " (the rest: entry and pop)", flush=True)
    sc.close()
