"""Renderer for bench.py's RT_BENCH_TEST_HOOK: the CPU oracle on the per-path random stream, so that the CPU tests can
run bench.py's launch / partition / reduce / report logic without a GPU.  Test infrastructure only."""
import time

import torch


def oracle_renderer(scene_name, nx, ny, depth):
    import oracle_lib as O
    from ray_tracing_amd.api import HostScene
    hs = HostScene(scene_name, nx / ny)
    state = {"ms": 0.0}

    def render(s0, s1):
        t0 = time.perf_counter()
        acc, _, _ = O.render(hs, nx, ny, s0, s1, 1, depth, O.ORC_RNG_COUNTER, 0, 2)
        state["ms"] = (time.perf_counter() - t0) * 1e3
        return torch.from_numpy(acc)

    def resolve(acc):
        return O.resolve(acc.numpy())

    return render, resolve, (lambda: None), (lambda: state["ms"]), (lambda: "oracle (CPU test hook)")
