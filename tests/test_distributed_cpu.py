"""The N > 1 path on CPU: two gloo ranks, disjoint sample ranges, one sum-reduce, resolve on rank 0.

There is no GPU here, so each rank's accumulator comes from the CPU oracle (per-path random stream, the same
keying the kernels use); what is under test is ray_tracing_amd.distributed: the partition, the reduce and the
invariance of the image under the number of ranks."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ray_tracing_amd import distributed as D

NX, NY, SPP, SEED = 24, 16, 5, 4


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib as O
    from ray_tracing_amd.api import HostScene
    hs = HostScene("cornell_box", NX / NY)

    def render(s0, s1):
        acc, _, _ = O.render(hs, NX, NY, s0, s1, 1, 50, O.ORC_RNG_COUNTER, SEED, 2)
        return torch.from_numpy(acc)

    accum, image = D.render_image(render, lambda a: O.resolve(a.numpy()), SPP)
    if rank == 0:
        np.savez(out_path, accum=accum.numpy(), image=image)
    dist.barrier()
    dist.destroy_process_group()


def test_sample_ranges_tile_the_budget():
    for world in (1, 2, 3, 4, 8):
        for spp in (0, 1, 5, 8, 1024, 16384):
            r = [D.sample_range(g, world, spp) for g in range(world)]
            assert r[0][0] == 0 and r[-1][1] == spp
            assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1
    with pytest.raises(ValueError):
        D.sample_range(2, 2, 8)


def test_two_ranks_equal_one_rank(tmp_path):
    import oracle_lib as O
    from ray_tracing_amd.api import HostScene
    out = str(tmp_path / "r0.npz")
    mp.start_processes(_worker, args=(2, _free_port(), out), nprocs=2, join=True, start_method="spawn")
    got = np.load(out)
    full, _, _ = O.render(HostScene("cornell_box", NX / NY), NX, NY, 0, SPP, 1, 50, O.ORC_RNG_COUNTER, SEED, 2)
    assert np.all(got["accum"][..., 0] == SPP)
    assert np.allclose(got["accum"], full, rtol=1e-5, atol=1e-6)  # same paths, float sums in another order
    same = (got["image"] == O.resolve(full)).mean()
    assert same > 0.99


def test_native_sample_range_equals_the_python_partition():
    """rt_sample_range (what rt_render_multi_to_host and `raytracer --devices` split the budget with) tiles any sample
    range exactly and agrees with ray_tracing_amd.distributed.sample_range."""
    import ctypes as C
    from ray_tracing_amd import _capi
    lib = _capi.load()
    b, e = C.c_int32(), C.c_int32()
    for world in (1, 2, 3, 4, 8, 64):
        for s0, s1 in ((0, 0), (0, 1), (0, 5), (3, 11), (0, 1024), (100, 16384 + 100)):
            got = []
            for g in range(world):
                assert lib.rt_sample_range(g, world, s0, s1, C.byref(b), C.byref(e)) == _capi.RT_OK
                got.append((b.value, e.value))
                pb, pe = D.sample_range(g, world, s1 - s0)
                assert (b.value - s0, e.value - s0) == (pb, pe)
            assert got[0][0] == s0 and got[-1][1] == s1 and all(got[i][1] == got[i + 1][0] for i in range(world - 1))
    assert lib.rt_sample_range(2, 2, 0, 8, C.byref(b), C.byref(e)) == _capi.RT_ERR_INVALID
    assert lib.rt_sample_range(0, 0, 0, 8, C.byref(b), C.byref(e)) == _capi.RT_ERR_INVALID


def test_multi_device_render_rejects_bad_device_lists(cornell_host):
    import ctypes as C
    from ray_tracing_amd import _capi, api
    lib = _capi.load()
    p = api.make_params(8, 8, 0, 2)
    acc = np.zeros((8, 8, 4), np.float32)
    twice = (C.c_int32 * 2)(0, 0)
    assert lib.rt_render_multi_to_host(cornell_host.desc, twice, 2, C.byref(p), 0, acc.ctypes.data_as(C.c_void_p), None) == _capi.RT_ERR_INVALID
    assert b"listed twice" in lib.rt_last_error()
    assert lib.rt_render_multi_to_host(cornell_host.desc, twice, 0, C.byref(p), 0, acc.ctypes.data_as(C.c_void_p), None) == _capi.RT_ERR_INVALID


def test_multi_context_rejects_bad_arguments_before_any_device_work(cornell_host):
    """rt_multi_create (the persistent form: scenes, streams, ONE communicator built once): the device list is checked before
    anything touches a device, a failed create leaves a null handle, and the render call refuses a null context."""
    import ctypes as C
    from ray_tracing_amd import _capi, api
    lib = _capi.load()
    ctx = C.c_void_p(123)
    twice = (C.c_int32 * 2)(1, 1)
    assert lib.rt_multi_create(cornell_host.desc, twice, 2, 0, C.byref(ctx)) == _capi.RT_ERR_INVALID and ctx.value is None
    assert b"listed twice" in lib.rt_last_error()
    assert lib.rt_multi_create(cornell_host.desc, twice, 65, 0, C.byref(ctx)) == _capi.RT_ERR_INVALID
    assert lib.rt_multi_create(cornell_host.desc, twice, 1, 2, C.byref(ctx)) == _capi.RT_ERR_INVALID and b"flags" in lib.rt_last_error()
    assert lib.rt_multi_create(cornell_host.desc, None, 1, 0, C.byref(ctx)) == _capi.RT_ERR_INVALID
    assert lib.rt_multi_create(cornell_host.desc, twice, 1, 0, None) == _capi.RT_ERR_INVALID
    p = api.make_params(8, 8, 0, 2)
    acc = np.zeros((8, 8, 4), np.float32)
    assert lib.rt_multi_render_to_host(None, C.byref(p), acc.ctypes.data_as(C.c_void_p), None) == _capi.RT_ERR_INVALID
    ms = C.c_float()
    assert lib.rt_multi_last_render_ms(None, C.byref(ms)) == _capi.RT_ERR_INVALID
    assert lib.rt_multi_set_option(None, 1, 0) == _capi.RT_ERR_INVALID
    lib.rt_multi_destroy(None)  # a null context is ignored
