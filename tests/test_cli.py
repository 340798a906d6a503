"""The command-line driver (ray_tracing_amd/cli/raytracer.cpp) against the reference executable's behaviour.

CPU part: option parsing and console text equal the compiled reference's, word for word, on the command lines
of tests/golden/cli_text.json (written by oracle/make_cli_golden.py from oracle/_ref/rt_ref); without a GPU the
program fails loudly.  GPU part: the PNG files hold exactly the pixels rt_resolve gives through the C-ABI."""
import json
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "ray_tracing_amd", "bin", "raytracer")
CASES = json.load(open(os.path.join(ROOT, "tests", "golden", "cli_text.json")))


@pytest.fixture(scope="module", autouse=True)
def built():
    if not os.path.exists(EXE):  # normally built by __graft_entry__.build(); g++ only, seconds
        subprocess.run(["make", "-C", os.path.join(ROOT, "ray_tracing_amd"), "bin/raytracer"], check=True)


def run(args, cwd=None):
    return subprocess.run([EXE] + list(args), capture_output=True, text=True, timeout=300, cwd=cwd)


def read_png(path):
    """Minimal decoder for 8-bit RGBA, non-interlaced PNG; returns uint32 [ny][nx] words 0xAABBGGRR."""
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, nx, ny = 8, b"", 0, 0
    while pos < len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        (crc,) = struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])
        assert zlib.crc32(typ + body) == crc, "bad chunk checksum"
        if typ == b"IHDR":
            nx, ny, depth, ctype, comp, flt, lace = struct.unpack(">IIBBBBB", body)
            assert (depth, ctype, comp, flt, lace) == (8, 6, 0, 0, 0)
        elif typ == b"IDAT":
            idat += body
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(ny, 4 * nx + 1)
    assert not raw[:, 0].any(), "only filter 0 is written"
    return np.ascontiguousarray(raw[:, 1:]).view("<u4").reshape(ny, nx)


@pytest.mark.parametrize("case", CASES, ids=[" ".join(c["args"]) for c in CASES])
def test_console_text_equals_the_reference(case):
    p = run(case["args"])
    assert p.returncode == 0, p.stderr
    assert p.stdout == case["stdout"]


def test_bad_command_lines_are_rejected():
    for args in (["--no-such-option"], ["-q"], ["-w"], ["-w", "abc", "-x"], ["--path", "cpu", "-w", "4", "-h", "4"]):
        p = run(args)
        assert p.returncode != 0 and "raytracer:" in p.stderr, args


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_no_device_is_a_loud_failure(tmp_path):
    p = run(["-w", "8", "-h", "8", "-s", "1", "-o", str(tmp_path / "o.png")])
    assert p.returncode != 0 and "device" in p.stderr.lower()
    assert not (tmp_path / "o.png").exists()


def _accum_file(path, nx, ny, samples, nss=1, depth=50, seed=1, scene=b"cornell_box"):
    with open(path, "wb") as f:
        f.write(b"RTACC1\0\0" + struct.pack("<5iI", nx, ny, samples, nss, depth, seed) + scene.ljust(40, b"\0"))
        f.write(np.zeros((ny, nx, 4), np.float32).tobytes())


def test_accumulator_file_of_another_render_is_refused(tmp_path):
    """--accum: the file must come from the same scene, size, -S, -r and --seed (checked before any device work)."""
    f = str(tmp_path / "a.rtacc")
    _accum_file(f, 8, 8, 2)
    for args in (["-w", "16", "-h", "8"], ["-w", "8", "-h", "8", "-r", "7"], ["-w", "8", "-h", "8", "--seed", "2"],
                 ["-w", "8", "-h", "8", "-S", "2"], ["-w", "8", "-h", "8", "--scene", "random_spheres"]):
        p = run(args + ["-s", "4", "--accum", f] + ([] if "-r" in args else ["-r", "50"]))
        assert p.returncode != 0 and "is not from this render" in p.stderr, args
    open(f, "wb").write(b"RTACC1\0\0 short")
    assert "is not from this render" in run(["-w", "8", "-h", "8", "-s", "4", "-r", "50", "--accum", f]).stderr
    assert "--save-every needs --accum" in run(["-w", "8", "-h", "8", "--save-every", "2"]).stderr
    assert "do not go together" in run(["-w", "8", "-h", "8", "-r", "50", "-m", "--accum", str(tmp_path / "none")]).stderr


@pytest.mark.gpu
def test_progressive_render_save_and_resume(tmp_path):
    """Rendering 12 samples in one go, in saved steps of 4, or as 5 samples now and the rest after a restart gives the same
    sums: the paths are keyed by (pixel, sample), so only float summation order can differ."""
    nx, ny, ns = 48, 32, 12
    common = ["-w", str(nx), "-h", str(ny), "-r", "20", "--seed", "4"]

    def sums(path):
        raw = open(path, "rb").read()
        head = struct.unpack("<5iI", raw[8:32])
        return head, np.frombuffer(raw[72:], np.float32).reshape(ny, nx, 4)

    a, b, c = (str(tmp_path / n) for n in ("a.rtacc", "b.rtacc", "c.rtacc"))
    assert run(common + ["-s", str(ns), "--accum", a, "-o", str(tmp_path / "a.png")]).returncode == 0
    assert run(common + ["-s", str(ns), "--accum", b, "--save-every", "4", "-o", str(tmp_path / "b.png")]).returncode == 0
    assert run(common + ["-s", "5", "--accum", c, "-o", str(tmp_path / "c5.png")]).returncode == 0
    assert sums(c)[0][2] == 5
    p = run(common + ["-s", str(ns), "--accum", c, "-o", str(tmp_path / "c.png")])
    assert p.returncode == 0 and "Resuming at sample 5 of 12" in p.stdout
    (ha, sa), (hb, sb), (hc, sc_) = sums(a), sums(b), sums(c)
    assert ha == hb == hc == (nx, ny, ns, 1, 20, 4)
    assert np.all(sa[..., 0] == ns) and np.all(sb[..., 0] == ns) and np.all(sc_[..., 0] == ns)
    assert np.allclose(sa, sb, rtol=1e-4, atol=1e-5) and np.allclose(sa, sc_, rtol=1e-4, atol=1e-5)
    pa, pb, pc = (read_png(str(tmp_path / n)).view(np.uint8).astype(int) for n in ("a.png", "b.png", "c.png"))
    assert np.abs(pa - pb).max() <= 1 and np.abs(pa - pc).max() <= 1
    # a finished file renders nothing more
    p = run(common + ["-s", str(ns), "--accum", c, "-o", str(tmp_path / "c2.png")])
    assert p.returncode == 0 and np.array_equal(read_png(str(tmp_path / "c2.png")), read_png(str(tmp_path / "c.png")))


@pytest.mark.gpu
def test_png_pixels_equal_rt_resolve(tmp_path):
    from ray_tracing_amd import api
    from ray_tracing_amd.api import HostScene, Scene
    nx, ny, ns = 96, 64, 6
    out = tmp_path / "img.png"
    p = run(["-w", str(nx), "-h", str(ny), "-s", str(ns), "-S", "2", "-r", "12", "--seed", "9", "-o", str(out), "-m", "-p"])
    assert p.returncode == 0, p.stderr
    assert "Single pass finished" in p.stdout and "Writing file." in p.stdout and f"{ns}/{ns} : 100.00%" in p.stdout
    sc = Scene(HostScene("cornell_box", nx / ny), 0)
    total = np.zeros((ny, nx, 4), np.float32)
    for s in range(ns):
        acc, rgba = api.render_to_host(sc, nx, ny, s, s + 1, sub_samples=2, max_depth=12, seed=9)
        got = read_png(str(tmp_path / f"img_{s}.png"))
        assert np.array_equal(got, rgba), f"sample {s}"
        total += acc
    # the final image: sums in sample order, then the reference's resolve (src/main.cpp:708-743)
    import oracle_lib as O
    got = read_png(str(out))
    assert np.array_equal(got, np.asarray(O.resolve(total)).reshape(ny, nx))
    c = np.sqrt(total[..., [3, 2, 1]] * np.float32(1.0 / ns), dtype=np.float32).clip(0, 1)
    q = (c * np.float32(255.99)).astype(np.uint32)
    assert np.array_equal(got, 0xFF000000 | (q[..., 2] << 16) | (q[..., 1] << 8) | q[..., 0])
    # passes: primary hits at pixel centres
    nrm = read_png(str(tmp_path / "img_NORMAL.png"))
    dep = read_png(str(tmp_path / "img_DEPTH.png"))
    assert (nrm >> 24 == 0xFF).all() and (dep >> 24 == 0xFF).all()
    assert len(np.unique(nrm)) >= 5  # walls, floor, ceiling, box faces, sphere
    assert (dep & 0xFF).max() == 255
    assert os.path.exists(tmp_path / "img_UV.png")
    sc.close()


@pytest.mark.gpu
def test_single_call_render_equals_the_api(tmp_path):
    from ray_tracing_amd import api
    from ray_tracing_amd.api import HostScene, Scene
    nx, ny, ns = 64, 64, 16
    out = tmp_path / "o.png"
    p = run(["-w", str(nx), "-h", str(ny), "-s", str(ns), "-r", "50", "-o", str(out), "--scene", "cornell_box_volumes"])
    assert p.returncode == 0, p.stderr
    sc = Scene(HostScene("cornell_box_volumes", nx / ny), 0)
    acc, rgba = api.render_to_host(sc, nx, ny, 0, ns, max_depth=50, seed=1)
    got = read_png(str(out))
    d = np.abs(got.view(np.uint8).astype(int) - rgba.view(np.uint8).astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 0.01  # float atomics may reorder sums between two runs
    sc.close()


@pytest.mark.gpu
def test_native_multi_device_render_with_rccl_on_one_device(tmp_path):
    """rt_render_multi_to_host / `raytracer --devices 0 --reduce-always`: the one-process multi-GPU path (a scene, a stream and
    an accumulator per device, disjoint sample ranges, ONE RCCL sum-reduce onto the first device, resolve there).  This box
    has one GPU, so the device list is degenerate, but everything runs: librccl is opened, a communicator is made, the
    reduce is issued on the render stream.  The image must equal rt_render_to_host's (same paths, same sums)."""
    import ctypes as C
    from ray_tracing_amd import _capi, api
    lib = _capi.load()
    nx, ny, ns = 64, 40, 12
    hs = api.HostScene("cornell_box", nx / ny)
    sc = api.Scene(hs, 0)
    want_acc, want_img = api.render_to_host(sc, nx, ny, 0, ns, seed=4)
    sc.close()
    p = api.make_params(nx, ny, 0, ns, seed=4)
    acc = np.zeros((ny, nx, 4), np.float32)
    img = np.zeros((ny, nx), np.uint32)
    dev = (C.c_int32 * 1)(0)
    for flags in (0, _capi.RT_MULTI_REDUCE_ALWAYS):
        acc[:] = 0
        st = lib.rt_render_multi_to_host(hs.desc, dev, 1, C.byref(p), flags, acc.ctypes.data_as(C.c_void_p), img.ctypes.data_as(C.c_void_p))
        assert st == _capi.RT_OK, lib.rt_last_error()
        assert np.all(acc[..., 0] == ns)
        assert np.allclose(acc, want_acc, rtol=1e-5, atol=1e-6)
        assert (img == want_img).mean() > 0.999
    # the executable: same pixels as the single-device run
    a, b = str(tmp_path / "a.png"), str(tmp_path / "b.png")
    common = ["-w", str(nx), "-h", str(ny), "-s", str(ns), "-r", "50", "--seed", "4"]
    pa = run(common + ["-o", a])
    pb = run(common + ["-o", b, "--devices", "0", "--reduce-always"])
    assert pa.returncode == 0 and pb.returncode == 0, pa.stderr + pb.stderr
    assert (read_png(a) == read_png(b)).mean() > 0.999
    assert run(common + ["-o", b, "--devices", "0,0"]).returncode != 0
    assert run(common + ["-o", b, "--devices", "x"]).returncode != 0


@pytest.mark.gpu
def test_multi_context_renders_image_after_image(tmp_path):
    """rt_multi_create / rt_multi_render_to_host / rt_multi_destroy: the context (scene, stream, accumulator per device, ONE
    communicator) is built once and K images rendered on it equal K single-device renders -- different seeds, a larger image
    after a smaller one (the buffers grow), the 8-bit image alone (h_accum NULL).  One GPU here, so the device list is
    degenerate; RT_MULTI_REDUCE_ALWAYS makes the one-rank communicator and the reduce run on every image."""
    import ctypes as C
    from ray_tracing_amd import _capi, api
    lib = _capi.load()
    hs = api.HostScene("cornell_box", 1.0)
    sc = api.Scene(hs, 0)
    dev = (C.c_int32 * 1)(0)
    for flags in (0, _capi.RT_MULTI_REDUCE_ALWAYS):
        ctx = C.c_void_p()
        assert lib.rt_multi_create(hs.desc, dev, 1, flags, C.byref(ctx)) == _capi.RT_OK, lib.rt_last_error()
        for nx, ny, ns, seed in ((48, 48, 6, 1), (48, 48, 6, 2), (80, 80, 4, 3), (32, 32, 9, 4)):
            want_acc, want_img = api.render_to_host(sc, nx, ny, 0, ns, seed=seed)
            p = api.make_params(nx, ny, 0, ns, seed=seed)
            acc = np.zeros((ny, nx, 4), np.float32)
            img = np.zeros((ny, nx), np.uint32)
            st = lib.rt_multi_render_to_host(ctx, C.byref(p), acc.ctypes.data_as(C.c_void_p), img.ctypes.data_as(C.c_void_p))
            assert st == _capi.RT_OK, lib.rt_last_error()
            assert np.all(acc[..., 0] == ns) and np.allclose(acc, want_acc, rtol=1e-5, atol=1e-6)
            assert (img == want_img).mean() > 0.999
            img2 = np.zeros((ny, nx), np.uint32)
            assert lib.rt_multi_render_to_host(ctx, C.byref(p), None, img2.ctypes.data_as(C.c_void_p)) == _capi.RT_OK
            assert (img2 == want_img).mean() > 0.999
            ms = C.c_float()
            assert lib.rt_multi_last_render_ms(ctx, C.byref(ms)) == _capi.RT_OK and ms.value > 0
        # a refused image leaves the context usable
        bad = api.make_params(48, 48, 5, 2)
        assert lib.rt_multi_render_to_host(ctx, C.byref(bad), acc.ctypes.data_as(C.c_void_p), None) != _capi.RT_OK
        assert lib.rt_multi_render_to_host(ctx, C.byref(p), None, None) == _capi.RT_ERR_INVALID
        p = api.make_params(48, 48, 0, 6, seed=1)
        acc = np.zeros((48, 48, 4), np.float32)
        assert lib.rt_multi_render_to_host(ctx, C.byref(p), acc.ctypes.data_as(C.c_void_p), None) == _capi.RT_OK
        assert np.allclose(acc, api.render_to_host(sc, 48, 48, 0, 6, seed=1)[0], rtol=1e-5, atol=1e-6)
        # options reach every device's scene: the exact path renders the same paths
        assert lib.rt_multi_set_option(ctx, _capi.RT_OPT_RENDER_PATH, _capi.RT_PATH_EXACT) == _capi.RT_OK
        assert lib.rt_multi_render_to_host(ctx, C.byref(p), acc.ctypes.data_as(C.c_void_p), None) == _capi.RT_OK
        assert np.all(acc[..., 0] == 6)
        lib.rt_multi_destroy(ctx)
    sc.close()
    # the executable: --repeat K renders K images on one context and prints each one's time
    out = str(tmp_path / "r.png")
    pr = run(["-w", "64", "-h", "40", "-s", "8", "-r", "50", "--seed", "4", "-o", out, "--devices", "0", "--reduce-always", "--repeat", "3"])
    assert pr.returncode == 0, pr.stderr
    assert pr.stdout.count("image ") == 3 and "context on 1 device(s)" in pr.stdout
    one = str(tmp_path / "one.png")
    assert run(["-w", "64", "-h", "40", "-s", "8", "-r", "50", "--seed", "4", "-o", one]).returncode == 0
    assert (read_png(out) == read_png(one)).mean() > 0.999
