"""include/rt_rng.h: the per-path random stream (replaces src/main.cpp:62-64).  CPU tests of the header's arithmetic as
the oracle compiled it; the kernels include the same header, and tests/test_gpu_parity.py compares them with the oracle
on the same stream."""
import numpy as np

import oracle_lib as O

U = np.uint32


def mix32(x):
    x = x.astype(np.uint32)
    x ^= x >> U(16)
    x *= U(0x7FEB352D)
    x ^= x >> U(15)
    x *= U(0x846CA68B)
    x ^= x >> U(16)
    return x


def mix32_w(x, w):
    return mix32(x.astype(np.uint32) ^ np.asarray(w, dtype=np.uint32))


def draws(seed, pixel, sample, n):
    with np.errstate(over="ignore"):
        key = mix32(np.array([sample], dtype=np.uint32) * U(0xC2B2AE35) + U(seed) * U(0x27D4EB2F) + U(1))
        x = mix32_w(key + np.arange(n, dtype=np.uint32) * U(0x9E3779B9), U(pixel))
    return (x >> U(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)


def test_header_arithmetic_is_the_documented_one():
    """The stream is part of the ABI (changing it changes every image): a numpy restatement of the documented formula
    must give the header's numbers bit for bit."""
    for seed, pixel, sample in [(0, 0, 0), (1, 5, 7), (77, 1024 * 1024 - 1, 16383), (0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF)]:
        got = O.rng_draws(seed, pixel, sample, 64)
        want = draws(seed, pixel, sample, 64)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (seed, pixel, sample)
        assert (got >= 0).all() and (got < 1).all()


def test_two_pixels_with_equal_sample_and_seed_draw_different_numbers():
    """Round 2's key had 32 bits in all, so paths of different pixels shared streams at BASELINE config 4's 2^37 paths.
    Now the pixel is a second word of the mixer: for one (sample, seed) no two pixels produce the same word at the same
    position (rt_mix32_w is a bijection of w for a fixed x), and their streams are uncorrelated -- also when the pixels
    differ in one bit (the first design, the pixel folded in between the two multiplies, failed exactly this: -0.49)."""
    lib = O.load()
    key = lib.orc_rng_key(3, 11)
    with np.errstate(over="ignore"):
        # bijection in w, checked on a window of 2^16 neighbouring pixels at the first four positions of the stream
        for n in range(4):
            x = U(key) + U(n) * U(0x9E3779B9)
            words = mix32_w(np.full(1 << 16, x, dtype=np.uint32), np.arange(1 << 16, dtype=np.uint32))
            assert len(np.unique(words)) == 1 << 16
        assert lib.orc_rng_mix_w(int(x), 12345) == int(mix32_w(np.array([x], dtype=np.uint32), U(12345))[0])
    a, b = O.rng_draws(3, 1000, 11, 4096), O.rng_draws(3, 1001, 11, 4096)
    assert not np.any(a == b) or (a == b).sum() < 4  # 24-bit outputs: equal by chance once in 2^24
    r = np.corrcoef(a, b)[0, 1]
    assert abs(r) < 4.0 / np.sqrt(4096)
    base = O.rng_draws(3, 1000, 11, 1 << 16)
    for bit in range(24):
        r = np.corrcoef(base, O.rng_draws(3, 1000 ^ (1 << bit), 11, 1 << 16))[0, 1]
        assert abs(r) < 4.5 / np.sqrt(1 << 16), (bit, r)
    # neighbouring pixels differ in one low bit of w: the very first draws (the jitter of the camera ray) must not be close
    first = np.array([O.rng_draws(0, p, 0, 2) for p in range(4096)])
    assert abs(np.corrcoef(first[:-1, 0], first[1:, 0])[0, 1]) < 0.06
    assert abs(first.mean() - 0.5) < 0.02 and abs(first.var() - 1.0 / 12.0) < 0.01


def test_keys_of_one_pixel_are_distinct_for_all_samples_of_a_seed():
    """rt_path_key is a bijection of `sample` for a fixed seed: checked on the first 2^20 samples of three seeds."""
    with np.errstate(over="ignore"):
        for seed in (0, 1, 0xDEADBEEF):
            k = mix32(np.arange(1 << 20, dtype=np.uint32) * U(0xC2B2AE35) + U(seed) * U(0x27D4EB2F) + U(1))
            assert len(np.unique(k)) == 1 << 20
    assert O.load().orc_rng_key(0xDEADBEEF, 5) == int(k[5])


def test_stream_is_uniform_and_serially_uncorrelated():
    d = O.rng_draws(9, 321, 4, 1 << 16).astype(np.float64)
    assert abs(d.mean() - 0.5) < 4 * np.sqrt(1 / 12 / len(d))
    hist = np.histogram(d, bins=64, range=(0, 1))[0]
    chi2 = ((hist - len(d) / 64) ** 2 / (len(d) / 64)).sum()
    assert chi2 < 64 + 5 * np.sqrt(2 * 64)
    assert abs(np.corrcoef(d[:-1], d[1:])[0, 1]) < 4 / np.sqrt(len(d))
