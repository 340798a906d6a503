/* rt_rng.h -- definition of the per-path counter-based random stream.
 *
 * Replaces the reference's single global engine
 *     global std::default_random_engine generator;           (src/main.cpp:62)
 *     #define RAN01() distribution(generator)                (src/main.cpp:63-64)
 * which every worker thread shares without a lock.  On the GPU each path
 * (pixel, sample) owns a stream: draw number n of a path is a pure function of
 * (seed, pixel, sample, n), so an image does not depend on how paths are
 * scheduled over lanes, waves or GPUs.  A path's identity has 64 bits: a 32-bit
 * key of (sample, seed) and the 32-bit pixel index (rt_mix32_w below).
 *
 * The kernels consume draws in the order the reference's code does along one
 * path (jitter u, jitter v, lens disk, shutter time, then per bounce: media met
 * during traversal, then the material / pdf draws).  The CPU oracle can run on
 * this same stream, which makes per-pixel comparison possible.
 * Two deliberate differences, both where the reference draws numbers and then
 * multiplies them by an exact zero: when the camera's lens radius is exactly
 * zero the lens-disk rejection loop is not run at all (src/camera.h:56), and
 * when a metal's fuzz is exactly zero neither is the unit-sphere rejection loop
 * of its scatter (src/material.h:71: reflected + fuzz*random_in_unit_sphere()).
 * No draws are spent on either; the image has the same distribution (on the
 * GPU the second loop, run by whole waves until their slowest lane accepts,
 * was 9 % of the Cornell box's render time for a vector that was never used).
 *
 * This header is plain C and is included by host code, the oracle and the HIP
 * kernels alike; it is part of the ABI (changing it changes every image).
 */
#ifndef RT_RNG_H_
#define RT_RNG_H_

#include <stdint.h>

#if defined(__HIPCC__) || defined(__CUDACC__)
#define RT_RNG_FN static __host__ __device__ __forceinline__
#else
#define RT_RNG_FN static inline
#endif

/* 32-bit finaliser ("lowbias32"): bijective, full avalanche in two multiplies */
RT_RNG_FN uint32_t rt_mix32(uint32_t x)
{
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

/* The finaliser over two words: the second one is XORed into the first before the mixing.  For a fixed `w` it is a
 * bijection of x, and for a fixed x a bijection of w: two streams that differ in `w` never produce the same word at the
 * same position, and because every input bit reaches every output bit through both multiplies, streams whose `w`
 * differ in a single low bit (neighbouring pixels) are uncorrelated (tests/test_rng.py measures it).
 * (Folding `w` in BETWEEN the two multiplies -- one round of mixing for it -- was tried first and is not enough: a
 * one-bit difference then shifts the product by a constant, and neighbouring pixels' draws came out with a
 * correlation of -0.49.)
 * `w` is the path's pixel index, so a path's identity is the 64-bit pair (key, pixel) although the state a lane carries
 * from draw to draw is one 32-bit counter beside the pixel word it carries anyway (rt_pixel_word: it is also where
 * the path's radiance goes).  The streams of a pixel are windows
 * of that pixel's own 2^32-long sequence mix(c ^ pixel).
 * Round 2's stream was keyed by 32 bits in all and every stream was a window of ONE sequence: at the 2^37 paths of a
 * 3840 x 2160 x 16384 spp image every key value was shared by about 32 paths in different pixels.  Now paths of
 * different pixels never share a stream, and within a pixel the keys of samples 0 .. 2^32-1 are all different
 * (rt_path_key is a bijection of `sample` for a fixed seed). */
RT_RNG_FN uint32_t rt_mix32_w(uint32_t x, uint32_t w) { return rt_mix32(x ^ w); }

/* sample = sample_index * sub_samples + sub_sample */
RT_RNG_FN uint32_t rt_path_key(uint32_t seed, uint32_t sample)
{
    return rt_mix32(sample * 0xC2B2AE35U + seed * 0x27D4EB2FU + 1U);
}

/* The pixel word of a render: the row in the upper and the column in the lower 16 bits, row 0 at the top of the image
 * (image sizes are at most 65536 x 65536).  A lane carries this ONE word as its path's place in the image -- the draws
 * use it as it is, and the tile-local index and the address in the accumulator come out of it with a shift and a mask
 * when the path ends -- where row * nx + column had to be recomputed at every place that draws. */
RT_RNG_FN uint32_t rt_pixel_word(uint32_t row, uint32_t column) { return (row << 16) | column; }

/* n-th draw of the path (key, pixel), uniform in [0,1) with 24 random bits.
 * pixel = rt_pixel_word(row, column) for a render; for the closest-hit entry points (rt_trace_rays) it is the index of
 * the ray in the batch and the key is rt_path_key(seed, 0). */
RT_RNG_FN float rt_draw(uint32_t key, uint32_t pixel, uint32_t n)
{
    uint32_t x = rt_mix32_w(key + n * 0x9E3779B9U, pixel);
    return (float)(x >> 8) * (1.0f / 16777216.0f);
}

#endif /* RT_RNG_H_ */
