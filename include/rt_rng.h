/* rt_rng.h -- definition of the per-path counter-based random stream.
 *
 * Replaces the reference's single global engine
 *     global std::default_random_engine generator;           (src/main.cpp:62)
 *     #define RAN01() distribution(generator)                (src/main.cpp:63-64)
 * which every worker thread shares without a lock.  On the GPU each path
 * (pixel, sample) owns a stream: draw number n of a path is a pure function of
 * (seed, pixel, sample, n), so an image does not depend on how paths are
 * scheduled over lanes, waves or GPUs.
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

/* pixel = row * nx + column with row 0 at the top of the image (the order of
 * the output buffer); sample = sample_index * sub_samples + sub_sample. */
RT_RNG_FN uint32_t rt_path_key(uint32_t seed, uint32_t pixel, uint32_t sample)
{
    uint32_t k = rt_mix32(pixel * 0x9E3779B9U + 0x85EBCA6BU);
    k = rt_mix32(k ^ (sample * 0xC2B2AE35U + seed * 0x27D4EB2FU + 1U));
    return k;
}

/* n-th draw of a path, uniform in [0,1) with 24 random bits */
RT_RNG_FN float rt_draw(uint32_t key, uint32_t n)
{
    uint32_t x = rt_mix32(key + n * 0x9E3779B9U);
    return (float)(x >> 8) * (1.0f / 16777216.0f);
}

#endif /* RT_RNG_H_ */
