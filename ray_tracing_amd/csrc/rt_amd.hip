// rt_amd.hip -- kernels and C-ABI entry points of include/rt_amd.h (gfx950).
//
// Kernels
//   render_kernel_uniform / render_kernel_lanes (exact path, body render_exact), render_fast_kernel,
//   render_fast_block_kernel, render_fastlanes_kernel (fast paths)
//                   the per-pixel Monte Carlo loop: replaces compute_one_sample_task::run
//                   (src/main.cpp:217-245) + color() (src/main.cpp:91-127) + the thread pool.
//   resolve_kernel  src/main.cpp:708-743.
//   trace_kernel    world->hit() for a batch of rays (the parity surface).
//   camera_kernel   camera::get_ray for a batch of samples.
//
// render_exact: one wave64 owns one task = (8x8 pixel tile, range of samples).
// The wave keeps 64 paths in registers, one per lane.  Each iteration of its loop
//   1. refills dead lanes from the task's pool of (pixel, sample) items: a ballot of
//      the dead lanes, a prefix count (mbcnt) to give each a distinct item, ray
//      generation for those lanes (live paths stay packed in the wave, so the wave
//      never runs half empty while the pool has work);
//   2. traces all 64 rays through the scene with the wave-uniform scan of
//      rt_device.hpp (scene data in SGPRs);
//   3. shades: emission, scatter, mixture-pdf importance sampling, throughput update;
//      finished paths add their radiance to the tile's accumulators in LDS.
// When the pool is empty and every lane is dead the 64 pixel sums go to HBM, once.
// HBM traffic is therefore 16 B per pixel per task; everything else lives in
// registers, SGPRs and 1 KiB of LDS per wave.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "rt_device.hpp"
#include "rt_error.h"
#include "rt_fast.hpp"
#include "rt_fastlanes.hpp"
#include "rt_order.h"
#include "rt_records.h"

using namespace rt;

namespace {

struct RenderArgs {
    int32_t nx, ny;
    int32_t tiles_x, tiles_y;
    int32_t sample_begin;
    int32_t chunk_samples; // samples per task
    int32_t n_chunks;
    int32_t n_samples;     // sample_end - sample_begin
    int32_t sub_samples, max_depth;
    uint32_t seed;
    int32_t atomic_out;    // 1: add into accum with atomics (several tasks per pixel or ACCUMULATE)
    int32_t n_tasks;
    int32_t stack_levels;  // per-lane fast path: inner levels of the tree = levels of the LDS stacks (dynamic LDS)
    int32_t chain;         // per-lane fast path: 1 = a wave takes further tasks from the counter when its pool is empty
    int32_t shade_batch;   // per-lane fast path: lanes whose walk is over wait until this many are (RT_SHADE_BATCH)
    int32_t leaf_batch;    // per-lane fast path: lanes standing on a leaf wait until this many do (RT_LEAF_BATCH)
    int32_t regroup_rounds; // per-lane block kernel: (two node steps + one leaf test) this many times between two regroups
};

constexpr int WAVES_PER_BLOCK = 4;

// Tile sums -> HBM: [A, B, G, R] per pixel (src/main.cpp:238-242), col /= sub_samples (src/main.cpp:236).
// The 8x8 tile is 8 row segments of 128 contiguous bytes.  Lane L moves float L of each 64-float quarter,
// so one store / atomic instruction covers two whole 128-byte segments (the shape float atomics run at
// full rate with) instead of 64 lanes each touching one dword of its own 16-byte pixel.
__device__ __forceinline__ void flush_tile(const float *acc, int lane, int tile_x, int tile_y, const RenderArgs &A, int n_samples,
                                           float *__restrict__ accum)
{
    const float inv = 1.0f / (float)A.sub_samples;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int f = k * 64 + lane;          // float index inside the tile, row-major, 32 floats per row
        const int trow = f >> 5, tcol = (f >> 2) & 7, comp = f & 3;
        const int pix = trow * 8 + tcol;
        const int col = tile_x * 8 + tcol, row = tile_y * 8 + trow;
        const float v = comp == 0 ? (float)n_samples : acc[(3 - comp) * 64 + pix] * inv; // B=acc[128..], G=acc[64..], R=acc[0..]
        if (col < A.nx && row < A.ny) {
            float *dst = accum + 4 * ((size_t)row * A.nx + col) + comp;
            if (A.atomic_out) atomicAdd(dst, v);
            else *dst = v;
        }
    }
}

// The per-lane stacks of world_hit_ordered: [level][lane] per wave, 8 KiB per wave, only in the kernels that walk per lane.
template <bool LANES>
struct LaneStacks {
    static __device__ __forceinline__ int *get(int, int) { return nullptr; }
};
template <>
struct LaneStacks<true> {
    static __device__ __forceinline__ int *get(int wave, int lane)
    {
        __shared__ int stacks[WAVES_PER_BLOCK][RT_ORD_MAX_DEPTH * 64];
        return &stacks[wave][lane];
    }
};
// closest hit for the kernels below: the wave-uniform walk for small scenes; per lane for large ones, ordered and
// culled when the entries are a binary tree (always, for trees the flattener wrote), else in the reference's order
template <bool LANES>
__device__ __forceinline__ bool closest_hit(const DevScene &S, const Ray &r, float t_min, float t_max, bool alive, Rng &rng, Hit &h,
                                            int *stack)
{
    if (!LANES) return world_hit(S, r, t_min, t_max, alive, rng, h);
    if (S.ordered) return world_hit_ordered(S, r, t_min, t_max, alive, rng, h, stack);
    return world_hit_lanes(S, r, t_min, t_max, alive, rng, h);
}

// LANES: per-lane walk of the tree instead of the wave-uniform one (world_hit)
template <bool LANES>
__device__ __forceinline__ void render_exact(DevScene S, RenderArgs A, float *__restrict__ accum)
{
    __shared__ float tile_acc[WAVES_PER_BLOCK][64 * 3];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int task = blockIdx.x * WAVES_PER_BLOCK + wave;
    if (task >= A.n_tasks) return; // whole wave leaves; no block-level barrier is used below

    // task -> tile and sample chunk.  Consecutive tasks are consecutive chunks of one tile.
    const int tile = task / A.n_chunks, chunk = task - tile * A.n_chunks;
    const int tile_y = tile / A.tiles_x, tile_x = tile - tile_y * A.tiles_x;
    const int s0 = chunk * A.chunk_samples;
    const int s1 = min(s0 + A.chunk_samples, A.n_samples);
    const int total = 64 * (s1 - s0) * A.sub_samples; // items in the pool; item = q * 64 + pixel-in-tile

    float *acc = tile_acc[wave];
    acc[lane] = 0.0f, acc[64 + lane] = 0.0f, acc[128 + lane] = 0.0f;
    int *stack = LaneStacks<LANES>::get(wave, lane);

    int next = 0; // wave-uniform pool cursor
    bool alive = false;
    Ray r = Ray{v3(0, 0, 0), v3(0, 0, 1), 0.0f};
    V3 T = v3(1, 1, 1);
    Rng rng{0u, 0u};
    int depth = 0, pix = lane;

    for (;;) {
        // ---- 1. refill dead lanes (ballot + prefix count) ----------------------
        const unsigned long long dead = __ballot(!alive);
        if (dead != 0ull && next < total) {
            const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(dead >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)dead, 0u));
            const int item = next + rank;
            if (!alive && item < total) {
                pix = item & 63;
                const int q = item >> 6;
                const int sample = A.sample_begin + s0 + q / A.sub_samples;
                const int sub = q - (q / A.sub_samples) * A.sub_samples;
                const int col = tile_x * 8 + (pix & 7), row = tile_y * 8 + (pix >> 3); // row 0 = image top
                if (col < A.nx && row < A.ny) {
                    rng.start(rt_path_key(A.seed, (uint32_t)(sample * A.sub_samples + sub)), rt_pixel_word((uint32_t)row, (uint32_t)col));
                    // compute_one_sample_task::run, src/main.cpp:230-233; j counts rows from the bottom
                    const int j = A.ny - 1 - row;
                    const float u = float(col + rng.next()) / float(A.nx);
                    const float v = float(j + rng.next()) / float(A.ny);
                    const V3 disk = S.cam.lens_radius == 0.0f ? v3(0, 0, 0) : random_in_unit_disk(rng); // include/rt_rng.h
                    const float time01 = rng.next();
                    r = camera_ray(S.cam, u, v, disk, time01);
                    T = v3(1, 1, 1);
                    depth = 0;
                    alive = true;
                }
            }
            next += __popcll(dead);
        }
        if (!__any(alive)) {
            if (next >= total) break;
            continue; // only out-of-image pixels were drawn; draw again
        }

        // ---- 2. closest hit ------------------------------------------------------
        Hit h;
        const bool hit_any = closest_hit<LANES>(S, r, 0.001f, RT_FLT_MAX, alive, rng, h, stack);

        // ---- 3. shade -------------------------------------------------------------
        if (alive) {
            V3 L;
            if (!shade(S, hit_any, h, r, T, depth, A.max_depth, rng, L)) {
                alive = false;
                // de_nan, src/vec3.h:184-191
                if (L.x == L.x && L.x != 0.0f) atomicAdd(&acc[pix], L.x);
                if (L.y == L.y && L.y != 0.0f) atomicAdd(&acc[64 + pix], L.y);
                if (L.z == L.z && L.z != 0.0f) atomicAdd(&acc[128 + pix], L.z);
            }
        }
    }

    flush_tile(acc, lane, tile_x, tile_y, A, s1 - s0, accum);
}

// src/main.cpp:708-743
__global__ void resolve_kernel(const float4 *__restrict__ accum, int n, uint32_t *__restrict__ out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 a = accum[i]; // A B G R
    float inv = 1.0f / (float)(int)a.x;
    float c[3] = {sqrtf(a.w * inv), sqrtf(a.z * inv), sqrtf(a.y * inv)};
    uint32_t q[3];
    for (int k = 0; k < 3; ++k) {
        float x = c[k];
        if (x < 0.0f) x = 0.0f;
        if (x > 1.0f) x = 1.0f;
        q[k] = (uint32_t)(x * 255.99f);
    }
    out[i] = 0xff000000u | (q[2] << 16) | (q[1] << 8) | q[0];
}

// The wave-uniform walk takes 130 VGPRs if left alone (three waves per SIMD); held to 128 it runs at four (+17 %).  The
// per-lane walk is bound by its LDS stacks and keeps the compiler's choice.
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4)))
void render_kernel_uniform(DevScene S, RenderArgs A, float *__restrict__ accum) { render_exact<false>(S, A, accum); }
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK)
void render_kernel_lanes(DevScene S, RenderArgs A, float *__restrict__ accum) { render_exact<true>(S, A, accum); }

template <bool LANES>
__global__ __launch_bounds__(256) void trace_kernel(DevScene S, int n, const float *__restrict__ rays, float t_min, float t_max,
                                                    uint32_t seed, rt_hit *__restrict__ out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool alive = i < n;
    int k = alive ? i : 0;
    const float *f = rays + 7 * (size_t)k;
    Ray r{v3(f[0], f[1], f[2]), v3(f[3], f[4], f[5]), f[6]};
    Rng rng{rt_path_key(seed, 0u), (uint32_t)k};
    Hit h;
    static_assert(WAVES_PER_BLOCK * 64 == 256, "trace_kernel is launched with 256 threads");
    int *stack = LaneStacks<LANES>::get(__builtin_amdgcn_readfirstlane(threadIdx.x >> 6), threadIdx.x & 63);
    bool any = closest_hit<LANES>(S, r, t_min, t_max, alive, rng, h, stack);
    if (!alive) return;
    rt_hit o;
    o.prim_id = any ? S.prims[h.prim].user_id : -1;
    o.material = any ? h.material : -1;
    o.t = any ? h.t : 0.0f, o.u = any ? h.u : 0.0f, o.v = any ? h.v : 0.0f;
    o.p[0] = any ? h.p.x : 0.0f, o.p[1] = any ? h.p.y : 0.0f, o.p[2] = any ? h.p.z : 0.0f;
    o.normal[0] = any ? h.n.x : 0.0f, o.normal[1] = any ? h.n.y : 0.0f, o.normal[2] = any ? h.n.z : 0.0f;
    o.reserved = 0.0f;
    out[i] = o;
}

__global__ void camera_kernel(rt_camera cam, int n, const float *__restrict__ st, const float *__restrict__ lens,
                              const float *__restrict__ time01, float *__restrict__ rays)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    V3 disk = lens ? v3(lens[2 * i], lens[2 * i + 1], 0.0f) : v3(0, 0, 0);
    Ray r = camera_ray(cam, st[2 * i], st[2 * i + 1], disk, time01 ? time01[i] : 0.0f);
    float *o = rays + 7 * (size_t)i;
    o[0] = r.o.x, o[1] = r.o.y, o[2] = r.o.z, o[3] = r.d.x, o[4] = r.d.y, o[5] = r.d.z, o[6] = r.time;
}


// ---------------------------------------------------------------------------
// The same loop on the register-resident record table (rt_fast.hpp): the render
// path for scenes of at most 64 records.  Lane L loads record L once; after that
// the scene is never read from memory again.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void load_records(const uint32_t *__restrict__ recs, int lane, rtf::Regs &R)
{
    const uint4 a = reinterpret_cast<const uint4 *>(recs)[2 * lane], b = reinterpret_cast<const uint4 *>(recs)[2 * lane + 1];
    R.d[0] = a.x, R.d[1] = a.y, R.d[2] = a.z, R.d[3] = a.w, R.d[4] = b.x, R.d[5] = b.y, R.d[6] = b.z, R.d[7] = b.w;
}

template <bool TEX, bool FLAT>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void render_fast_kernel(DevScene S, const uint32_t *__restrict__ recs,
                                                                            rtf::Layout Y, RenderArgs A, float *__restrict__ accum)
{
    __shared__ float tile_acc[WAVES_PER_BLOCK][64 * 3];
    __shared__ uint32_t lrecs[(RT_REC_MAX + 1) * 8]; // the record table, for the broadcast reads of the flat scan
    rtf::stage_records(recs, lrecs, Y.n_stream, threadIdx.x, 64 * WAVES_PER_BLOCK);
    __syncthreads(); // before any wave leaves
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int task = blockIdx.x * WAVES_PER_BLOCK + wave;
    if (task >= A.n_tasks) return;
    rtf::Regs R;
    load_records(recs, lane, R);

    const int tile = task / A.n_chunks, chunk = task - tile * A.n_chunks;
    const int tile_y = tile / A.tiles_x, tile_x = tile - tile_y * A.tiles_x;
    const int s0 = chunk * A.chunk_samples;
    const int s1 = min(s0 + A.chunk_samples, A.n_samples);
    const int total = 64 * (s1 - s0) * A.sub_samples;
    const float inv_nx = 1.0f / float(A.nx), inv_ny = 1.0f / float(A.ny);

    float *acc = tile_acc[wave];
    acc[lane] = 0.0f, acc[64 + lane] = 0.0f, acc[128 + lane] = 0.0f;

    int next = 0;
    bool alive = false;
    Ray r = Ray{v3(0, 0, 0), v3(0, 0, 1), 0.0f};
    V3 T = v3(1, 1, 1);
    Rng rng{0u, 0u}; // (this kernel carries the pixel word: it has the register)
    int depth = 0, pix = lane;

    for (;;) {
        const unsigned long long dead = __ballot(!alive);
        if (dead != 0ull && next < total) {
            const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(dead >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)dead, 0u));
            const int item = next + rank;
            if (!alive && item < total) {
                pix = item & 63;
                const int q = item >> 6;
                const int qs = A.sub_samples == 1 ? q : q / A.sub_samples; // the division only when it is needed
                const int sample = A.sample_begin + s0 + qs;
                const int sub = q - qs * A.sub_samples;
                const int col = tile_x * 8 + (pix & 7), row = tile_y * 8 + (pix >> 3);
                if (col < A.nx && row < A.ny) {
                    rng.start(rt_path_key(A.seed, (uint32_t)(sample * A.sub_samples + sub)), rt_pixel_word((uint32_t)row, (uint32_t)col));
                    const int j = A.ny - 1 - row;
                    const float u = (float(col) + rng.next()) * inv_nx;
                    const float v = (float(j) + rng.next()) * inv_ny;
                    const V3 disk = S.cam.lens_radius == 0.0f ? v3(0, 0, 0) : rtf::rnd_unit_disk(rng); // include/rt_rng.h
                    const float time01 = rng.next();
                    r = rtf::camera_ray(S.cam, u, v, disk, time01);
                    T = v3(1, 1, 1);
                    depth = 0;
                    alive = true;
                }
            }
            next += __popcll(dead);
        }
        if (!__any(alive)) {
            if (next >= total) break;
            continue;
        }

        float t;
        int rec;
        rtf::scan<FLAT>(recs, lrecs, Y, r, 0.001f, RT_FLT_MAX, alive, t, rec);
#ifdef RT_MEASURE_SCAN_TWICE // development (tools/build_variant.sh): the same scan again, to price the scan by difference
        {
            float t2;
            int rec2;
            rtf::scan<FLAT>(recs, lrecs, Y, r, 0.0010001f, RT_FLT_MAX, alive, t2, rec2);
            if (t2 < t) t = t2, rec = rec2;
        }
#endif
        rtf::Surface sf;
        rtf::build_surface(R, Y, r, t, rec, sf); // every lane takes part in the cross-lane fetches

#ifdef RT_MEASURE_SHADE_TWICE // development (tools/build_variant.sh): the shade once more on a copy of the path, kept alive at zero weight
        float keep = 0.0f;
        if (alive) {
            Ray r2 = r;
            V3 T2 = T, L2;
            int depth2 = depth;
            Rng rng2{rng.x ^ 0x5bd1e995u, rng.w};
            const bool g2 = rtf::shade<TEX, TEX>(S, recs, Y, rec >= 0, sf, r2, T2, depth2, A.max_depth, rng2, L2);
            keep = (g2 ? r2.d.x + T2.y : L2.z) * 1e-30f;
        }
#endif
        if (alive) {
            V3 L;
            // (this kernel has the registers for the perlin corners written out: two_perlin_spheres spends most of its time there)
            if (!rtf::shade<TEX, TEX>(S, recs, Y, rec >= 0, sf, r, T, depth, A.max_depth, rng, L)) {
                alive = false;
                if (L.x == L.x && L.x != 0.0f) atomicAdd(&acc[pix], L.x);
                if (L.y == L.y && L.y != 0.0f) atomicAdd(&acc[64 + pix], L.y);
                if (L.z == L.z && L.z != 0.0f) atomicAdd(&acc[128 + pix], L.z);
            }
#ifdef RT_MEASURE_SHADE_TWICE
            T.x += keep * 1e-30f;
#endif
        }
    }

    flush_tile(acc, lane, tile_x, tile_y, A, s1 - s0, accum);
}

// ---------------------------------------------------------------------------
// render_fast_block_kernel: the same bounce loop, but the four waves of a block REGROUP their 256 paths between the
// scan and the shade.  A wave shades 64 paths of mixed fate -- some leave the scene, some hit the lamp, some refract,
// most scatter diffusely towards the light or into the cosine lobe -- one branch after the other, each with a few
// lanes on: the vector pipes are the limit and half the lanes are off (profiles/r01_headline_fast_path.txt).  Here, after
// the scan every lane settles what needs no company (miss, lamp, depth limit: the path ends) and names its class:
//     0 specular (metal / dielectric / isotropic)   1 lambertian sampling a rect light
//     2 lambertian sampling a sphere light or the cosine lobe                 3 dead
// The block sorts its paths by class through LDS (two barriers), so a wave shades mostly one class, and the dead lanes
// end up together in the last wave(s), where the refill from the block's pool runs with every lane on.  The draws keep
// the reference's order: the coin and the light member are drawn before the exchange, the direction after it.
// Work unit: a 16x16 pixel tile x a range of samples.
// ---------------------------------------------------------------------------
#ifndef RT_BLOCK_WAVES
#define RT_BLOCK_WAVES 4
#endif
constexpr int BLOCK_WAVES = RT_BLOCK_WAVES;
#ifdef RT_COUNT_SHADE // development (tools/shade_counts.py): who shades in render_fast_block_kernel, per wave and round
// [0] wave-rounds, [1] wave-rounds that scan, [2] lanes alive at the scan, [3] wave-rounds that run scatter_specular, [4] its lanes,
// [5] wave-rounds that run scatter_lambert, [6] its lanes, [7] wave-rounds with rect-light lanes, [8] those lanes,
// [9] wave-rounds with axis-sampling lanes, [10] those lanes, [11] wave-rounds that refill, [12] lanes refilled
__device__ unsigned long long g_shade_counts[16];
#endif
// Eight waves per SIMD: the kernel needs 58 VGPRs but, left alone, 106 SGPRs, and above 100 a SIMD of this generation holds
// seven waves.  Told to fit eight, the compiler keeps to 96 SGPRs; Cornell 1024 spp 80.6 -> 73.1 ms.
#ifndef RT_BLOCK_WAVES_PER_EU
#define RT_BLOCK_WAVES_PER_EU 8
#endif
// NW waves per block; the tile is (4*NW) x 16 pixels
// CHAIN: the block does not end with its task.  When the task's pool is empty it adds the tile's sums to the image, takes the
// next task from a counter in memory (`next_task`, zeroed by the host) and refills from that pool at once, while the last
// paths of the previous task are still on their way: those add their radiance to the image directly.  Without this the
// lanes of a block idle while the longest of its last paths finish (a quarter of a 16-paths-per-lane task).
template <bool FLAT, int NW, bool CHAIN>
__global__ __launch_bounds__(64 * NW)
__attribute__((amdgpu_waves_per_eu(RT_BLOCK_WAVES_PER_EU, RT_BLOCK_WAVES_PER_EU)))
void render_fast_block_kernel(DevScene S, const uint32_t *__restrict__ recs, rtf::Layout Y,
                                                                RenderArgs A, float *__restrict__ accum, unsigned *__restrict__ next_task)
{
    constexpr int NT = 64 * NW, TW = 4 * NW; // threads = pixels of the tile; tile width
    static_assert((TW & (TW - 1)) == 0 && TW <= 32, "the tile-local index is cut out of the pixel word with masks");
    __shared__ float acc[3 * NT];
    __shared__ uint4 xch[3][NT]; // the path state in three 16-byte pieces, [piece][slot]
    __shared__ uint2 fronts[4];  // [round & 3].{x,y} = slots taken so far from the bottom (specular) and from the top (diffuse)
    __shared__ unsigned pool_seen[4]; // [round & 3]: the block's one reading of pool_next in that round
    __shared__ unsigned pool_next;
    __shared__ int task_now;     // the task the pool belongs to (CHAIN)
    __shared__ uint32_t lrecs[(RT_REC_MAX + 1) * 8]; // the record table as rtf::stage_records lays it out: read by the scan (broadcast)
                                                     // and, per lane, by the surface and light fetches
    // the camera is only needed by the refill: read from LDS there, it does not hold 24 SGPRs for the whole kernel
    // (the kernel is held to 80 SGPRs for eight waves per SIMD)
    __shared__ __align__(16) rt_camera lcam;
    const int tid = threadIdx.x, lane = tid & 63;
    int task = blockIdx.x; // CHAIN: the first one; the counter hands out gridDim.x, gridDim.x + 1, ...
    rtf::stage_records(recs, lrecs, Y.n_stream, tid, NT);
    if (tid < (int)(sizeof(rt_camera) / 4)) reinterpret_cast<uint32_t *>(&lcam)[tid] = reinterpret_cast<const uint32_t *>(&S.cam)[tid];

    int tile_x, tile_y, s0, s1; // TW x 16 tiles
    unsigned total;
    // A path's place in the image is ONE word, rt_pixel_word(row, column) (include/rt_rng.h): it keys the path's random
    // stream, so it travels with the path (rng.w, a word of the exchange), and when the path ends it says where the
    // radiance goes -- into the tile's sums in LDS if the pixel lies in the tile the block works on, else (CHAIN: a path
    // of a tile that has gone to the image already) straight into the image.
    uint32_t tile_word = 0u; // the pixel word of the tile's first pixel
    constexpr uint32_t IN_TILE = (15u << 16) | (uint32_t)(TW - 1);
    auto open_task = [&](int tk) {
        const int tile = tk / A.n_chunks, chunk = tk - tile * A.n_chunks;
        tile_y = tile / A.tiles_x, tile_x = tile - tile_y * A.tiles_x;
        tile_word = rt_pixel_word((uint32_t)(tile_y * 16), (uint32_t)(tile_x * TW));
        s0 = chunk * A.chunk_samples;
        s1 = min(s0 + A.chunk_samples, A.n_samples);
        total = tk < A.n_tasks ? (unsigned)NT * (unsigned)((s1 - s0) * A.sub_samples) : 0u;
    };
    open_task(task);
    const float inv_nx = 1.0f / float(A.nx), inv_ny = 1.0f / float(A.ny), inv_sub = 1.0f / (float)A.sub_samples;
    // the tile's sums -> the image: [A, B, G, R] per pixel (src/main.cpp:238-242); every thread moves its own pixel.
    // (Moving whole 256-byte row segments per instruction, as flush_tile does, halves the HBM traffic of the atomics --
    // 205 MB instead of 400 MB per launch, at 5 GB/s either way -- but costs this kernel registers it does not have at eight
    // waves per SIMD: 1.3 % slower.)
    auto flush = [&](bool with_atomics) {
        const int col = tile_x * TW + (tid % TW), row = tile_y * 16 + (tid / TW);
        if (col < A.nx && row < A.ny) {
            float *dst = accum + 4 * ((size_t)row * A.nx + col);
            const float v4[4] = {(float)(s1 - s0), acc[2 * NT + tid] * inv_sub, acc[NT + tid] * inv_sub, acc[tid] * inv_sub};
            if (with_atomics) {
                for (int c = 0; c < 4; ++c) atomicAdd(dst + c, v4[c]);
            } else {
                *reinterpret_cast<float4 *>(dst) = make_float4(v4[0], v4[1], v4[2], v4[3]);
            }
        }
    };
    acc[tid] = 0.0f, acc[NT + tid] = 0.0f, acc[2 * NT + tid] = 0.0f;
    if (tid == 0) pool_next = 0u;
    if (tid < 4) fronts[tid] = make_uint2(0u, 0u);
    __syncthreads();

    bool alive = false;
    V3 o = v3(0, 0, 0), d = v3(0, 0, 1), T = v3(1, 1, 1);
    float time = 0.0f;
    Rng rng{0u, 0u};
    int depth = 0;
    int turn = 0, rk = 0;
#ifdef RT_COUNT_SHADE
    unsigned sc[16] = {0u};
#endif

    for (;;) {
        // ---- refill: dead lanes (together in the last waves after the regroup) take items from the block's pool --------
        const unsigned long long dead = __ballot(!alive);
        if (dead != 0ull) {
            unsigned base = 0u;
            if (lane == 0) base = atomicAdd(&pool_next, (unsigned)__popcll(dead)); // may run past `total`; bounded by the host
            base = __builtin_amdgcn_readfirstlane(base);
            if (base < total) {
                const unsigned rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(dead >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)dead, 0u));
                const unsigned item = base + rank;
                if (!alive && item < total) {
                    const int pix = (int)(item % (unsigned)NT);
                    const int q = (int)(item / (unsigned)NT);
                    const int qs = A.sub_samples == 1 ? q : q / A.sub_samples;
                    const int sample = A.sample_begin + s0 + qs;
                    const int sub = q - qs * A.sub_samples;
                    const int col = tile_x * TW + (pix % TW), row = tile_y * 16 + (pix / TW);
                    if (col < A.nx && row < A.ny) {
                        rng.start(rt_path_key(A.seed, (uint32_t)(sample * A.sub_samples + sub)), rt_pixel_word((uint32_t)row, (uint32_t)col));
                        const int j = A.ny - 1 - row;
                        const float u = (float(col) + rng.next()) * inv_nx;
                        const float v = (float(j) + rng.next()) * inv_ny;
                        const rt_camera cam = lcam;
                        const V3 disk = cam.lens_radius == 0.0f ? v3(0, 0, 0) : rtf::rnd_unit_disk(rng); // include/rt_rng.h
                        const float time01 = rng.next();
                        const Ray r = rtf::camera_ray(cam, u, v, disk, time01);
                        o = r.o, d = r.d, time = r.time;
                        T = v3(1, 1, 1);
                        depth = 0;
                        alive = true;
                    }
                }
            }
        }

#ifdef RT_COUNT_SHADE
        {
            const int na = __popcll(__ballot(alive)), nd = __popcll(dead);
            sc[0] += 1u, sc[1] += na > 0 ? 1u : 0u, sc[2] += (unsigned)na;
            sc[11] += (nd > 0 && na + nd > 64 - nd) ? 1u : 0u; // (approximate: a wave with dead lanes before the refill)
            sc[12] += (unsigned)(na - (64 - nd));
        }
#endif
        // ---- scan (waves without a live lane skip it) ---------------------------------------------------------------------
        rtf::Best best = rtf::best_start(RT_FLT_MAX);
        if (__any(alive)) best = rtf::scan_best<FLAT>(recs, lrecs, Y, Ray{o, d, time}, 0.001f, RT_FLT_MAX, alive);
#ifdef RT_MEASURE_SCAN_TWICE // development (tools/build_variant.sh): the same scan again, to price the scan by difference
        if (__any(alive)) {
            const rtf::Best b2 = rtf::scan_best<FLAT>(recs, lrecs, Y, Ray{o, d, time}, 0.0010001f, RT_FLT_MAX, alive);
            best = b2 < best ? b2 : best;
        }
#endif

        // ---- what needs no company: the path ends (miss, lamp, depth limit) or gets its class -----------------------------
        // The low word of the best hit is the key of the record (rt_records.h): kind, flip, material and index without a fetch;
        // the material record comes from the table in LDS, each lane reading its own.
        const uint32_t key = rtf::best_key_of(best);
        const bool hit = key != RT_REC_KEY_NONE;
        const float t = rtf::best_t_of(best);
        const int m = Y.materials_base + (hit ? (int)((key >> RT_KW_MAT_SHIFT) & RT_KW_MAT_MASK) : 0);
        const uint4 ma = *reinterpret_cast<const uint4 *>(lrecs + 8 * m);
        const uint32_t mat_kind = ma.x;
        int cls = 3, to_light = 0, light = 0;
        V3 p = o;
        if (alive) {
            if (!hit || !(depth < A.max_depth) || mat_kind == RT_MAT_DIFFUSE_LIGHT) {
                alive = false;
                if (hit && mat_kind == RT_MAT_DIFFUSE_LIGHT) { // radiance = T * emitted; de_nan, src/vec3.h:184-191
                    const uint2 mb = *reinterpret_cast<const uint2 *>(lrecs + 8 * m + 4);
                    const V3 L = v3(T.x * __uint_as_float(ma.w), T.y * __uint_as_float(mb.x), T.z * __uint_as_float(mb.y));
                    const uint32_t w = rng.w;
                    if (CHAIN && ((w ^ tile_word) & ~IN_TILE) != 0u) { // a path of a tile that has gone to the image already
                        float *dst = accum + 4 * ((size_t)(w >> 16) * A.nx + (w & 0xffffu));
                        if (L.x == L.x && L.x != 0.0f) atomicAdd(dst + 3, L.x * inv_sub);
                        if (L.y == L.y && L.y != 0.0f) atomicAdd(dst + 2, L.y * inv_sub);
                        if (L.z == L.z && L.z != 0.0f) atomicAdd(dst + 1, L.z * inv_sub);
                    } else {
                        const int pix = (int)(((w >> 16) & 15u) * (uint32_t)TW + (w & (uint32_t)(TW - 1)));
                        if (L.x == L.x && L.x != 0.0f) atomicAdd(&acc[pix], L.x);
                        if (L.y == L.y && L.y != 0.0f) atomicAdd(&acc[NT + pix], L.y);
                        if (L.z == L.z && L.z != 0.0f) atomicAdd(&acc[2 * NT + pix], L.z);
                    }
                }
            } else {
                ++depth;
                p = rtf::fma3(t, d, o);
                cls = 0;
                if (mat_kind == RT_MAT_LAMBERTIAN) {
                    cls = 2;
                    to_light = rng.next() < 0.5f ? 1 : 0;
                    if (to_light) {
                        light = rtf::lights_index(Y, rng);
                        cls = ((Y.rect_lights >> light) & 1u) ? 1 : 2; // a rect needs no basis; a sphere (or nothing) joins the cap lanes
                    }
                }
            }
        }

        // ---- regroup by class: ONE barrier ---------------------------------------------------------------------------------
        // The specular lanes take slots from the bottom of the exchange upwards, the diffuse lanes (within a wave: those that
        // sample a rect light, then the others) from the top downwards, each wave with one LDS atomic per end; what lies
        // between the two fronts is the dead lanes' -- they have no state to hand over and write nothing.  Nobody needs the
        // other waves' counts before writing, so the round has one barrier, after the writes; behind it the two totals say
        // what every slot holds.  (Round 2 published per-wave counts, waited, computed a prefix over 4 x 4 counts and waited
        // again: 40 vector instructions and a barrier more per round.)  The counters are used in turn, four sets: the set of
        // the next round is cleared before this round's barrier, when its last readers (two rounds back) are long gone.
        const unsigned long long b0 = __ballot(cls == 0), b1 = __ballot(cls == 1), b2 = __ballot(cls == 2);
        const int n0 = __popcll(b0), n1 = __popcll(b1), n12 = __popcll(b1 | b2);
        unsigned base_spec = 0u, base_diff = 0u;
        if (lane == 0) {
            base_spec = atomicAdd(&fronts[rk].x, (unsigned)n0);
            base_diff = atomicAdd(&fronts[rk].y, (unsigned)n12);
        }
        base_spec = __builtin_amdgcn_readfirstlane(base_spec), base_diff = __builtin_amdgcn_readfirstlane(base_diff);
        const unsigned long long mine = cls == 0 ? b0 : cls == 1 ? b1 : b2;
        const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mine >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mine, 0u));
        const int dst = cls == 0 ? (int)base_spec + rank : NT - 1 - ((int)base_diff + (cls == 2 ? n1 : 0) + rank);
        if (cls != 3) {
            // (the ray's time is not carried: nothing the record table can hold -- static spheres and rects -- reads it)
            // the record that was hit travels as its index and whether it is a box (where its key lies in the LDS copy)
            const uint32_t rec_bits = RT_KEY_INDEX(key) | ((key & 15u) == RT_REC_BOX ? 64u : 0u);
            const uint32_t packed = (uint32_t)depth | ((uint32_t)cls << 16) | ((uint32_t)to_light << 18) | ((uint32_t)light << 19) | (rec_bits << 23);
            xch[0][dst] = make_uint4(__float_as_uint(p.x), __float_as_uint(p.y), __float_as_uint(p.z), __float_as_uint(d.x));
            xch[1][dst] = make_uint4(__float_as_uint(d.y), __float_as_uint(d.z), __float_as_uint(T.x), __float_as_uint(T.y));
            xch[2][dst] = make_uint4(__float_as_uint(T.z), rng.x, packed, rng.w);
        }
        if (tid == 0) {
            fronts[(rk + 1) & 3] = make_uint2(0u, 0u);
            pool_seen[rk] = pool_next; // ONE reading of the pool for the whole block (the refills of this round may still be coming
                                       // in: the pool only shrinks, so a stale reading delays the block's decisions by a round at most)
        }
        __syncthreads();
        const uint2 front = fronts[rk];
        const unsigned pool_now = pool_seen[rk];
        rk = (rk + 1) & 3;
        const bool none_alive = front.x + front.y == 0u;
        if (!CHAIN && none_alive && pool_now >= total) break; // the same for every wave of the block
        // The slots are sorted by class, so read in place wave 0 would always get the costliest paths and the last wave the
        // waiting lanes -- and a block's waves sit on different SIMDs.  The block therefore reads the slots rotated by one
        // wave more every round: each SIMD sees every class in turn.
        const int slot = (tid + turn) & (NT - 1);
        turn += 64;
        const uint4 q0 = xch[0][slot], q1 = xch[1][slot], q2 = xch[2][slot];
        alive = (slot < (int)front.x) | (slot >= NT - (int)front.y); // between the fronts: nobody's (what the slot holds is stale)
        p = v3(__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z));
        d = v3(__uint_as_float(q0.w), __uint_as_float(q1.x), __uint_as_float(q1.y));
        T = v3(__uint_as_float(q1.z), __uint_as_float(q1.w), __uint_as_float(q2.x));
        rng.x = q2.y, rng.w = q2.w;
        const uint32_t word = q2.z;
        depth = (int)(word & 0xffffu), cls = alive ? (int)((word >> 16) & 3u) : 3;
        to_light = (int)((word >> 18) & 1u), light = (int)((word >> 19) & 15u);
        const int hit_rec = (int)((word >> 23) & 63u);
        const bool hit_box = ((word >> 29) & 1u) != 0u;
        o = p;
        if (CHAIN && pool_now >= total) { // the pool is empty (the same finding in every wave of the block)
            if (task >= A.n_tasks) { // nothing left to take
                if (none_alive) break;
            } else {
                // the tile goes to the image and the next task opens; the paths still on their way find, when they end, that
                // their pixel is not in the block's tile any more and add to the image themselves
                flush(true); // (every thread reads and clears the sums of its own pixel only: no barrier in between)
                acc[tid] = 0.0f, acc[NT + tid] = 0.0f, acc[2 * NT + tid] = 0.0f;
                if (tid == 0) {
                    task_now = (int)(atomicAdd(next_task, 1u) + gridDim.x);
                    pool_next = 0u;
                }
                __syncthreads();
                task = task_now;
                open_task(task);
            }
        }

        // ---- shade: the lanes of a wave now mostly share a class -------------------------------------------------------------
        if (__any(alive)) {
            rtf::Surface sf;
            {   // the record this lane's path hit: its key lies in dword 3 (box) or 4 of the LDS copy
                const uint32_t *r = lrecs + 8 * hit_rec;
                rtf::build_surface_lds(lrecs, Y, alive ? (hit_box ? r[3] : r[4]) : (uint32_t)RT_REC_RECT, p, sf); // (a dead lane: any valid key)
            }
            V3 nd = d, mul = sf.color;
            float ntime = time;
            bool go = alive;
#ifdef RT_COUNT_SHADE
            {
                const int n0 = __popcll(__ballot(alive & (cls == 0))), n1 = __popcll(__ballot(alive & (cls == 1))), n2 = __popcll(__ballot(alive & (cls == 2)));
                sc[3] += n0 > 0 ? 1u : 0u, sc[4] += (unsigned)n0;
                sc[5] += (n1 + n2) > 0 ? 1u : 0u, sc[6] += (unsigned)(n1 + n2);
                sc[7] += n1 > 0 ? 1u : 0u, sc[8] += (unsigned)n1;
                sc[9] += n2 > 0 ? 1u : 0u, sc[10] += (unsigned)n2;
            }
#endif
#ifdef RT_MEASURE_SHADE_TWICE // development: the scatter once more on another stream, its result kept at zero weight
            {
                Rng rng2{rng.x ^ 0x5bd1e995u, rng.w};
                V3 nd2 = d, mul2 = sf.color;
                bool go2 = alive;
                if (alive & (cls == 0)) rtf::scatter_specular(sf, d, rng2, nd2, mul2, ntime);
                if (alive & (cls != 0)) {
                    rtf::LightPick lp{RT_REC_LIGHT_NONE, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
                    if (to_light) lp = rtf::light_of_index_lds(lrecs, Y, light);
                    rtf::scatter_lambert(recs, Y, sf, sf.color, to_light != 0, lp, rng2, nd2, mul2, go2);
                }
                T.x += (go2 ? nd2.x + mul2.y : 1.0f) * 1e-37f;
            }
#endif
            if (alive & (cls == 0)) rtf::scatter_specular(sf, d, rng, nd, mul, ntime);
            if (alive & (cls != 0)) {
                rtf::LightPick lp{RT_REC_LIGHT_NONE, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
                if (to_light) lp = rtf::light_of_index_lds(lrecs, Y, light);
                rtf::scatter_lambert(recs, Y, sf, sf.color, to_light != 0, lp, rng, nd, mul, go);
            }
            d = nd, time = ntime;
            T = v3(T.x * mul.x, T.y * mul.y, T.z * mul.z);
            alive = go && (T.x != 0.0f || T.y != 0.0f || T.z != 0.0f); // a zero weight is final
        }
    }

#ifdef RT_COUNT_SHADE
    if (lane == 0)
        for (int k = 0; k < 13; ++k) atomicAdd(&g_shade_counts[k], (unsigned long long)sc[k]);
#endif
    if (!CHAIN) flush(A.atomic_out != 0); // CHAIN: every tile went out when the block moved on, the last one included (its pool
                                         // was empty, nothing was left to take: flushed below)
    if (CHAIN) flush(true);
}

// world->hit() on the record table: the SAME scan the render kernels of this scene run (FLAT: the run scan over the
// block's LDS copy of the table, as render_fast_block_kernel; else the generic one), so what rt_trace_rays reports on
// RT_PATH_FAST is what the bounces of rt_render see
template <bool FLAT>
__global__ __launch_bounds__(256) void trace_fast_kernel(const uint32_t *__restrict__ recs, const int32_t *__restrict__ rec_user,
                                                         rtf::Layout Y, int n, const float *__restrict__ rays, float t_min,
                                                         float t_max, rt_hit *__restrict__ out)
{
    __shared__ uint32_t lrecs[(RT_REC_MAX + 1) * 8];
    rtf::stage_records(recs, lrecs, Y.n_stream, threadIdx.x, 256);
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    rtf::Regs R;
    load_records(recs, lane, R);
    const bool alive = i < n;
    const int k = alive ? i : 0;
    const float *f = rays + 7 * (size_t)k;
    Ray r{v3(f[0], f[1], f[2]), v3(f[3], f[4], f[5]), f[6]};
    float t;
    int rec;
    rtf::scan<FLAT>(recs, lrecs, Y, r, t_min, t_max, alive, t, rec);
    rtf::Surface sf;
    rtf::build_surface(R, Y, r, t, rec, sf);
    const uint32_t kw = rtf::sh_u(R.d[7], rec < 0 ? 0 : rec);
    if (!alive) return;
    const bool any = rec >= 0;
    rt_hit o;
    // (a box record names its face: the ids of the six rects follow the 64 record ids)
    o.prim_id = any ? ((kw & 15u) == RT_REC_BOX ? rec_user[RT_REC_MAX + 6 * rec + sf.face] : rec_user[rec]) : -1;
    o.material = any ? (int)((kw >> RT_KW_MAT_SHIFT) & RT_KW_MAT_MASK) : -1;
    o.t = any ? t : 0.0f, o.u = 0.0f, o.v = 0.0f;
    o.p[0] = any ? sf.p.x : 0.0f, o.p[1] = any ? sf.p.y : 0.0f, o.p[2] = any ? sf.p.z : 0.0f;
    o.normal[0] = any ? sf.n.x : 0.0f, o.normal[1] = any ? sf.n.y : 0.0f, o.normal[2] = any ? sf.n.z : 0.0f;
    o.reserved = 0.0f;
    out[i] = o;
}

// ---------------------------------------------------------------------------
// The same loop for scenes that do not fit the record table: every lane walks the tree on its own
// (rt_fastlanes.hpp).  `recs` holds only the light records.
// ---------------------------------------------------------------------------
// Lanes do not wait for each other at the end of a bounce: a lane whose walk is over idles until RT_SHADE_BATCH lanes
// are in that state (or nobody walks any more); then those lanes are shaded, dead ones are refilled from the pool, and
// all of them start their next walk while the others are still in the middle of theirs.  Likewise lanes standing on a
// leaf wait until RT_LEAF_BATCH of them do (or nobody is opening nodes).  This keeps the lanes of a wave busy although
// every ray takes its own number of steps through the tree.
#ifndef RT_SHADE_BATCH
#define RT_SHADE_BATCH 48
#endif
#ifndef RT_LEAF_BATCH
#define RT_LEAF_BATCH 16
#endif
// Up to this many lanes of a shade phase that ask for a noise texture are served one by one by the whole wave
// (rtf::wave_noise_value); more, and every lane evaluates its own (0 in a build: never).
#ifndef RT_WAVE_NOISE_MAX
#define RT_WAVE_NOISE_MAX 16
#endif
// Node steps between two looks at who waits for a leaf test or for shading (template parameter STEPS; the looks cost a fifth
// of a node step).  1 / 2 / 3 / 4 steps: random spheres 6.1 / 6.4 / 6.3 / 6.2, Next-Week final 1.63 / 1.75 / 1.75 / 1.66, 1M spheres
// 1.04 / 1.09 / 1.15 / 1.18 Gpaths/s: two, and four for deep trees.
// Six waves per SIMD is what the LDS stacks allow (six blocks per CU, rt_bvh.h); left alone the kernels take 89 / 101 VGPRs
// (five / four waves).  Held to 80: random spheres +10 %, Next-Week final +12 %, 1M spheres +4 %.
#ifndef RT_LANES_WAVES_PER_EU
#define RT_LANES_WAVES_PER_EU 6
#endif
#ifdef RT_COUNT_WALK // development (tools/work_counts.py): what the walks of a launch touch -- the algorithmic bytes of the roofline
// [0] node steps (one 32-byte node each), [1] leaf tests (one 32-byte leaf each), [2] rays, [3] paths: per lane;
// [4] node steps, [5] leaf phases, [6] shade phases a WAVE ran (whatever the number of lanes on), [7] lanes shaded
__device__ unsigned long long g_walk_counts[16]; // [8..10]: leaf phases that held one / two / three classes of leaves (spheres and rects | boxes | the rest)
#define RT_COUNT(k, cond) n_count[k] += (cond) ? 1u : 0u
#define RT_COUNT_WAVE(k, any) do { const bool any_ = (any); n_count[k] += (lane == 0 && any_) ? 1u : 0u; } while (0)
#else
#define RT_COUNT(k, cond)
#define RT_COUNT_WAVE(k, any)
#endif
#ifdef RT_TIME_PHASES
__device__ unsigned long long g_phase_cycles[16]; // wave cycles in [0] looks + refill + task hand-over, [1] node steps, [2] leaf tests (entry, pop),
// [3] surface, [4] shade; the leaf phase by branch: [5] fetch + decode + chain, [6] sphere, [7] rect, [8] moving sphere, [9] box, [10] medium, [11] join
#endif
template <int TEX, int BOX, int STEPS> // TEX: 0 constant colours only, 1 textures, 2 textures, a noise texture among them (with BOX = 2);
                                        // BOX: 0 / 1 / 2 = the leaf kinds walk_leaf is compiled for
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK)
__attribute__((amdgpu_waves_per_eu(RT_LANES_WAVES_PER_EU, RT_LANES_WAVES_PER_EU)))
void render_fastlanes_kernel(DevScene S, const uint32_t *__restrict__ recs,
                                                                                 rtf::Layout Y, RenderArgs A, float *__restrict__ accum, unsigned *__restrict__ next_task)
{
    __shared__ float tile_acc[WAVES_PER_BLOCK][64 * 3];
    __shared__ __align__(16) rt_camera lcam[WAVES_PER_BLOCK]; // read by the refill from LDS: the camera does not hold 24 SGPRs (one copy per wave: no barrier)
    extern __shared__ int bvh_stacks[]; // [WAVES_PER_BLOCK][A.stack_levels][64], sized by the launch
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (lane < (int)(sizeof(rt_camera) / 4)) reinterpret_cast<uint32_t *>(&lcam[wave])[lane] = reinterpret_cast<const uint32_t *>(&S.cam)[lane];
    int task = blockIdx.x * WAVES_PER_BLOCK + wave; // A.chain: the first one; the counter hands out the ones after the grid's
    if (task >= A.n_tasks) return;

    int tile_x, tile_y, s0, s1, total;
    // A path's place in the image is its pixel word (rng.w = rt_pixel_word(row, column), include/rt_rng.h): it keys the
    // random stream, and when the path ends it says where the radiance goes -- the tile's sums in LDS if the pixel lies in
    // the tile the wave works on, else (A.chain: a path of a tile that has gone to the image already) the image itself.
    uint32_t tile_word = 0u;
    constexpr uint32_t IN_TILE = (7u << 16) | 7u;
    auto open_task = [&](int tk) {
        const int tile = tk / A.n_chunks, chunk = tk - tile * A.n_chunks;
        tile_y = tile / A.tiles_x, tile_x = tile - tile_y * A.tiles_x;
        tile_word = rt_pixel_word((uint32_t)(tile_y * 8), (uint32_t)(tile_x * 8));
        s0 = chunk * A.chunk_samples;
        s1 = min(s0 + A.chunk_samples, A.n_samples);
        total = tk < A.n_tasks ? 64 * (s1 - s0) * A.sub_samples : 0;
    };
    open_task(task);
    const float inv_nx = 1.0f / float(A.nx), inv_ny = 1.0f / float(A.ny);

    float *acc = tile_acc[wave];
    acc[lane] = 0.0f, acc[64 + lane] = 0.0f, acc[128 + lane] = 0.0f;
    int *stack = &bvh_stacks[wave * A.stack_levels * 64 + lane];

    int next = 0;
    bool alive = false;
    Ray r = Ray{v3(0, 0, 0), v3(0, 0, 1), 0.0f};
    V3 T = v3(1, 1, 1);
    Rng rng{0u, 0u};
    int depth = 0;
    rtf::Walk w;
    rtf::walk_start(w, r, RT_FLT_MAX, false);
#ifdef RT_COUNT_WALK
    unsigned n_count[16] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
#endif

#ifdef RT_TIME_PHASES // development (tools/phase_times.py): cycles a wave spends in each phase of the loop
    __shared__ unsigned long long tp_all[WAVES_PER_BLOCK][16];
    unsigned long long *tp = tp_all[wave];
    if (lane < 16) tp[lane] = lane == 15 ? clock64() : 0ull; // [15]: the time of the last stamp (rtf::phase_stamp)
#define RT_PHASE(k) rtf::phase_stamp(tp, k)
#else
#define RT_PHASE(k)
#endif
    for (;;) {
        // ---- lanes whose walk is over: shade, refill, start the next walk ----------------------
        const bool waiting = alive & (w.cur == RT_BVH_NONE);
        const int n_waiting = __popcll(__ballot(waiting));
        const bool walking_any = __any(w.cur != RT_BVH_NONE);
        if (n_waiting >= A.shade_batch || !walking_any) {
            RT_COUNT_WAVE(6, n_waiting > 0);
            RT_COUNT(7, waiting);
            float noise_c = -1.0f;
            if (TEX == 2) {
                // Noise textures, while few lanes ask: the whole wave evaluates one lane's point (rtf::wave_noise_value: 60
                // instructions per point instead of 1400 for the wave), and that lane goes on with a constant colour.  That a lane
                // will ask is a flag on its best hit (RT_LEAF_PRIM_NOISE, rt_bvh.h: no fetch decides it; a box's faces, which are
                // named later, evaluate their own).
                unsigned long long ask = __ballot(waiting & (w.best_prim >= 0) & ((w.best_prim & RT_LEAF_PRIM_NOISE) != 0));
                if (ask != 0ull && __popcll(ask) <= RT_WAVE_NOISE_MAX) {
                    const V3 hp = rtf::fma3(w.best_t, r.d, r.o);
                    do {
                        const int src = __ffsll((long long)ask) - 1;
                        const int prim_s = (int)rtf::rl_u((uint32_t)w.best_prim, src) & RT_LEAF_PRIM_MASK;
                        const float c = rtf::wave_noise_value(S.fshade[prim_s].tex_scale, rtf::rl_f(__float_as_uint(hp.x), src),
                                                              rtf::rl_f(__float_as_uint(hp.y), src), rtf::rl_f(__float_as_uint(hp.z), src), lane);
                        noise_c = lane == src ? c : noise_c;
                        ask &= ask - 1ull;
                    } while (ask != 0ull);
                }
            }
            if (waiting) {
                const bool hit_any = w.best_prim >= 0;
                rtf::Surface sf;
                sf.p = r.o; // a miss ends the path; shade still writes the (dead) ray from it
                RT_PHASE(0);
                if (hit_any) rtf::build_surface_lanes<BOX>(S, r, w.best_t, w.best_prim, sf);
                if (TEX == 2 && noise_c >= 0.0f) sf.color = v3(noise_c, noise_c, noise_c), sf.texture = -1;
                RT_PHASE(3);
                V3 L;
                const bool goes_on = rtf::shade<TEX != 0>(S, recs, Y, hit_any, sf, r, T, depth, A.max_depth, rng, L);
                RT_PHASE(4);
                if (!goes_on) {
                    alive = false;
                    const uint32_t pw = rng.w;
                    const int pix = (int)(((pw >> 16) & 7u) * 8u + (pw & 7u));
                    if (A.chain && ((pw ^ tile_word) & ~IN_TILE) != 0u) { // a path of a tile that has gone to the image already
                        float *dst = accum + 4 * ((size_t)(pw >> 16) * A.nx + (pw & 0xffffu));
                        const float inv_sub = 1.0f / (float)A.sub_samples;
                        if (L.x == L.x && L.x != 0.0f) atomicAdd(dst + 3, L.x * inv_sub);
                        if (L.y == L.y && L.y != 0.0f) atomicAdd(dst + 2, L.y * inv_sub);
                        if (L.z == L.z && L.z != 0.0f) atomicAdd(dst + 1, L.z * inv_sub);
                    } else {
                        if (L.x == L.x && L.x != 0.0f) atomicAdd(&acc[pix], L.x);
                        if (L.y == L.y && L.y != 0.0f) atomicAdd(&acc[64 + pix], L.y);
                        if (L.z == L.z && L.z != 0.0f) atomicAdd(&acc[128 + pix], L.z);
                    }
                }
            }
            bool fresh = waiting & alive; // scattered: the new ray is in r
            // The pool is empty: the tile goes to the image and the wave takes the next task, so its free lanes refill at once
            // instead of idling until the longest of the last paths ends; the paths still on their way find, when they end,
            // that their pixel is not in the wave's tile any more and add to the image themselves.
            if (A.chain && next >= total && task < A.n_tasks) {
                flush_tile(acc, lane, tile_x, tile_y, A, s1 - s0, accum);
                acc[lane] = 0.0f, acc[64 + lane] = 0.0f, acc[128 + lane] = 0.0f;
                unsigned nt = 0u;
                if (lane == 0) nt = atomicAdd(next_task, 1u);
                task = (int)(__builtin_amdgcn_readfirstlane(nt) + gridDim.x * WAVES_PER_BLOCK);
                open_task(task);
                next = 0;
            }
            const unsigned long long dead = __ballot(!alive);
            if (dead != 0ull && next < total) {
                const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(dead >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)dead, 0u));
                const int item = next + rank;
                if (!alive && item < total) {
                    const int pix = item & 63;
                    const int q = item >> 6;
                    const int qs = A.sub_samples == 1 ? q : q / A.sub_samples;
                    const int sample = A.sample_begin + s0 + qs;
                    const int sub = q - qs * A.sub_samples;
                    const int col = tile_x * 8 + (pix & 7), row = tile_y * 8 + (pix >> 3);
                    if (col < A.nx && row < A.ny) {
                        rng.start(rt_path_key(A.seed, (uint32_t)(sample * A.sub_samples + sub)), rt_pixel_word((uint32_t)row, (uint32_t)col));
                        const int j = A.ny - 1 - row;
                        const float u = (float(col) + rng.next()) * inv_nx;
                        const float v = (float(j) + rng.next()) * inv_ny;
                        const rt_camera cam = lcam[wave];
                        const V3 disk = cam.lens_radius == 0.0f ? v3(0, 0, 0) : rtf::rnd_unit_disk(rng); // include/rt_rng.h
                        const float time01 = rng.next();
                        r = rtf::camera_ray(cam, u, v, disk, time01);
                        T = v3(1, 1, 1);
                        depth = 0;
                        alive = true;
                        fresh = true;
                        RT_COUNT(3, true);
                    }
                }
                next += __popcll(dead);
            }
            RT_COUNT(2, fresh);
            if (fresh) rtf::walk_start(w, r, RT_FLT_MAX, true);
            if (!__any(alive)) {
                if (next >= total) break;
                continue; // only out-of-image pixels were drawn; draw again
            }
        }
        RT_PHASE(0); // looks, shade, refill
        // ---- the walks ------------------------------------------------------------------------------
#pragma unroll
        for (int step = 0; step < STEPS; ++step) {
            RT_COUNT(0, w.cur >= 0);
            RT_COUNT_WAVE(4, __any(w.cur >= 0));
            if (w.cur >= 0) rtf::walk_node(S, w, 0.001f, stack);
        }
        RT_PHASE(1); // node steps
        const bool at_leaf = (w.cur < 0) & (w.cur != RT_BVH_NONE);
        if (__popcll(__ballot(at_leaf)) >= A.leaf_batch || !__any(w.cur >= 0)) {
            RT_COUNT(1, at_leaf);
            RT_COUNT_WAVE(5, __any(at_leaf));
#ifdef RT_COUNT_WALK
            {
                int cls = -1;
                if (at_leaf) {
                    const uint32_t lk = S.leaves[~w.cur].kind_chain & 255u;
                    cls = (lk <= RT_PRIM_YZ_RECT || lk == RT_LEAF_MOVING_SPHERE) ? 0 : lk == RT_PRIM_GROUP ? 1 : 2;
                }
                const int n_cls = (__any(cls == 0) ? 1 : 0) + (__any(cls == 1) ? 1 : 0) + (__any(cls == 2) ? 1 : 0);
                RT_COUNT_WAVE(8, n_cls == 1);
                RT_COUNT_WAVE(9, n_cls == 2);
                RT_COUNT_WAVE(10, n_cls == 3);
            }
#endif
            if (at_leaf) { // (a constant_medium draws, src/volume.h:43)
#ifdef RT_TIME_PHASES
                rtf::walk_leaf<BOX>(S, w, r, 0.001f, rng, stack, tp);
#else
                rtf::walk_leaf<BOX>(S, w, r, 0.001f, rng, stack);
#endif
            }
        }
        RT_PHASE(2); // leaf tests
    }

    flush_tile(acc, lane, tile_x, tile_y, A, s1 - s0, accum);
#ifdef RT_TIME_PHASES
    if (lane == 0)
        for (int k = 0; k < 12; ++k) atomicAdd(&g_phase_cycles[k], tp[k]);
#endif
#undef RT_PHASE
#ifdef RT_COUNT_WALK
    for (int k = 0; k < 16; ++k) atomicAdd(&g_walk_counts[k], (unsigned long long)n_count[k]);
#endif
}

// ---------------------------------------------------------------------------
// render_fastlanes_block_kernel: the per-lane walk with the paths of a BLOCK regrouped every few node steps.
// In render_fastlanes_kernel a lane whose walk is over waits, in its wave, until forty-eight lanes do; a node step therefore
// runs with 30 - 39 of a wave's 64 lanes on.  Here the four waves of a block trade paths through LDS after every round of
// a few node steps and leaf tests -- finished walks first, then dead lanes, then the walks still under way -- so the
// finished walks of all four waves are shaded together by whole waves, the dead lanes are refilled together, and the
// other waves hold nothing but walks.  What moves is the path (ray, throughput, stream, depth) and its walk's cursor
// (node, stack height, best hit): 16 dwords.  The stack does NOT move: it belongs to a HOME (one of the block's 256,
// 16 bits per level), and the path carries its home's number, which is why the tree must have fewer than 32768 nodes
// and leaves; 1/d, -o/d and d.d are recomputed after the move.  Work unit: a 16x16 tile x a range of samples, chained
// like the block kernel's.
// ---------------------------------------------------------------------------
// Measured (round 3, profiles/r03_lanes_regroup_experiment.txt): node steps run with 45 - 50 of 64 lanes on instead of 32 - 37, but
// a regroup costs about two node steps per lane and a walk changes state every four or five, and the leaf tests lose their
// batching: random spheres 7.16 -> 6.4, Next-Week final 1.89 -> 1.49 Gpaths/s at the best cadence (6 - 12 rounds), only the
// 4096-sphere field gains (2.27 -> 2.49).  So it is not the default; RT_OPT_LANES_REGROUP = 1 selects it.
#ifndef RT_LANES_REGROUP_DEFAULT
#define RT_LANES_REGROUP_DEFAULT 0
#endif
#ifndef RT_REGROUP_ROUNDS_DEFAULT
#define RT_REGROUP_ROUNDS_DEFAULT 8
#endif
#ifndef RT_LANES_BLOCK_WAVES_PER_EU
#define RT_LANES_BLOCK_WAVES_PER_EU 5
#endif
template <bool TEX, bool BOX>
__global__ __launch_bounds__(256)
__attribute__((amdgpu_waves_per_eu(RT_LANES_BLOCK_WAVES_PER_EU, RT_LANES_BLOCK_WAVES_PER_EU)))
void render_fastlanes_block_kernel(DevScene S, const uint32_t *__restrict__ recs, rtf::Layout Y, RenderArgs A, float *__restrict__ accum,
                                   unsigned *__restrict__ next_task)
{
    constexpr int NW = 4, NT = 64 * NW, TW = 16;
    __shared__ float acc[3 * NT];
    __shared__ uint4 xch[4][NT]; // the path and its walk's cursor in four 16-byte pieces, [piece][slot]
    __shared__ int4 cnt[NW];     // [wave].{x,y,z} = lanes to shade, dead lanes, lanes whose walk is under way
    __shared__ unsigned pool_next;
    __shared__ int task_now;
    __shared__ __align__(16) rt_camera lcam;
    extern __shared__ short home_stacks[]; // [A.stack_levels][NT]: one stack per home
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int task = blockIdx.x;
    if (tid < (int)(sizeof(rt_camera) / 4)) reinterpret_cast<uint32_t *>(&lcam)[tid] = reinterpret_cast<const uint32_t *>(&S.cam)[tid];

    int tile_x, tile_y, s0, s1;
    unsigned total;
    uint32_t tile_word = 0u; // rt_pixel_word of the tile's first pixel (see render_fast_block_kernel)
    constexpr uint32_t IN_TILE = (15u << 16) | (uint32_t)(TW - 1);
    auto open_task = [&](int tk) {
        const int tile = tk / A.n_chunks, chunk = tk - tile * A.n_chunks;
        tile_y = tile / A.tiles_x, tile_x = tile - tile_y * A.tiles_x;
        tile_word = rt_pixel_word((uint32_t)(tile_y * 16), (uint32_t)(tile_x * TW));
        s0 = chunk * A.chunk_samples;
        s1 = min(s0 + A.chunk_samples, A.n_samples);
        total = tk < A.n_tasks ? (unsigned)NT * (unsigned)((s1 - s0) * A.sub_samples) : 0u;
    };
    open_task(task);
    const float inv_nx = 1.0f / float(A.nx), inv_ny = 1.0f / float(A.ny), inv_sub = 1.0f / (float)A.sub_samples;
    auto flush = [&](bool with_atomics) {
        const int col = tile_x * TW + (tid % TW), row = tile_y * 16 + (tid / TW);
        if (col < A.nx && row < A.ny) {
            float *dst = accum + 4 * ((size_t)row * A.nx + col);
            const float v4[4] = {(float)(s1 - s0), acc[2 * NT + tid] * inv_sub, acc[NT + tid] * inv_sub, acc[tid] * inv_sub};
            if (with_atomics) {
                for (int c = 0; c < 4; ++c) atomicAdd(dst + c, v4[c]);
            } else {
                *reinterpret_cast<float4 *>(dst) = make_float4(v4[0], v4[1], v4[2], v4[3]);
            }
        }
    };
    acc[tid] = 0.0f, acc[NT + tid] = 0.0f, acc[2 * NT + tid] = 0.0f;
    if (tid == 0) pool_next = 0u;
    __syncthreads();

    bool alive = false;
    Ray r = Ray{v3(0, 0, 0), v3(0, 0, 1), 0.0f};
    V3 T = v3(1, 1, 1);
    Rng rng{0u, 0u};
    int depth = 0, home = tid, turn = 0;
    rtf::Walk w;
    rtf::walk_start(w, r, RT_FLT_MAX, false);
#ifdef RT_COUNT_WALK
    unsigned n_count[16] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
#endif

    for (;;) {
        // ---- refill: the dead lanes (together after the regroup) take items from the block's pool and start their walks ----
        const unsigned long long dead = __ballot(!alive);
        if (dead != 0ull) {
            unsigned base = 0u;
            if (lane == 0) base = atomicAdd(&pool_next, (unsigned)__popcll(dead)); // may run past `total`; bounded by the host
            base = __builtin_amdgcn_readfirstlane(base);
            if (base < total) {
                const unsigned rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(dead >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)dead, 0u));
                const unsigned item = base + rank;
                if (!alive && item < total) {
                    const int pix = (int)(item % (unsigned)NT);
                    const int q = (int)(item / (unsigned)NT);
                    const int qs = A.sub_samples == 1 ? q : q / A.sub_samples;
                    const int sample = A.sample_begin + s0 + qs;
                    const int sub = q - qs * A.sub_samples;
                    const int col = tile_x * TW + (pix % TW), row = tile_y * 16 + (pix / TW);
                    if (col < A.nx && row < A.ny) {
                        rng.start(rt_path_key(A.seed, (uint32_t)(sample * A.sub_samples + sub)), rt_pixel_word((uint32_t)row, (uint32_t)col));
                        const int j = A.ny - 1 - row;
                        const float u = (float(col) + rng.next()) * inv_nx;
                        const float v = (float(j) + rng.next()) * inv_ny;
                        const rt_camera cam = lcam;
                        const V3 disk = cam.lens_radius == 0.0f ? v3(0, 0, 0) : rtf::rnd_unit_disk(rng); // include/rt_rng.h
                        const float time01 = rng.next();
                        r = rtf::camera_ray(cam, u, v, disk, time01);
                        T = v3(1, 1, 1);
                        depth = 0;
                        alive = true;
                        rtf::walk_start(w, r, RT_FLT_MAX, true);
                        RT_COUNT(3, true);
                        RT_COUNT(2, true);
                    }
                }
            }
        }

        // ---- a round of the walks: A.regroup_rounds x (two node steps, one leaf test) ------------------------------------------
        const rtf::HomeStack16 stack{home_stacks + home, NT};
        for (int rep = 0; rep < A.regroup_rounds; ++rep) {
            if (!__any(alive & (w.cur != RT_BVH_NONE))) break;
#pragma unroll
            for (int step = 0; step < 2; ++step) {
                RT_COUNT(0, w.cur >= 0);
                RT_COUNT_WAVE(4, __any(w.cur >= 0));
                if (w.cur >= 0) rtf::walk_node(S, w, 0.001f, stack);
            }
            const bool at_leaf = (w.cur < 0) & (w.cur != RT_BVH_NONE);
            RT_COUNT(1, at_leaf);
            RT_COUNT_WAVE(5, __any(at_leaf));
#ifdef RT_COUNT_WALK
            {
                int cls = -1;
                if (at_leaf) {
                    const uint32_t lk = S.leaves[~w.cur].kind_chain & 255u;
                    cls = (lk <= RT_PRIM_YZ_RECT || lk == RT_LEAF_MOVING_SPHERE) ? 0 : lk == RT_PRIM_GROUP ? 1 : 2;
                }
                const int n_cls = (__any(cls == 0) ? 1 : 0) + (__any(cls == 1) ? 1 : 0) + (__any(cls == 2) ? 1 : 0);
                RT_COUNT_WAVE(8, n_cls == 1);
                RT_COUNT_WAVE(9, n_cls == 2);
                RT_COUNT_WAVE(10, n_cls == 3);
            }
#endif
            if (at_leaf) rtf::walk_leaf<(BOX ? 2 : 0)>(S, w, r, 0.001f, rng, stack); // (a constant_medium draws, src/volume.h:43)
        }

        // ---- regroup: walks that are over first (to be shaded), then the dead lanes (to be refilled), then the walks under way ----
        const int cls = !alive ? 1 : (w.cur == RT_BVH_NONE ? 0 : 2);
        const unsigned long long b0 = __ballot(cls == 0), b1 = __ballot(cls == 1);
        const unsigned long long mine = cls == 0 ? b0 : cls == 1 ? b1 : ~(b0 | b1);
        const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mine >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mine, 0u));
        if (lane == 0) cnt[wave] = make_int4(__popcll(b0), __popcll(b1), 64 - __popcll(b0 | b1), 0);
        __syncthreads();
        int tot0 = 0, n_dead = 0, low = 0;
#pragma unroll
        for (int w2 = 0; w2 < NW; ++w2) {
            const int4 n = cnt[w2];
            const int same = cls == 0 ? n.x : cls == 1 ? n.y : n.z;
            tot0 += n.x, n_dead += n.y;
            low += w2 < wave ? same : 0;
        }
        const unsigned pool_now = pool_next; // stable between the two barriers: only the refill changes it
        const int dst = (cls == 0 ? 0 : cls == 1 ? tot0 : tot0 + n_dead) + low + rank;
        {
            const uint32_t packed = (uint32_t)depth | ((uint32_t)w.sp << 16) | ((uint32_t)home << 21) | ((uint32_t)cls << 29);
            xch[0][dst] = make_uint4(__float_as_uint(r.o.x), __float_as_uint(r.o.y), __float_as_uint(r.o.z), __float_as_uint(r.d.x));
            xch[1][dst] = make_uint4(__float_as_uint(r.d.y), __float_as_uint(r.d.z), __float_as_uint(r.time), __float_as_uint(T.x));
            xch[2][dst] = make_uint4(__float_as_uint(T.y), __float_as_uint(T.z), rng.x, rng.w);
            xch[3][dst] = make_uint4(packed, (uint32_t)w.cur, __float_as_uint(w.best_t), (uint32_t)w.best_prim);
        }
        __syncthreads();
        if (!A.chain && n_dead == NT && pool_now >= total) break; // the same for every wave of the block
        // (read rotated by one wave more every round, so that every SIMD sees every kind of lane in turn)
        const int slot = (tid + turn) & (NT - 1);
        turn += 64;
        int cls_in;
        {
            const uint4 q0 = xch[0][slot], q1 = xch[1][slot], q2 = xch[2][slot], q3 = xch[3][slot];
            r.o = v3(__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z));
            r.d = v3(__uint_as_float(q0.w), __uint_as_float(q1.x), __uint_as_float(q1.y));
            r.time = __uint_as_float(q1.z);
            T = v3(__uint_as_float(q1.w), __uint_as_float(q2.x), __uint_as_float(q2.y));
            rng.x = q2.z, rng.w = q2.w;
            depth = (int)(q3.x & 0xffffu), home = (int)((q3.x >> 21) & 255u), cls_in = (int)(q3.x >> 29);
            alive = cls_in != 1;
            // the walk's cursor; what depends on the ray alone is made again for the walks under way
            w.sp = (int)((q3.x >> 16) & 31u);
            w.cur = cls_in == 2 ? (int)q3.y : RT_BVH_NONE;
            w.best_t = __uint_as_float(q3.z), w.best_prim = (int)q3.w;
            if (cls_in == 2) rtf::walk_ray_constants(w, r);
        }
        if (A.chain && pool_now >= total) { // the pool is empty (the same finding in every wave of the block)
            if (task >= A.n_tasks) { // nothing left to take
                if (n_dead == NT) break;
            } else {
                // the tile goes to the image and the next task opens; the paths still on their way find, when they end, that
                // their pixel is not in the block's tile any more and add to the image themselves
                flush(true); // (every thread reads and clears the sums of its own pixel only: no barrier in between)
                acc[tid] = 0.0f, acc[NT + tid] = 0.0f, acc[2 * NT + tid] = 0.0f;
                if (tid == 0) {
                    task_now = (int)(atomicAdd(next_task, 1u) + gridDim.x);
                    pool_next = 0u;
                }
                __syncthreads();
                task = task_now;
                open_task(task);
            }
        }

        // ---- shade the walks that are over: whole waves of them ---------------------------------------------------------------
        const bool to_shade = cls_in == 0;
        RT_COUNT_WAVE(6, __any(to_shade));
        RT_COUNT(7, to_shade);
        if (to_shade) {
            const bool hit_any = w.best_prim >= 0;
            rtf::Surface sf;
            sf.p = r.o; // a miss ends the path; shade still writes the (dead) ray from it
            if (hit_any) rtf::build_surface_lanes<(BOX ? 2 : 0)>(S, r, w.best_t, w.best_prim, sf);
            V3 L;
            if (!rtf::shade<TEX>(S, recs, Y, hit_any, sf, r, T, depth, A.max_depth, rng, L)) {
                alive = false;
                const uint32_t pw = rng.w;
                if (A.chain && ((pw ^ tile_word) & ~IN_TILE) != 0u) { // a path of a tile that has gone to the image already
                    float *dst2 = accum + 4 * ((size_t)(pw >> 16) * A.nx + (pw & 0xffffu));
                    if (L.x == L.x && L.x != 0.0f) atomicAdd(dst2 + 3, L.x * inv_sub);
                    if (L.y == L.y && L.y != 0.0f) atomicAdd(dst2 + 2, L.y * inv_sub);
                    if (L.z == L.z && L.z != 0.0f) atomicAdd(dst2 + 1, L.z * inv_sub);
                } else {
                    const int pix = (int)(((pw >> 16) & 15u) * (uint32_t)TW + (pw & (uint32_t)(TW - 1)));
                    if (L.x == L.x && L.x != 0.0f) atomicAdd(&acc[pix], L.x);
                    if (L.y == L.y && L.y != 0.0f) atomicAdd(&acc[NT + pix], L.y);
                    if (L.z == L.z && L.z != 0.0f) atomicAdd(&acc[2 * NT + pix], L.z);
                }
            } else {
                rtf::walk_start(w, r, RT_FLT_MAX, true); // scattered: the new ray is in r
                RT_COUNT(2, true);
            }
        }
    }

    if (!A.chain) flush(A.atomic_out != 0);
    else flush(true); // the last tile (every other one went out when the block moved on)
#ifdef RT_COUNT_WALK
    for (int k = 0; k < 16; ++k) atomicAdd(&g_walk_counts[k], (unsigned long long)n_count[k]);
#endif
}

__global__ __launch_bounds__(256) void trace_fastlanes_kernel(DevScene S, int n, const float *__restrict__ rays, float t_min, float t_max,
                                                              uint32_t seed, int stack_levels, rt_hit *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool alive = i < n;
    const int k = alive ? i : 0;
    const float *f = rays + 7 * (size_t)k;
    Ray r{v3(f[0], f[1], f[2]), v3(f[3], f[4], f[5]), f[6]};
    Rng rng{rt_path_key(seed, 0u), (uint32_t)k};
    extern __shared__ int bvh_stacks[]; // [4][stack_levels][64]
    int *stack = &bvh_stacks[__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * stack_levels * 64 + (threadIdx.x & 63)];
    float t;
    int prim;
    const bool any = rtf::hit_bvh(S, r, t_min, t_max, alive, rng, t, prim, stack);
    if (!alive) return;
    rtf::Surface sf;
    sf.p = sf.n = v3(0, 0, 0), sf.u = sf.v = 0.0f;
    rt_hit o;
    sf.face = prim;
    if (any) rtf::build_surface_lanes<2>(S, r, t, prim, sf);
    o.prim_id = any ? S.fprims[sf.face].user_id : -1; // a box names the face rect that was hit
    o.material = any ? S.fprims[sf.face].material : -1;
    o.t = any ? t : 0.0f, o.u = sf.u, o.v = sf.v;
    o.p[0] = sf.p.x, o.p[1] = sf.p.y, o.p[2] = sf.p.z;
    o.normal[0] = sf.n.x, o.normal[1] = sf.n.y, o.normal[2] = sf.n.z;
    o.reserved = 0.0f;
    out[i] = o;
}

// ---------------------------------------------------------------------------
// rt_debug_eval: known-answer evaluation of the sampling / shading functions on the device (include/rt_amd.h).  Not on any
// render path: one thread per item, 16 floats in, 16 floats out, the functions called exactly as the kernels call them,
// both families (rt:: exact, rtf:: fast) side by side, so that a failing image test can be traced to the function.
// ---------------------------------------------------------------------------
__device__ __forceinline__ Rng eval_rng(const float *in, int at) { return Rng{__float_as_uint(in[at]), __float_as_uint(in[at + 1])}; }

__global__ __launch_bounds__(256) void debug_eval_kernel(DevScene S, const uint32_t *__restrict__ recs, rtf::Layout Y, int kind, int n,
                                                         const float *__restrict__ in_all, float *__restrict__ out_all)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *in = in_all + 16 * (size_t)i;
    float *out = out_all + 16 * (size_t)i;
    float o16[16];
    for (int k = 0; k < 16; ++k) o16[k] = 0.0f;
    if (kind == RT_EVAL_LIGHT_PDF) {
        // in: kind (0 xz_rect, 1 sphere), o, v, f[5], normal  out: rt::light_pdf_value, rt::cosine_pdf_value, the same two of rtf::
        const bool rect = in[0] == 0.0f;
        rt_prim pr{};
        pr.kind = rect ? RT_PRIM_XZ_RECT : RT_PRIM_SPHERE;
        for (int k = 0; k < 5; ++k) pr.f[k] = in[7 + k];
        const V3 o = v3(in[1], in[2], in[3]), v = v3(in[4], in[5], in[6]), nrm = v3(in[12], in[13], in[14]);
        o16[0] = light_pdf_value(pr, o, v);
        Onb uvw;
        uvw.build_from_w(nrm);
        o16[1] = cosine_pdf_value(uvw, v);
        rtf::SRec c{}; // the light record rt_records.cpp (emit_lights) writes for this primitive
        for (int k = 0; k < 5; ++k) c.d[k] = __float_as_uint(pr.f[k]);
        const float area = (pr.f[1] - pr.f[0]) * (pr.f[3] - pr.f[2]);
        c.d[5] = __float_as_uint(area), c.d[6] = __float_as_uint(1.0f / area);
        c.d[7] = rect ? RT_REC_LIGHT_RECT : RT_REC_LIGHT_SPHERE;
        o16[2] = rtf::light_rec_pdf(c, o, rtf::funit(v));
        o16[3] = rtf::cos_over_pi(rtf::funit(nrm), rtf::funit(v));
    } else if (kind == RT_EVAL_ONB) {
        // in: n  out: rt:: u v w, unit(n), sphere_uv(unit(n))   (the layout of the oracle's orc_kat_onb_uv)
        const V3 nrm = v3(in[0], in[1], in[2]);
        Onb b;
        b.build_from_w(nrm);
        const V3 p = unit(nrm);
        float su, sv;
        sphere_uv(p, su, sv);
        const float v[14] = {b.u.x, b.u.y, b.u.z, b.v.x, b.v.y, b.v.z, b.w.x, b.w.y, b.w.z, p.x, p.y, p.z, su, sv};
        for (int k = 0; k < 14; ++k) o16[k] = v[k];
    } else if (kind == RT_EVAL_ONB_FAST) {
        // in: n, a  out: rtf:: u v w, funit(n), local(a)
        const V3 nrm = v3(in[0], in[1], in[2]);
        rtf::Onb b;
        b.build_from_w(nrm);
        const V3 p = rtf::funit(nrm), l = b.local(v3(in[3], in[4], in[5]));
        const float v[15] = {b.u.x, b.u.y, b.u.z, b.v.x, b.v.y, b.v.z, b.w.x, b.w.y, b.w.z, p.x, p.y, p.z, l.x, l.y, l.z};
        for (int k = 0; k < 15; ++k) o16[k] = v[k];
    } else if (kind == RT_EVAL_OPTICS) {
        // in: v, n, ni_over_nt, cosine, ref_idx  out: reflect, ok, refracted, schlick   (orc_kat_optics)
        const V3 a = v3(in[0], in[1], in[2]), nn = v3(in[3], in[4], in[5]);
        const V3 rf = reflect(a, nn);
        V3 rr = v3(0, 0, 0);
        const bool ok = refract(a, nn, in[6], rr);
        const float v[8] = {rf.x, rf.y, rf.z, ok ? 1.0f : 0.0f, ok ? rr.x : 0.0f, ok ? rr.y : 0.0f, ok ? rr.z : 0.0f, schlick(in[7], in[8])};
        for (int k = 0; k < 8; ++k) o16[k] = v[k];
    } else if (kind == RT_EVAL_PERLIN) {
        // in: p  out: rt:: noise, turb; rtf:: noise, turb
        const V3 p = v3(in[0], in[1], in[2]);
        o16[0] = perlin_noise(S, p), o16[1] = perlin_turb(S, p);
        o16[2] = rtf::fast_perlin_noise(p), o16[3] = rtf::fast_perlin_turb(p);
        o16[4] = rtf::fast_perlin_noise<true>(p), o16[5] = rtf::fast_perlin_turb<true>(p); // the corners written out (record-table kernel)
    } else if (kind == RT_EVAL_TEXTURE) {
        // in: texture index of the scene's table, u, v, p  out: rt::tex_value, rtf::fast_tex_value
        const int idx = (int)in[0];
        const V3 p = v3(in[3], in[4], in[5]);
        const V3 a = tex_value(S, idx, in[1], in[2], p), b = rtf::fast_tex_value(S, idx, in[1], in[2], p);
        o16[0] = a.x, o16[1] = a.y, o16[2] = a.z, o16[3] = b.x, o16[4] = b.y, o16[5] = b.z;
    } else if (kind == RT_EVAL_SAMPLERS) {
        // in: key, pixel (bit patterns), o  out: rt:: lights_random(o), random_cosine_direction, random_in_unit_sphere,
        // random_in_unit_disk (x, y), each from the start of the stream (key, pixel); rtf:: rnd_unit_sphere, rnd_unit_disk (x, y)
        const V3 o = v3(in[2], in[3], in[4]);
        Rng r0 = eval_rng(in, 0), r1 = r0, r2 = r0, r3 = r0, r4 = r0, r5 = r0;
        const V3 a = lights_random(S, o, r0), b = random_cosine_direction(r1), c = random_in_unit_sphere(r2), d = random_in_unit_disk(r3);
        const V3 e = rtf::rnd_unit_sphere(r4), f = rtf::rnd_unit_disk(r5);
        const float v[16] = {a.x, a.y, a.z, b.x, b.y, b.z, c.x, c.y, c.z, d.x, d.y, e.x, e.y, e.z, f.x, f.y};
        for (int k = 0; k < 16; ++k) o16[k] = v[k];
    }
    for (int k = 0; k < 16; ++k) out[k] = o16[k];
}

// RT_EVAL_TEXTURE, second pass: out[6] = rtf::wave_noise_value of the item's point where its texture is a noise texture (what
// the per-lane render kernel calls when few lanes of a shade phase ask).  Whole waves: lanes past n stay and take part.
__global__ __launch_bounds__(256) void debug_wave_noise_kernel(DevScene S, int n, const float *__restrict__ in_all, float *__restrict__ out_all)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
    const float *in = in_all + 16 * (size_t)(i < n ? i : 0);
    const V3 p = v3(in[3], in[4], in[5]);
    const rt_texture t = S.textures[(int)in[0]];
    unsigned long long ask = __ballot((i < n) & (t.kind == RT_TEX_NOISE));
    float mine = 0.0f;
    while (ask != 0ull) {
        const int src = __ffsll((long long)ask) - 1;
        const float c = rtf::wave_noise_value(rtf::rl_f(__float_as_uint(t.scale), src), rtf::rl_f(__float_as_uint(p.x), src),
                                              rtf::rl_f(__float_as_uint(p.y), src), rtf::rl_f(__float_as_uint(p.z), src), lane);
        mine = lane == src ? c : mine;
        ask &= ask - 1ull;
    }
    if (i < n) out_all[16 * (size_t)i + 6] = mine;
}

// RT_EVAL_BOUNCE: world->hit() and ONE bounce of color() for the ray in[0..6] on the stream (key, pixel) = in[7], in[8].
// out: hit (0/1), material kind, continues (0/1), scattered direction, throughput factor (T = 1 before the bounce; a path that
// ends reports its radiance there), t.   EXACT family: the kernels of RT_PATH_EXACT.
template <bool LANES>
__global__ __launch_bounds__(256) void debug_bounce_exact_kernel(DevScene S, int n, const float *__restrict__ in_all, float *__restrict__ out_all)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool alive = i < n;
    const float *in = in_all + 16 * (size_t)(alive ? i : 0);
    Ray r{v3(in[0], in[1], in[2]), v3(in[3], in[4], in[5]), in[6]};
    Rng rng = eval_rng(in, 7);
    int *stack = LaneStacks<LANES>::get(__builtin_amdgcn_readfirstlane(threadIdx.x >> 6), threadIdx.x & 63);
    Hit h;
    const bool any = closest_hit<LANES>(S, r, 0.001f, RT_FLT_MAX, alive, rng, h, stack);
    if (!alive) return;
    V3 T = v3(1, 1, 1), L = v3(0, 0, 0);
    int depth = 0;
    const int mat_kind = any ? S.materials[h.material].kind : -1;
    const bool go = shade(S, any, h, r, T, depth, 50, rng, L);
    float *out = out_all + 16 * (size_t)i;
    const V3 w = go ? T : L;
    const float v[16] = {any ? 1.0f : 0.0f, (float)mat_kind, go ? 1.0f : 0.0f, go ? r.d.x : 0.0f, go ? r.d.y : 0.0f, go ? r.d.z : 0.0f, w.x, w.y, w.z,
                         any ? h.t : 0.0f, 0, 0, 0, 0, 0, 0};
    for (int k = 0; k < 16; ++k) out[k] = v[k];
}
// the same on the record table (the scan, surface and shade of render_fast_kernel / render_fast_block_kernel)
template <bool FLAT>
__global__ __launch_bounds__(256) void debug_bounce_fast_kernel(DevScene S, const uint32_t *__restrict__ recs, rtf::Layout Y, int n,
                                                                const float *__restrict__ in_all, float *__restrict__ out_all)
{
    __shared__ uint32_t lrecs[(RT_REC_MAX + 1) * 8];
    rtf::stage_records(recs, lrecs, Y.n_stream, threadIdx.x, 256);
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool alive = i < n;
    const float *in = in_all + 16 * (size_t)(alive ? i : 0);
    rtf::Regs R;
    load_records(recs, threadIdx.x & 63, R);
    Ray r{v3(in[0], in[1], in[2]), v3(in[3], in[4], in[5]), in[6]};
    Rng rng = eval_rng(in, 7);
    float t;
    int rec;
    rtf::scan<FLAT>(recs, lrecs, Y, r, 0.001f, RT_FLT_MAX, alive, t, rec);
    rtf::Surface sf;
    rtf::build_surface(R, Y, r, t, rec, sf);
    if (!alive) return;
    V3 T = v3(1, 1, 1), L = v3(0, 0, 0);
    int depth = 0;
    const bool any = rec >= 0;
    const bool go = rtf::shade<true>(S, recs, Y, any, sf, r, T, depth, 50, rng, L);
    float *out = out_all + 16 * (size_t)i;
    const V3 w = go ? T : L;
    const float v[16] = {any ? 1.0f : 0.0f, any ? (float)sf.mat_kind : -1.0f, go ? 1.0f : 0.0f, go ? r.d.x : 0.0f, go ? r.d.y : 0.0f, go ? r.d.z : 0.0f,
                         w.x, w.y, w.z, any ? t : 0.0f, 0, 0, 0, 0, 0, 0};
    for (int k = 0; k < 16; ++k) out[k] = v[k];
}
// and on the library's own tree (the walk, surface and shade of render_fastlanes_kernel)
__global__ __launch_bounds__(256) void debug_bounce_lanes_kernel(DevScene S, const uint32_t *__restrict__ recs, rtf::Layout Y, int n,
                                                                 int stack_levels, const float *__restrict__ in_all, float *__restrict__ out_all)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool alive = i < n;
    const float *in = in_all + 16 * (size_t)(alive ? i : 0);
    Ray r{v3(in[0], in[1], in[2]), v3(in[3], in[4], in[5]), in[6]};
    Rng rng = eval_rng(in, 7);
    extern __shared__ int bvh_stacks[]; // [4][stack_levels][64]
    int *stack = &bvh_stacks[__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * stack_levels * 64 + (threadIdx.x & 63)];
    float t;
    int prim;
    const bool any = rtf::hit_bvh(S, r, 0.001f, RT_FLT_MAX, alive, rng, t, prim, stack);
    if (!alive) return;
    rtf::Surface sf;
    sf.p = r.o, sf.n = v3(0, 0, 0), sf.u = sf.v = 0.0f, sf.face = prim, sf.mat_kind = 0u, sf.texture = -1, sf.p0 = 0.0f, sf.color = v3(0, 0, 0);
    if (any) rtf::build_surface_lanes<2>(S, r, t, prim, sf);
    V3 T = v3(1, 1, 1), L = v3(0, 0, 0);
    int depth = 0;
    const bool go = rtf::shade<true>(S, recs, Y, any, sf, r, T, depth, 50, rng, L);
    float *out = out_all + 16 * (size_t)i;
    const V3 w = go ? T : L;
    const float v[16] = {any ? 1.0f : 0.0f, any ? (float)sf.mat_kind : -1.0f, go ? 1.0f : 0.0f, go ? r.d.x : 0.0f, go ? r.d.y : 0.0f, go ? r.d.z : 0.0f,
                         w.x, w.y, w.z, any ? t : 0.0f, 0, 0, 0, 0, 0, 0};
    for (int k = 0; k < 16; ++k) out[k] = v[k];
}

} // namespace

// ===========================================================================
// host side of the C-ABI
// ===========================================================================
struct rt_scene {
    int device = 0;
    void *blob = nullptr; // one allocation holding every table
    DevScene dev;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    // register-resident form (rt_records.h), when the scene fits
    bool fast_ok = false;
    bool fast_textured = false; // some material's texture is not a constant
    bool lanes_noise = false;   // per-lane tree: some leaf carries RT_LEAF_PRIM_NOISE (render_fastlanes_kernel's TEX = 2)
    bool fast_flat = false;     // the record stream holds no box records and is stored in runs
    bool lanes_ok = false;      // the scene does not fit the table but the per-lane fast path (rt_fastlanes.hpp) can run it
    bool lanes_boxes = false;   // its tree holds leaves other than static spheres and rects: boxes (rt_box.h), media, moving spheres
    int lanes_leaf_kinds = 0;   // 0: static spheres and rects in world space only; 1: + moving spheres; 2: anything (render_fastlanes_kernel's BOX)
    int bvh_levels = 1;         // inner levels of that tree (rt_bvh_build)
    bool lanes_small = false;   // that tree has fewer than 32767 nodes and leaves: its references fit the 16-bit home stacks of render_fastlanes_block_kernel
    const char *fast_why = "";
    rtf::Layout layout{0, 0, 0, 0, {0, 0, 0, 0}, 0, 0, 0u, 0.0f};
    uint32_t *d_records = nullptr; // 64 x 8 dwords, then 64 user ids and 64 x 6 face ids (inside blob)
    int32_t *d_rec_user = nullptr;
    unsigned *d_task_counters = nullptr; // inside blob: 64 task counters of the chained kernels, used in turn (each zeroed
                                         // before its launch), so launches on different streams do not share one
    std::atomic<unsigned> launch_seq{0};
    unsigned char *scratch = nullptr; // this device's copy of the static staging area (rt_trace_rays_noalloc)
    int n_cus = 256;
    int render_path = RT_PATH_AUTO, trace_path = RT_PATH_AUTO;
    int last_render_path = RT_PATH_EXACT;
    // rt_scene_set_option: tuning / test switches of rt_render (0 = the library's choice)
    int opt_block_kernel = 1, opt_chain = 1, opt_target_tasks = 0, opt_paths_per_lane = 0, opt_node_steps = 0, opt_chain_blocks_per_cu = 0;
    int opt_shade_batch = 0, opt_leaf_batch = 0;
    int opt_block_waves = 0; // RT_OPT_BLOCK_WAVES: waves of the block kernel's block (0: by the size of the launch)
    int opt_lanes_regroup = -1, opt_regroup_rounds = 0; // RT_OPT_LANES_REGROUP (-1: the library's choice), RT_OPT_REGROUP_ROUNDS
};

#define RT_HIP(call)                                                                                                   \
    do {                                                                                                               \
        hipError_t e_ = (call);                                                                                        \
        if (e_ != hipSuccess) return rt_fail(RT_ERR_NO_DEVICE, "%s: %s", #call, hipGetErrorString(e_));                \
    } while (0)

// ---------------------------------------------------------------------------
// rt_trace_rays_noalloc: rt_trace_rays for callers that cannot allocate.  The device staging of the rays and of the
// hits (what rt_trace_rays takes from hipMalloc on every call) lives in `scratch`, a file-scope device array that every
// GPU the library runs on holds until the process ends.  In host code the NAME `scratch` is a stand-in, not a device
// address: the address on each device comes from hipGetSymbolAddress, asked once per device with that device current
// (rt_scene_create does it, so the entry point itself only looks it up), and only that pointer is ever handed to a
// copy or a kernel.
// ---------------------------------------------------------------------------
constexpr size_t NOALLOC_RAYS_BYTES = (size_t)RT_TRACE_NOALLOC_MAX_RAYS * 28;
constexpr size_t NOALLOC_HITS_OFFSET = (NOALLOC_RAYS_BYTES + 255) & ~(size_t)255;
constexpr size_t NOALLOC_BYTES = NOALLOC_HITS_OFFSET + (size_t)RT_TRACE_NOALLOC_MAX_RAYS * sizeof(rt_hit);
static __device__ __align__(256) unsigned char scratch[NOALLOC_BYTES]; // 4.75 MiB per GPU

constexpr int NOALLOC_MAX_DEVICES = 64;
static std::mutex g_scratch_mutex;
static void *g_scratch_ptr[NOALLOC_MAX_DEVICES]; // [device]: that device's copy of `scratch`
static std::mutex g_scratch_in_use[NOALLOC_MAX_DEVICES]; // [device]: held from copy-in to copy-out of one call

// `device` must be current.  Asks the runtime once per device.
static int scratch_of_device(int device, void **out)
{
    if (device < 0 || device >= NOALLOC_MAX_DEVICES) return rt_fail(RT_ERR_INVALID, "device %d: the static staging area serves devices 0..%d", device, NOALLOC_MAX_DEVICES - 1);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void *ptr = nullptr;
        RT_HIP(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        if (!ptr) return rt_fail(RT_ERR_NO_DEVICE, "hipGetSymbolAddress(scratch) returned no address on device %d", device);
        g_scratch_ptr[device] = ptr;
    }
    *out = g_scratch_ptr[device];
    return RT_OK;
}

static int validate(const rt_scene_desc *d)
{
    if (!d) return rt_fail(RT_ERR_INVALID, "rt_scene_create: null description");
    if (d->struct_size != sizeof(rt_scene_desc) || d->abi_version != RT_ABI_VERSION)
        return rt_fail(RT_ERR_INVALID, "rt_scene_create: description is for ABI %u size %u, library has ABI %d size %zu",
                       d->abi_version, d->struct_size, RT_ABI_VERSION, sizeof(rt_scene_desc));
    if (d->n_entries < 1 || !d->entries || d->n_prims < 1 || !d->prims || d->n_chains < 1 || !d->chains ||
        d->n_materials < 1 || !d->materials || d->n_textures < 0 || d->n_xforms < 0 || d->n_lights < 0)
        return rt_fail(RT_ERR_INVALID, "rt_scene_create: empty or missing table");
    if (d->n_chains >= (1 << 23)) return rt_fail(RT_ERR_INVALID, "rt_scene_create: too many chains");
    if (d->chains[0].count != 0) return rt_fail(RT_ERR_INVALID, "rt_scene_create: chain 0 must be the identity");
    for (int i = 0; i < d->n_chains; ++i) {
        const rt_chain &c = d->chains[i];
        if (c.count < 0 || c.first < 0 || (c.count > 0 && (int64_t)c.first + c.count > d->n_xforms))
            return rt_fail(RT_ERR_INVALID, "rt_scene_create: chain %d out of range", i);
    }
    for (int i = 0; i < d->n_xforms; ++i)
        if (d->xforms[i].kind != RT_XFORM_TRANSLATE && d->xforms[i].kind != RT_XFORM_ROTATE_Y)
            return rt_fail(RT_ERR_INVALID, "rt_scene_create: xform %d has unknown kind", i);
    for (int i = 0; i < d->n_textures; ++i) {
        const rt_texture &t = d->textures[i];
        if (t.kind < RT_TEX_CONSTANT || t.kind > RT_TEX_IMAGE) return rt_fail(RT_ERR_INVALID, "rt_scene_create: texture %d has unknown kind", i);
        if (t.kind == RT_TEX_CHECKER && (t.a < 0 || t.a >= d->n_textures || t.b < 0 || t.b >= d->n_textures))
            return rt_fail(RT_ERR_INVALID, "rt_scene_create: checker texture %d refers outside the table", i);
        if (t.kind == RT_TEX_IMAGE && (t.nx < 1 || t.ny < 1 || t.offset < 0 || !d->texels ||
                                       t.offset + (int64_t)3 * t.nx * t.ny > d->n_texel_bytes))
            return rt_fail(RT_ERR_INVALID, "rt_scene_create: image texture %d outside the texel blob", i);
    }
    for (int i = 0; i < d->n_materials; ++i) {
        const rt_material &m = d->materials[i];
        if (m.kind < RT_MAT_LAMBERTIAN || m.kind > RT_MAT_ISOTROPIC) return rt_fail(RT_ERR_INVALID, "rt_scene_create: material %d has unknown kind", i);
        if (m.kind != RT_MAT_DIELECTRIC && (m.texture < 0 || m.texture >= d->n_textures))
            return rt_fail(RT_ERR_INVALID, "rt_scene_create: material %d has no texture", i);
    }
    auto plain = [](int k) { return k >= RT_PRIM_SPHERE && k <= RT_PRIM_YZ_RECT; };
    for (int i = 0; i < d->n_prims; ++i) {
        const rt_prim &p = d->prims[i];
        if (p.kind < RT_PRIM_SPHERE || p.kind > RT_PRIM_MEDIUM) return rt_fail(RT_ERR_INVALID, "rt_scene_create: primitive %d has unknown kind", i);
        if (p.chain < 0 || p.chain >= d->n_chains) return rt_fail(RT_ERR_INVALID, "rt_scene_create: primitive %d chain out of range", i);
        if (p.kind == RT_PRIM_GROUP) {
            if (p.count < 1 || p.first < 0 || (int64_t)p.first + p.count > d->n_prims)
                return rt_fail(RT_ERR_INVALID, "rt_scene_create: group %d children out of range", i);
            for (int k = 0; k < p.count; ++k) {
                const rt_prim &c = d->prims[p.first + k];
                if (!plain(c.kind) || c.material < 0 || c.material >= d->n_materials)
                    return rt_fail(RT_ERR_INVALID, "rt_scene_create: group %d child %d must be a plain primitive with a material", i, k);
                // a list is one object under one instance chain (src/hitable_list.h, src/transforms.h); the render paths
                // apply the GROUP's chain to its members
                if (c.chain != p.chain)
                    return rt_fail(RT_ERR_INVALID, "rt_scene_create: group %d child %d carries chain %d, its group chain %d", i, k, c.chain, p.chain);
            }
        } else {
            if (p.material < 0 || p.material >= d->n_materials)
                return rt_fail(RT_ERR_INVALID, "rt_scene_create: primitive %d material out of range", i);
            if (p.kind == RT_PRIM_MEDIUM) {
                if (p.first < 0 || p.first >= d->n_prims || d->prims[p.first].kind == RT_PRIM_MEDIUM)
                    return rt_fail(RT_ERR_INVALID, "rt_scene_create: medium %d has no usable boundary", i);
                if (!(p.f[0] > 0.0f)) return rt_fail(RT_ERR_INVALID, "rt_scene_create: medium %d density must be positive", i);
            }
        }
    }
    for (int i = 0; i < d->n_entries; ++i) {
        const rt_entry &e = d->entries[i];
        int kind = e.b & 255, chain = e.b >> 8;
        if (chain < 0 || chain >= d->n_chains) return rt_fail(RT_ERR_INVALID, "rt_scene_create: entry %d chain out of range", i);
        if (kind == RT_ENTRY_INNER) {
            // the skip link must move forward or the scan would never end
            if (e.a <= i || e.a > d->n_entries) return rt_fail(RT_ERR_INVALID, "rt_scene_create: entry %d skip link %d is not ahead", i, e.a);
        } else if (kind == RT_ENTRY_LEAF) {
            if (e.a < 0 || e.a >= d->n_prims) return rt_fail(RT_ERR_INVALID, "rt_scene_create: entry %d primitive out of range", i);
        } else {
            return rt_fail(RT_ERR_INVALID, "rt_scene_create: entry %d has unknown kind", i);
        }
    }
    for (int i = 0; i < d->n_lights; ++i) {
        int l = d->lights[i];
        if (l >= d->n_prims) return rt_fail(RT_ERR_INVALID, "rt_scene_create: light %d out of range", i);
        if (l < -1) return rt_fail(RT_ERR_INVALID, "rt_scene_create: light %d is %d (a primitive index, or -1)", i, l);
        if (l >= 0 && d->prims[l].kind != RT_PRIM_XZ_RECT && d->prims[l].kind != RT_PRIM_SPHERE)
            return rt_fail(RT_ERR_INVALID, "rt_scene_create: light %d must be an xz_rect or a sphere (or -1)", i);
        // translate / rotate_y do not forward pdf_value / random (src/hitable.h:19-20), so a member under an instance chain
        // samples as "no override": it must be given as -1 (rt_flatten does), its coordinates are never used as world space
        if (l >= 0 && d->prims[l].chain != 0)
            return rt_fail(RT_ERR_INVALID, "rt_scene_create: light %d is under an instance chain; pass -1 for it (src/hitable.h:19-20)", i);
    }
    bool lambert = false;
    for (int i = 0; i < d->n_materials; ++i) lambert |= d->materials[i].kind == RT_MAT_LAMBERTIAN;
    if (lambert && d->n_lights < 1)
        return rt_fail(RT_ERR_INVALID, "rt_scene_create: lambertian surfaces need a non-empty important_hitables list");
    return RT_OK;
}

extern "C" int rt_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return rt_fail(RT_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}

extern "C" int rt_scene_create(const rt_scene_desc *d, int device, rt_scene **out) { return rt_scene_create_flags(d, device, 0u, out); }

extern "C" int rt_scene_create_flags(const rt_scene_desc *d, int device, uint32_t flags, rt_scene **out)
{
    if (!out) return rt_fail(RT_ERR_INVALID, "rt_scene_create: null output");
    if (flags & ~RT_CREATE_KEEP_BOX_RECORDS) return rt_fail(RT_ERR_INVALID, "rt_scene_create_flags: unknown flags 0x%x", flags);
    *out = nullptr;
    int st = validate(d);
    if (st != RT_OK) return st;
    int ndev = rt_device_count();
    if (ndev < 0) return ndev;
    if (device < 0 || device >= ndev) return rt_fail(RT_ERR_NO_DEVICE, "rt_scene_create: device %d of %d", device, ndev);
    RT_HIP(hipSetDevice(device));
    // this device's copy of the static staging area, asked for once per device.  Only rt_trace_rays_noalloc needs it: a
    // failure here (a device ordinal beyond the table) is that entry point's to report, not the scene's
    void *scratch_ptr = nullptr;
    if (scratch_of_device(device, &scratch_ptr) != RT_OK) scratch_ptr = nullptr;

    // one blob, every table 256-byte aligned
    struct part {
        const void *src;
        size_t bytes, off;
    };
    rt_record_table tab;
    const char *why = "";
    const bool fast_ok = rt_records_build(d, (flags & RT_CREATE_KEEP_BOX_RECORDS) ? 1 : 0, &tab, &why) != 0;
    std::vector<int32_t> order;
    int tree_depth = 0;
    const bool ordered = rt_order_build(d, order, &tree_depth);
    // no record table: the per-lane fast path, with its own tree (rt_bvh.h) and the light records only
    std::vector<rt_bvh_node> bvh;
    std::vector<rt_leaf> leaves;
    std::vector<rt_prim> fprims;
    int bvh_depth = 0;
    const bool lanes_ok = !fast_ok && rt_bvh_build(d, bvh, leaves, fprims, &bvh_depth) && rt_records_build_lights(d, &tab) != 0;
    if (!lanes_ok) bvh.clear(), leaves.clear(), fprims.clear();
    // per primitive of that table, what shading reads: material and texture folded into one record (rt_bvh.h)
    std::vector<rt_shade_rec> fshade(fprims.size());
    for (size_t i = 0; i < fprims.size(); ++i) {
        rt_shade_rec &r = fshade[i];
        memset(&r, 0, sizeof r);
        r.texture = -1, r.color[0] = r.color[1] = r.color[2] = 1.0f;
        const int mi = fprims[i].material;
        if (mi < 0 || mi >= d->n_materials) continue; // a list: its members carry the materials
        const rt_material &m = d->materials[mi];
        r.mat_kind = (uint32_t)m.kind, r.p0 = m.p0;
        if (m.texture >= 0) {
            const rt_texture &tx = d->textures[m.texture];
            if (tx.kind == RT_TEX_CONSTANT) r.color[0] = tx.color[0], r.color[1] = tx.color[1], r.color[2] = tx.color[2];
            else r.texture = m.texture, r.tex_kind = (uint32_t)tx.kind, r.tex_scale = tx.scale;
        }
    }
    part parts[15] = {
        {d->entries, sizeof(rt_entry) * (size_t)d->n_entries, 0},
        {d->prims, sizeof(rt_prim) * (size_t)d->n_prims, 0},
        {d->xforms, sizeof(rt_xform) * (size_t)d->n_xforms, 0},
        {d->chains, sizeof(rt_chain) * (size_t)d->n_chains, 0},
        {d->materials, sizeof(rt_material) * (size_t)d->n_materials, 0},
        {d->textures, sizeof(rt_texture) * (size_t)d->n_textures, 0},
        {d->texels, (size_t)d->n_texel_bytes, 0},
        {d->lights, sizeof(int32_t) * (size_t)d->n_lights, 0},
        {tab.rec, sizeof tab.rec, 0},
        {tab.user_id, sizeof tab.user_id + sizeof tab.face_user, 0}, // face_user follows user_id (asserted below)
        {order.data(), sizeof(int32_t) * order.size(), 0},
        {bvh.data(), sizeof(rt_bvh_node) * bvh.size(), 0},
        {fprims.data(), sizeof(rt_prim) * fprims.size(), 0},
        {leaves.data(), sizeof(rt_leaf) * leaves.size(), 0},
        {fshade.data(), sizeof(rt_shade_rec) * fshade.size(), 0},
    };
    size_t total = 0;
    for (part &p : parts) {
        p.off = total;
        total += (p.bytes + 255) & ~(size_t)255;
        total += 256; // a table may be empty; keep pointers distinct and in bounds
    }
    const size_t counter_off = total; // the task counter of the chained block kernel
    total += 256;
    std::vector<uint8_t> staging(total, 0);
    for (part &p : parts)
        if (p.bytes) memcpy(staging.data() + p.off, p.src, p.bytes);

    rt_scene *s = new (std::nothrow) rt_scene;
    if (!s) return rt_fail(RT_ERR_NOMEM, "rt_scene_create: out of host memory");
    s->device = device;
    s->scratch = static_cast<unsigned char *>(scratch_ptr);
    hipError_t e = hipMalloc(&s->blob, total);
    if (e != hipSuccess) {
        delete s;
        return rt_fail(RT_ERR_NOMEM, "hipMalloc(%zu): %s", total, hipGetErrorString(e));
    }
    e = hipMemcpy(s->blob, staging.data(), total, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipEventCreate(&s->ev0);
    if (e == hipSuccess) e = hipEventCreate(&s->ev1);
    if (e != hipSuccess) {
        rt_fail(RT_ERR_NO_DEVICE, "rt_scene_create: %s", hipGetErrorString(e));
        rt_scene_destroy(s);
        return RT_ERR_NO_DEVICE;
    }
    uint8_t *b = (uint8_t *)s->blob;
    DevScene &D = s->dev;
    D.entries = (const rt_entry *)(b + parts[0].off);
    D.prims = (const rt_prim *)(b + parts[1].off);
    D.xforms = (const rt_xform *)(b + parts[2].off);
    D.chains = (const rt_chain *)(b + parts[3].off);
    D.materials = (const rt_material *)(b + parts[4].off);
    D.textures = (const rt_texture *)(b + parts[5].off);
    D.texels = (const uint8_t *)(b + parts[6].off);
    D.lights = (const int32_t *)(b + parts[7].off);
    D.ord = (const int2 *)(b + parts[10].off);
    D.bvh = lanes_ok ? (const rt_bvh_node *)(b + parts[11].off) : nullptr;
    D.fprims = lanes_ok ? (const rt_prim *)(b + parts[12].off) : nullptr;
    D.leaves = lanes_ok ? (const rt_leaf *)(b + parts[13].off) : nullptr;
    D.fshade = lanes_ok ? (const rt_shade_rec *)(b + parts[14].off) : nullptr;
    D.ordered = ordered ? 1 : 0;
    D.n_entries = d->n_entries, D.n_lights = d->n_lights;
    D.cam = d->camera;
    s->fast_ok = fast_ok, s->fast_why = why;
    for (int m = 0; fast_ok && m < d->n_materials; ++m)
        s->fast_textured |= (int32_t)tab.rec[tab.materials_base + m][1] >= 0;
    s->lanes_ok = lanes_ok;
    s->bvh_levels = lanes_ok && bvh_depth > 0 ? bvh_depth : 1;
    s->lanes_small = lanes_ok && bvh.size() < 32767 && leaves.size() < 32767 && s->bvh_levels <= 31;
    for (const rt_leaf &lf : leaves) { // anything but static spheres and rects: the kernels with (more of) the rest of walk_leaf
        s->lanes_noise |= (lf.prim & RT_LEAF_PRIM_NOISE) != 0;
        const uint32_t k = lf.kind_chain & 255u;
        if ((lf.kind_chain >> 8) != 0u) s->lanes_leaf_kinds = 2, s->lanes_boxes = true; // under an instance chain: only the kernels for "anything" take rays down chains
        if (k == RT_PRIM_SPHERE || k == RT_PRIM_XY_RECT || k == RT_PRIM_XZ_RECT || k == RT_PRIM_YZ_RECT) continue;
        s->lanes_boxes = true;
        s->lanes_leaf_kinds = std::max(s->lanes_leaf_kinds, k == RT_LEAF_MOVING_SPHERE ? 1 : 2);
    }
    for (int m = 0; lanes_ok && m < d->n_materials; ++m)
        s->fast_textured |= d->materials[m].texture >= 0 && d->textures[d->materials[m].texture].kind != RT_TEX_CONSTANT;
    s->d_records = (uint32_t *)(b + parts[8].off);
    static_assert(offsetof(rt_record_table, face_user) == offsetof(rt_record_table, user_id) + sizeof tab.user_id, "one upload");
    s->d_rec_user = (int32_t *)(b + parts[9].off);
    s->d_task_counters = (unsigned *)(b + counter_off);
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) s->n_cus = cus;
    }
    s->layout = rtf::Layout{tab.n_stream, tab.materials_base, tab.lights_base, tab.n_lights, {tab.run[0], tab.run[1], tab.run[2], tab.run[3]}, tab.n_runs, tab.n_spheres, 0u, tab.n_lights > 0 ? 1.0f / (float)tab.n_lights : 0.0f};
    for (int k = 0; k < tab.n_lights && k < 32; ++k)
        if ((tab.rec[tab.lights_base + k][7] & 15u) == RT_REC_LIGHT_RECT) s->layout.rect_lights |= 1u << k;
    s->fast_flat = fast_ok && tab.flat != 0 && tab.n_runs > 0; // the run scan; else the generic one (it also walks flat streams)
    *out = s;
    return RT_OK;
}

extern "C" void rt_scene_destroy(rt_scene *s)
{
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    if (s->blob) (void)hipFree(s->blob);
    delete s;
}

static int check_params(const rt_render_params *p, const char *who)
{
    if (!p || p->struct_size != sizeof(rt_render_params)) return rt_fail(RT_ERR_INVALID, "%s: bad rt_render_params", who);
    if (p->nx < 1 || p->ny < 1 || p->nx > 65536 || p->ny > 65536) return rt_fail(RT_ERR_INVALID, "%s: image size %dx%d", who, p->nx, p->ny);
    if (p->sample_begin < 0 || p->sample_end < p->sample_begin) return rt_fail(RT_ERR_INVALID, "%s: sample range [%d,%d)", who, p->sample_begin, p->sample_end);
    if (p->sub_samples < 1 || p->sub_samples > 4096) return rt_fail(RT_ERR_INVALID, "%s: sub_samples %d", who, p->sub_samples);
    if (p->max_depth < 0) return rt_fail(RT_ERR_INVALID, "%s: max_depth %d", who, p->max_depth);
    if ((int64_t)(p->sample_end - p->sample_begin) * p->sub_samples > (1 << 24))
        return rt_fail(RT_ERR_INVALID, "%s: more than 2^24 paths per pixel in one call; split the sample range", who);
    return RT_OK;
}

extern "C" int rt_render(rt_scene *s, const rt_render_params *p, float *d_accum, void *stream_)
{
    if (!s || !d_accum) return rt_fail(RT_ERR_INVALID, "rt_render: null argument");
    int st = check_params(p, "rt_render");
    if (st != RT_OK) return st;
    hipStream_t stream = (hipStream_t)stream_;
    if (s->render_path == RT_PATH_FAST && !s->fast_ok && !s->lanes_ok)
        return rt_fail(RT_ERR_UNSUPPORTED, "rt_render: the fast path was requested but cannot run this scene (%s; and no ordered walk)", s->fast_why);
    // AUTO: the record table when the scene fits, else the per-lane fast path when it can run the scene (measured 2-7x the
    // exact walk also on trees of a dozen nodes: tools/path_bench.py); the exact path is what is left
    const bool use_fast = s->render_path == RT_PATH_FAST || (s->render_path == RT_PATH_AUTO && (s->fast_ok || s->lanes_ok));
    RT_HIP(hipSetDevice(s->device));
    const size_t bytes = sizeof(float) * 4 * (size_t)p->nx * p->ny;
    const bool accumulate = (p->flags & RT_RENDER_ACCUMULATE) != 0;
    const int ns = p->sample_end - p->sample_begin;
    if (ns == 0) {
        if (!accumulate) RT_HIP(hipMemsetAsync(d_accum, 0, bytes, stream));
        return RT_OK;
    }
    // the record-table path has two kernels: one wave per 8x8 tile (render_fast_kernel), or one block of four waves per
    // 16x16 tile that regroups its paths by class between scan and shade (render_fast_block_kernel: no textures, depth
    // and light list small enough for its packed state word)
    const bool block_off = s->opt_block_kernel == 0; // RT_OPT_BLOCK_KERNEL: the wave kernel instead
    const bool block_kernel = use_fast && s->fast_ok && !s->fast_textured && !block_off && p->max_depth < 65536 && s->layout.n_lights <= 16;
    // the per-lane path has two kernels as well: one wave per 8x8 tile, or (trees whose references fit 16 bits) one block of four
    // waves per 16x16 tile that regroups its paths every few node steps (render_fastlanes_block_kernel)
    const bool lanes_block = use_fast && !s->fast_ok && s->lanes_ok && s->lanes_small && p->max_depth < 65536 &&
                             (s->opt_lanes_regroup < 0 ? RT_LANES_REGROUP_DEFAULT : s->opt_lanes_regroup) != 0;
    const bool tile16 = block_kernel || lanes_block;
    // The block kernel's block is four waves (a 16x16 tile) or, for large chained launches, eight (32x16): fewer of its waves sit
    // at the borders between the classes of the exchange (Cornell 1024x1024: +1 % at 1024 spp, +2.2 % at 128 spp; 512x512x64 spp
    // -6 %: the last blocks of a small launch are larger).  RT_OPT_BLOCK_WAVES says which; 0: by the number of paths.
    const int64_t launch_paths = (int64_t)p->nx * p->ny * ns * p->sub_samples;
    const int block_waves = !block_kernel ? BLOCK_WAVES : s->opt_block_waves > 0 ? s->opt_block_waves
                            : (s->opt_chain != 0 && launch_paths >= ((int64_t)1 << 27)) ? 2 * BLOCK_WAVES : BLOCK_WAVES;
    const int tile_w = block_kernel ? 4 * block_waves : tile16 ? 16 : 8, tile_h = tile16 ? 16 : 8, waves_per_task = block_kernel ? block_waves : tile16 ? 4 : 1;
    RenderArgs A{};
    A.nx = p->nx, A.ny = p->ny;
    A.tiles_x = (p->nx + tile_w - 1) / tile_w, A.tiles_y = (p->ny + tile_h - 1) / tile_h;
    const int64_t tiles = (int64_t)A.tiles_x * A.tiles_y;
    // Tiles differ a lot in cost (glass sphere vs. open front), so the image is cut into many more
    // tasks than the chip holds waves (8192 at 8 waves/SIMD) and the dispatcher -- or, for the chained
    // kernels, the task counter -- balances them; measured on Cornell 1024x1024x1024spp: 16 Ki tasks 5.1, 128 Ki 6.9, 256 Ki 7.0, 1 Mi 6.3
    // Gpaths/s (one task per wave, early in the round).  Each task keeps at least 8 paths per lane -- 4 when that
    // leaves too few tasks (16 / 8 for the kernels that do not chain tasks).  Before tasks were chained the drain of a task's pool (a few long paths finishing alone) made
    // 16 / 8 the better rule; chained, 32 / 16 / 8 / 4 paths per lane give Cornell 128 spp 10.2 / 9.7 / 9.6 / 9.6 ms,
    // random spheres 64 spp 6.5 / 6.1 / 6.7 / 6.5, Next-Week final 1.61 / 1.74 / 1.79 / 1.67, 1M spheres 16 spp
    // 1.06 / 1.16 / 1.19 / 1.06 Gpaths/s.
    int64_t target_tasks = 262144;
    if (s->opt_target_tasks > 0) target_tasks = s->opt_target_tasks; // RT_OPT_TARGET_TASKS
    target_tasks /= waves_per_task;
    int64_t want_chunks = (target_tasks + tiles - 1) / tiles;
    const int64_t all_paths = (int64_t)ns * p->sub_samples;
    // chained kernels (RT_OPT_CHAIN_TASKS = 0: the one-task-per-wave / per-block forms)
    const bool chain = block_kernel && s->opt_chain != 0;
    A.chain = (use_fast && s->lanes_ok && s->opt_chain != 0) ? 1 : 0;
    int64_t hi_paths = (chain || A.chain) ? 8 : 16, lo_paths = hi_paths / 2;
    if (s->opt_paths_per_lane > 0) { hi_paths = s->opt_paths_per_lane; lo_paths = std::max<int64_t>(1, hi_paths / 2); } // RT_OPT_PATHS_PER_LANE
    const int64_t want = want_chunks, cap_hi = (all_paths + hi_paths - 1) / hi_paths, cap_lo = (all_paths + lo_paths - 1) / lo_paths;
    if (want_chunks > cap_hi) want_chunks = cap_hi;
    const int64_t few = 49152 / waves_per_task; // below this many tasks the end of the launch shows: the lower bound on paths per lane
    if (tiles * want_chunks < few) want_chunks = std::min(std::min((few + tiles - 1) / tiles, cap_lo), want);
    if (want_chunks > ns) want_chunks = ns;
    if (want_chunks < 1) want_chunks = 1;
    {   // a task's pool is counted in 32 bits (up to 512 items per sample in the block kernel)
        const int64_t min_chunks = ((int64_t)ns * p->sub_samples + (1 << 21) - 1) >> 21;
        if (want_chunks < min_chunks && min_chunks <= ns) want_chunks = min_chunks;
    }
    A.chunk_samples = (int)((ns + want_chunks - 1) / want_chunks);
    A.n_chunks = (ns + A.chunk_samples - 1) / A.chunk_samples;
    A.sample_begin = p->sample_begin, A.n_samples = ns;
    A.sub_samples = p->sub_samples, A.max_depth = p->max_depth, A.seed = p->seed;
    A.atomic_out = (accumulate || A.n_chunks > 1 || chain || A.chain) ? 1 : 0; // the chained kernels always add into the image
    const int64_t n_tasks = tiles * A.n_chunks;
    if (n_tasks > 0x7fffffff / 8) return rt_fail(RT_ERR_INVALID, "rt_render: %lld tasks; split the call", (long long)n_tasks);
    A.n_tasks = (int)n_tasks;
    if (A.atomic_out && !accumulate) RT_HIP(hipMemsetAsync(d_accum, 0, bytes, stream));
    const int blocks = (A.n_tasks + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    unsigned *const d_task_counter = s->d_task_counters + (s->launch_seq.fetch_add(1u) & 63u);
    RT_HIP(hipEventRecord(s->ev0, stream));
    if (lanes_block) {
        // one block per task, or (chained) as many blocks as the chip holds at once, each taking tasks from the counter
        const size_t lds = (size_t)s->bvh_levels * 256 * sizeof(short); // the home stacks
        A.stack_levels = s->bvh_levels;
        A.regroup_rounds = s->opt_regroup_rounds > 0 ? s->opt_regroup_rounds : RT_REGROUP_ROUNDS_DEFAULT;
#define RT_LAUNCH_LANES_BLOCK(TEX, BOX) \
    do { \
        dim3 gs((unsigned)A.n_tasks); \
        int fit = 0; \
        if (A.chain) { \
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit, render_fastlanes_block_kernel<TEX, BOX>, 256, lds) != hipSuccess || fit < 1) fit = 4; \
            if (s->opt_chain_blocks_per_cu > 0) fit = s->opt_chain_blocks_per_cu; \
            gs.x = (unsigned)std::min<int64_t>(A.n_tasks, (int64_t)s->n_cus * fit); \
            RT_HIP(hipMemsetAsync(d_task_counter, 0, sizeof(unsigned), stream)); \
        } \
        hipLaunchKernelGGL((render_fastlanes_block_kernel<TEX, BOX>), gs, dim3(256), lds, stream, s->dev, s->d_records, s->layout, A, d_accum, d_task_counter); \
    } while (0)
        if (s->fast_textured && s->lanes_boxes) RT_LAUNCH_LANES_BLOCK(true, true);
        else if (s->fast_textured) RT_LAUNCH_LANES_BLOCK(true, false);
        else if (s->lanes_boxes) RT_LAUNCH_LANES_BLOCK(false, true);
        else RT_LAUNCH_LANES_BLOCK(false, false);
#undef RT_LAUNCH_LANES_BLOCK
    } else if (use_fast && s->lanes_ok) {
        // chained: as many blocks as the chip holds at once (six waves per SIMD), each wave takes tasks from the counter
        int64_t per_cu = RT_LANES_WAVES_PER_EU;
        if (s->opt_chain_blocks_per_cu > 0) per_cu = s->opt_chain_blocks_per_cu; // RT_OPT_CHAIN_BLOCKS_PER_CU
        const dim3 g((unsigned)(A.chain ? std::min<int64_t>(blocks, (int64_t)s->n_cus * per_cu) : blocks)), b(64 * WAVES_PER_BLOCK);
        if (A.chain) RT_HIP(hipMemsetAsync(d_task_counter, 0, sizeof(unsigned), stream));
        const size_t lds = (size_t)WAVES_PER_BLOCK * s->bvh_levels * 256; // the walks' stacks
        A.stack_levels = s->bvh_levels;
        A.shade_batch = s->opt_shade_batch > 0 ? s->opt_shade_batch : RT_SHADE_BATCH;
        A.leaf_batch = s->opt_leaf_batch > 0 ? s->opt_leaf_batch : RT_LEAF_BATCH;
        int steps = s->bvh_levels > 20 ? 4 : 2;
        // deep trees: shade and test leaves in smaller batches (a lane that waits is a lane that does not walk, and walks are
        // long there); tools/batch_sweep.py, swept again after the round-3 walk (a node step costs less now, so waiting for a
        // fuller shade costs less): 1M spheres is at its best at (40, 4) -- 1776 against 1738 Mpaths/s at (32, 4) -- and the others near
        // (48, 16): random spheres 8786 against 8640 at (40, 16), smoke-box Cornell 9031 against 8534, Next-Week final 2314 against 2290
        // a tree of static spheres and rects only has a cheap leaf test, and waiting for a batch of it pays less: 4096 spheres
        // 2603 -> 2813 Mpaths/s at (48, 8)
        if (s->lanes_leaf_kinds == 0 && s->opt_leaf_batch <= 0) A.leaf_batch = 8;
        if (s->bvh_levels > 20) {
            if (s->opt_shade_batch <= 0) A.shade_batch = 40;
            if (s->opt_leaf_batch <= 0) A.leaf_batch = 4;
        }
        if (s->opt_node_steps == 2 || s->opt_node_steps == 4) steps = s->opt_node_steps; // RT_OPT_NODE_STEPS
        // A chained grid must be what the chip holds at once: a block that has to wait for a place starts when the others are
        // done, and does what is left alone.  How many blocks a CU holds depends on the LDS the tree's depth asks for (six at
        // 23 levels, five at 28), so the runtime is asked.
#define RT_LAUNCH_LANES(TEX, BOX, STEPS) \
    do { \
        dim3 gs = g; \
        int fit = 0; \
        if (A.chain && s->opt_chain_blocks_per_cu <= 0 && \
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit, render_fastlanes_kernel<TEX, BOX, STEPS>, 64 * WAVES_PER_BLOCK, lds) == hipSuccess && fit > 0) \
            gs.x = (unsigned)std::min<int64_t>(blocks, (int64_t)s->n_cus * std::min(fit, (int)per_cu)); \
        hipLaunchKernelGGL((render_fastlanes_kernel<TEX, BOX, STEPS>), gs, b, lds, stream, s->dev, s->d_records, s->layout, A, d_accum, d_task_counter); \
    } while (0)
#define RT_LAUNCH_LANES_STEPS(TEX, BOX) \
    do { if (steps == 4) RT_LAUNCH_LANES(TEX, BOX, 4); else RT_LAUNCH_LANES(TEX, BOX, 2); } while (0)
        if (s->lanes_noise) RT_LAUNCH_LANES_STEPS(2, 2); // (the kernel for any leaf kind, whatever the tree's)
        else if (s->fast_textured && s->lanes_leaf_kinds == 2) RT_LAUNCH_LANES_STEPS(1, 2);
        else if (s->fast_textured && s->lanes_leaf_kinds == 1) RT_LAUNCH_LANES_STEPS(1, 1);
        else if (s->fast_textured) RT_LAUNCH_LANES_STEPS(1, 0);
        else if (s->lanes_leaf_kinds == 2) RT_LAUNCH_LANES_STEPS(0, 2);
        else if (s->lanes_leaf_kinds == 1) RT_LAUNCH_LANES_STEPS(0, 1);
        else RT_LAUNCH_LANES_STEPS(0, 0);
#undef RT_LAUNCH_LANES_STEPS
#undef RT_LAUNCH_LANES
    } else if (use_fast && block_kernel && chain) {
        // as many blocks as the chip holds at once (eight per CU); each takes tasks from the counter until none is left
        const int64_t per_cu = RT_BLOCK_WAVES_PER_EU * 4 / block_waves;
        const dim3 g((unsigned)std::min<int64_t>(A.n_tasks, (int64_t)s->n_cus * per_cu)), b(64 * block_waves);
        RT_HIP(hipMemsetAsync(d_task_counter, 0, sizeof(unsigned), stream));
        if (block_waves == 2 * BLOCK_WAVES && s->fast_flat)
            hipLaunchKernelGGL((render_fast_block_kernel<true, 2 * BLOCK_WAVES, true>), g, b, 0, stream, s->dev, s->d_records, s->layout, A, d_accum, d_task_counter);
        else if (block_waves == 2 * BLOCK_WAVES)
            hipLaunchKernelGGL((render_fast_block_kernel<false, 2 * BLOCK_WAVES, true>), g, b, 0, stream, s->dev, s->d_records, s->layout, A, d_accum, d_task_counter);
        else if (s->fast_flat)
            hipLaunchKernelGGL((render_fast_block_kernel<true, BLOCK_WAVES, true>), g, b, 0, stream, s->dev, s->d_records, s->layout, A, d_accum, d_task_counter);
        else
            hipLaunchKernelGGL((render_fast_block_kernel<false, BLOCK_WAVES, true>), g, b, 0, stream, s->dev, s->d_records, s->layout, A, d_accum, d_task_counter);
    } else if (use_fast && block_kernel) {
        if (s->fast_flat)
            hipLaunchKernelGGL((render_fast_block_kernel<true, BLOCK_WAVES, false>), dim3(A.n_tasks), dim3(64 * BLOCK_WAVES), 0, stream, s->dev, s->d_records, s->layout, A, d_accum, d_task_counter);
        else
            hipLaunchKernelGGL((render_fast_block_kernel<false, BLOCK_WAVES, false>), dim3(A.n_tasks), dim3(64 * BLOCK_WAVES), 0, stream, s->dev, s->d_records, s->layout, A, d_accum, d_task_counter);
    } else if (use_fast) {
        const dim3 g(blocks), b(64 * WAVES_PER_BLOCK);
        if (s->fast_textured && s->fast_flat)
            hipLaunchKernelGGL((render_fast_kernel<true, true>), g, b, 0, stream, s->dev, s->d_records, s->layout, A, d_accum);
        else if (s->fast_textured)
            hipLaunchKernelGGL((render_fast_kernel<true, false>), g, b, 0, stream, s->dev, s->d_records, s->layout, A, d_accum);
        else if (s->fast_flat)
            hipLaunchKernelGGL((render_fast_kernel<false, true>), g, b, 0, stream, s->dev, s->d_records, s->layout, A, d_accum);
        else
            hipLaunchKernelGGL((render_fast_kernel<false, false>), g, b, 0, stream, s->dev, s->d_records, s->layout, A, d_accum);
    }
    else if (s->dev.n_entries > RT_UNIFORM_WALK_MAX)
        hipLaunchKernelGGL(render_kernel_lanes, dim3(blocks), dim3(64 * WAVES_PER_BLOCK), 0, stream, s->dev, A, d_accum);
    else
        hipLaunchKernelGGL(render_kernel_uniform, dim3(blocks), dim3(64 * WAVES_PER_BLOCK), 0, stream, s->dev, A, d_accum);
    RT_HIP(hipGetLastError());
    s->last_render_path = use_fast ? RT_PATH_FAST : RT_PATH_EXACT;
    RT_HIP(hipEventRecord(s->ev1, stream));
    s->timed = true;
    return RT_OK;
}

extern "C" int rt_last_render_ms(rt_scene *s, float *ms)
{
    if (!s || !ms) return rt_fail(RT_ERR_INVALID, "rt_last_render_ms: null argument");
    if (!s->timed) return rt_fail(RT_ERR_INVALID, "rt_last_render_ms: no render has been recorded on this scene");
    RT_HIP(hipSetDevice(s->device));
    RT_HIP(hipEventSynchronize(s->ev1));
    RT_HIP(hipEventElapsedTime(ms, s->ev0, s->ev1));
    return RT_OK;
}

extern "C" int rt_resolve(const float *d_accum, int32_t nx, int32_t ny, uint32_t *d_rgba8, void *stream_)
{
    if (!d_accum || !d_rgba8 || nx < 1 || ny < 1) return rt_fail(RT_ERR_INVALID, "rt_resolve: bad argument");
    const int n = nx * ny;
    hipLaunchKernelGGL(resolve_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream_, (const float4 *)d_accum, n, d_rgba8);
    RT_HIP(hipGetLastError());
    return RT_OK;
}

extern "C" int rt_render_to_host(rt_scene *s, const rt_render_params *p, float *h_accum, uint32_t *h_rgba8)
{
    if (!s || !h_accum) return rt_fail(RT_ERR_INVALID, "rt_render_to_host: null argument");
    int st = check_params(p, "rt_render_to_host");
    if (st != RT_OK) return st;
    RT_HIP(hipSetDevice(s->device));
    const size_t n = (size_t)p->nx * p->ny;
    float *d_accum = nullptr;
    uint32_t *d_img = nullptr;
    RT_HIP(hipMalloc(&d_accum, n * 16));
    hipError_t e = h_rgba8 ? hipMalloc(&d_img, n * 4) : hipSuccess;
    if (e == hipSuccess) {
        rt_render_params q = *p;
        q.flags &= ~RT_RENDER_ACCUMULATE;
        st = rt_render(s, &q, d_accum, nullptr);
        if (st == RT_OK && h_rgba8) st = rt_resolve(d_accum, p->nx, p->ny, d_img, nullptr);
        if (st == RT_OK) e = hipMemcpy(h_accum, d_accum, n * 16, hipMemcpyDeviceToHost);
        if (st == RT_OK && e == hipSuccess && h_rgba8) e = hipMemcpy(h_rgba8, d_img, n * 4, hipMemcpyDeviceToHost);
    }
    (void)hipFree(d_accum);
    if (d_img) (void)hipFree(d_img);
    if (e != hipSuccess) return rt_fail(RT_ERR_NO_DEVICE, "rt_render_to_host: %s", hipGetErrorString(e));
    return st;
}

extern "C" int rt_camera_rays(rt_scene *s, int32_t n, const float *h_st, const float *h_lens, const float *h_time01, float *h_rays)
{
    if (!s || n < 0 || (n > 0 && (!h_st || !h_rays))) return rt_fail(RT_ERR_INVALID, "rt_camera_rays: bad argument");
    if (n == 0) return RT_OK;
    RT_HIP(hipSetDevice(s->device));
    float *d = nullptr; // st | lens | time | rays
    const size_t N = (size_t)n;
    RT_HIP(hipMalloc(&d, sizeof(float) * N * (2 + 2 + 1 + 7)));
    float *d_st = d, *d_lens = d + 2 * N, *d_time = d + 4 * N, *d_rays = d + 5 * N;
    hipError_t e = hipMemcpy(d_st, h_st, 8 * N, hipMemcpyHostToDevice);
    if (e == hipSuccess && h_lens) e = hipMemcpy(d_lens, h_lens, 8 * N, hipMemcpyHostToDevice);
    if (e == hipSuccess && h_time01) e = hipMemcpy(d_time, h_time01, 4 * N, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(camera_kernel, dim3((n + 255) / 256), dim3(256), 0, nullptr, s->dev.cam, n, d_st, h_lens ? d_lens : nullptr,
                           h_time01 ? d_time : nullptr, d_rays);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(h_rays, d_rays, 28 * N, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return rt_fail(RT_ERR_NO_DEVICE, "rt_camera_rays: %s", hipGetErrorString(e));
    return RT_OK;
}

// the closest-hit kernel the scene's trace option selects, on the null stream (shared by the two entry points below)
static hipError_t launch_trace(rt_scene *s, int32_t n, const float *d_rays, float t_min, float t_max, uint32_t seed, rt_hit *d_hits)
{
    if (s->trace_path == RT_PATH_FAST && s->lanes_ok)
        hipLaunchKernelGGL(trace_fastlanes_kernel, dim3((n + 255) / 256), dim3(256), (size_t)4 * s->bvh_levels * 256, nullptr, s->dev, n, d_rays, t_min, t_max, seed, s->bvh_levels, d_hits);
    else if (s->trace_path == RT_PATH_FAST && s->fast_flat)
        hipLaunchKernelGGL(trace_fast_kernel<true>, dim3((n + 255) / 256), dim3(256), 0, nullptr, s->d_records, s->d_rec_user, s->layout, n,
                           d_rays, t_min, t_max, d_hits);
    else if (s->trace_path == RT_PATH_FAST)
        hipLaunchKernelGGL(trace_fast_kernel<false>, dim3((n + 255) / 256), dim3(256), 0, nullptr, s->d_records, s->d_rec_user, s->layout, n,
                           d_rays, t_min, t_max, d_hits);
    else if (s->dev.n_entries > RT_UNIFORM_WALK_MAX)
        hipLaunchKernelGGL(trace_kernel<true>, dim3((n + 255) / 256), dim3(256), 0, nullptr, s->dev, n, d_rays, t_min, t_max, seed, d_hits);
    else
        hipLaunchKernelGGL(trace_kernel<false>, dim3((n + 255) / 256), dim3(256), 0, nullptr, s->dev, n, d_rays, t_min, t_max, seed, d_hits);
    return hipGetLastError();
}

// The record-table scan keeps a lane's best hit as the bits of t above a rank key (rtf::Best) and compares them as one
// unsigned number: that orders like t only for t > 0, which t >= t_min guarantees when t_min is positive.  With
// t_min <= 0 a hit at negative t (sign bit set) would compare as farther than everything and be dropped, where the exact
// path and the reference report it; such a range is refused on that path instead of answered differently.
static int check_trace_range(const rt_scene *s, float t_min, const char *who)
{
    if (s->trace_path == RT_PATH_FAST && s->fast_ok && !(t_min > 0.0f))
        return rt_fail(RT_ERR_INVALID, "%s: t_min = %g on the record-table fast path, which needs t_min > 0 (use RT_PATH_EXACT for this range)", who, (double)t_min);
    return RT_OK;
}

extern "C" int rt_trace_rays(rt_scene *s, int32_t n, const float *h_rays, float t_min, float t_max, uint32_t seed, rt_hit *h_hits)
{
    if (!s || n < 0 || (n > 0 && (!h_rays || !h_hits))) return rt_fail(RT_ERR_INVALID, "rt_trace_rays: bad argument");
    if (n == 0) return RT_OK;
    if (s->trace_path == RT_PATH_FAST && !s->fast_ok && !s->lanes_ok)
        return rt_fail(RT_ERR_UNSUPPORTED, "rt_trace_rays: the fast path was requested but cannot run this scene (%s; and no ordered walk)", s->fast_why);
    const int st = check_trace_range(s, t_min, "rt_trace_rays");
    if (st != RT_OK) return st;
    RT_HIP(hipSetDevice(s->device));
    const size_t N = (size_t)n;
    float *d_rays = nullptr;
    rt_hit *d_hits = nullptr;
    RT_HIP(hipMalloc(&d_rays, 28 * N));
    hipError_t e = hipMalloc(&d_hits, sizeof(rt_hit) * N);
    if (e == hipSuccess) e = hipMemcpy(d_rays, h_rays, 28 * N, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = launch_trace(s, n, d_rays, t_min, t_max, seed, d_hits);
    if (e == hipSuccess) e = hipMemcpy(h_hits, d_hits, sizeof(rt_hit) * N, hipMemcpyDeviceToHost);
    (void)hipFree(d_rays);
    if (d_hits) (void)hipFree(d_hits);
    if (e != hipSuccess) return rt_fail(RT_ERR_NO_DEVICE, "rt_trace_rays: %s", hipGetErrorString(e));
    return RT_OK;
}

extern "C" int rt_trace_rays_noalloc(rt_scene *s, int32_t n, const float *h_rays, float t_min, float t_max, uint32_t seed, rt_hit *h_hits)
{
    if (!s || n < 0 || (n > 0 && (!h_rays || !h_hits))) return rt_fail(RT_ERR_INVALID, "rt_trace_rays_noalloc: bad argument");
    int st = RT_OK;
    if (n > RT_TRACE_NOALLOC_MAX_RAYS)
        return rt_fail(RT_ERR_INVALID, "rt_trace_rays_noalloc: %d rays, the static staging area holds %d; split the batch or use rt_trace_rays", n,
                       RT_TRACE_NOALLOC_MAX_RAYS);
    if (n == 0) return RT_OK;
    if (s->trace_path == RT_PATH_FAST && !s->fast_ok && !s->lanes_ok)
        return rt_fail(RT_ERR_UNSUPPORTED, "rt_trace_rays_noalloc: the fast path was requested but cannot run this scene (%s; and no ordered walk)", s->fast_why);
    if (!s->scratch || s->device < 0 || s->device >= NOALLOC_MAX_DEVICES)
        return rt_fail(RT_ERR_NO_DEVICE, "rt_trace_rays_noalloc: no staging area on device %d (it serves devices 0..%d)", s->device, NOALLOC_MAX_DEVICES - 1);
    st = check_trace_range(s, t_min, "rt_trace_rays_noalloc");
    if (st != RT_OK) return st;
    RT_HIP(hipSetDevice(s->device));
    std::lock_guard<std::mutex> in_use(g_scratch_in_use[s->device]); // the array is one per GPU: one call at a time on it
    const size_t N = (size_t)n;
    float *d_rays = reinterpret_cast<float *>(s->scratch);
    rt_hit *d_hits = reinterpret_cast<rt_hit *>(s->scratch + NOALLOC_HITS_OFFSET);
    // nothing has to start at zero: the kernel reads rays [0, n) -- all just written -- and writes every hit [0, n) it returns
    RT_HIP(hipMemcpy(d_rays, h_rays, 28 * N, hipMemcpyHostToDevice));
    RT_HIP(launch_trace(s, n, d_rays, t_min, t_max, seed, d_hits));
    RT_HIP(hipMemcpy(h_hits, d_hits, sizeof(rt_hit) * N, hipMemcpyDeviceToHost));
    return RT_OK;
}

extern "C" int rt_debug_eval(rt_scene *s, int kind, int32_t n, const float *h_in, float *h_out)
{
    if (!s || n < 0 || (n > 0 && (!h_in || !h_out))) return rt_fail(RT_ERR_INVALID, "rt_debug_eval: bad argument");
    if (kind < RT_EVAL_LIGHT_PDF || kind > RT_EVAL_BOUNCE_FAST) return rt_fail(RT_ERR_INVALID, "rt_debug_eval: unknown kind %d", kind);
    if (kind == RT_EVAL_BOUNCE_FAST && !s->fast_ok && !s->lanes_ok)
        return rt_fail(RT_ERR_UNSUPPORTED, "rt_debug_eval: the fast path cannot run this scene (%s; and no ordered walk)", s->fast_why);
    if (n == 0) return RT_OK;
    RT_HIP(hipSetDevice(s->device));
    const size_t bytes = sizeof(float) * 16 * (size_t)n;
    float *d_in = nullptr, *d_out = nullptr;
    RT_HIP(hipMalloc(&d_in, bytes));
    hipError_t e = hipMalloc(&d_out, bytes);
    if (e == hipSuccess) e = hipMemcpy(d_in, h_in, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        const dim3 g((n + 255) / 256), b(256);
        if (kind == RT_EVAL_BOUNCE) {
            if (s->dev.n_entries > RT_UNIFORM_WALK_MAX) hipLaunchKernelGGL(debug_bounce_exact_kernel<true>, g, b, 0, nullptr, s->dev, n, d_in, d_out);
            else hipLaunchKernelGGL(debug_bounce_exact_kernel<false>, g, b, 0, nullptr, s->dev, n, d_in, d_out);
        } else if (kind == RT_EVAL_BOUNCE_FAST) {
            if (s->lanes_ok)
                hipLaunchKernelGGL(debug_bounce_lanes_kernel, g, b, (size_t)4 * s->bvh_levels * 256, nullptr, s->dev, s->d_records, s->layout, n, s->bvh_levels, d_in, d_out);
            else if (s->fast_flat)
                hipLaunchKernelGGL(debug_bounce_fast_kernel<true>, g, b, 0, nullptr, s->dev, s->d_records, s->layout, n, d_in, d_out);
            else
                hipLaunchKernelGGL(debug_bounce_fast_kernel<false>, g, b, 0, nullptr, s->dev, s->d_records, s->layout, n, d_in, d_out);
        } else {
            hipLaunchKernelGGL(debug_eval_kernel, g, b, 0, nullptr, s->dev, s->d_records, s->layout, kind, n, d_in, d_out);
            if (kind == RT_EVAL_TEXTURE) hipLaunchKernelGGL(debug_wave_noise_kernel, g, b, 0, nullptr, s->dev, n, d_in, d_out);
        }
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(h_out, d_out, bytes, hipMemcpyDeviceToHost);
    (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    if (e != hipSuccess) return rt_fail(RT_ERR_NO_DEVICE, "rt_debug_eval: %s", hipGetErrorString(e));
    return RT_OK;
}

extern "C" int rt_scene_set_option(rt_scene *s, int option, int value)
{
    if (!s) return rt_fail(RT_ERR_INVALID, "rt_scene_set_option: null scene");
    if (option == RT_OPT_RENDER_PATH || option == RT_OPT_TRACE_PATH) {
        if (value != RT_PATH_AUTO && value != RT_PATH_EXACT && value != RT_PATH_FAST)
            return rt_fail(RT_ERR_INVALID, "rt_scene_set_option: value %d", value);
        if (value == RT_PATH_FAST && !s->fast_ok && !s->lanes_ok)
            return rt_fail(RT_ERR_UNSUPPORTED, "rt_scene_set_option: the fast path cannot run this scene (%s; and no ordered walk)", s->fast_why);
        (option == RT_OPT_RENDER_PATH ? s->render_path : s->trace_path) = value;
        return RT_OK;
    }
    if (value < 0) return rt_fail(RT_ERR_INVALID, "rt_scene_set_option: option %d value %d", option, value);
    switch (option) {
    case RT_OPT_BLOCK_KERNEL: s->opt_block_kernel = value != 0; return RT_OK;
    case RT_OPT_CHAIN_TASKS: s->opt_chain = value != 0; return RT_OK;
    case RT_OPT_TARGET_TASKS: s->opt_target_tasks = value; return RT_OK;
    case RT_OPT_PATHS_PER_LANE: s->opt_paths_per_lane = value; return RT_OK;
    case RT_OPT_NODE_STEPS:
        if (value != 0 && value != 2 && value != 4) return rt_fail(RT_ERR_INVALID, "rt_scene_set_option: node steps are 2 or 4 (0: default)");
        s->opt_node_steps = value;
        return RT_OK;
    case RT_OPT_SHADE_BATCH:
    case RT_OPT_LEAF_BATCH:
        if (value > 64) return rt_fail(RT_ERR_INVALID, "rt_scene_set_option: a batch is at most the 64 lanes of a wave");
        (option == RT_OPT_SHADE_BATCH ? s->opt_shade_batch : s->opt_leaf_batch) = value;
        return RT_OK;
    case RT_OPT_BLOCK_WAVES:
        if (value != 0 && value != BLOCK_WAVES && value != 2 * BLOCK_WAVES)
            return rt_fail(RT_ERR_INVALID, "rt_scene_set_option: the block kernel's block has %d or %d waves (0: by the size of the launch)", BLOCK_WAVES, 2 * BLOCK_WAVES);
        s->opt_block_waves = value;
        return RT_OK;
    case RT_OPT_LANES_REGROUP: s->opt_lanes_regroup = value != 0; return RT_OK;
    case RT_OPT_REGROUP_ROUNDS:
        if (value > 64) return rt_fail(RT_ERR_INVALID, "rt_scene_set_option: at most 64 walk rounds between two regroups");
        s->opt_regroup_rounds = value;
        return RT_OK;
    case RT_OPT_CHAIN_BLOCKS_PER_CU:
        if (value > 64) return rt_fail(RT_ERR_INVALID, "rt_scene_set_option: at most 64 blocks per CU");
        s->opt_chain_blocks_per_cu = value;
        return RT_OK;
    }
    return rt_fail(RT_ERR_INVALID, "rt_scene_set_option: unknown option %d", option);
}

#ifdef RT_COUNT_WALK
// development builds only (tools/work_counts.py): node steps, leaf tests, rays and paths of the per-lane kernels since the last reset
extern "C" int rt_debug_walk_counts(unsigned long long *out8, int reset)
{
    void *p = nullptr;
    RT_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(g_walk_counts)));
    RT_HIP(hipDeviceSynchronize());
    if (out8) RT_HIP(hipMemcpy(out8, p, 128, hipMemcpyDeviceToHost)); // (sixteen counters)
    if (reset) RT_HIP(hipMemset(p, 0, 128));
    return RT_OK;
}
#endif

#ifdef RT_TIME_PHASES
// development builds only (tools/phase_times.py): wave cycles per phase of render_fastlanes_kernel since the last reset
extern "C" int rt_debug_phase_cycles(unsigned long long *out4, int reset) // out4: sixteen counters
{
    void *p = nullptr;
    RT_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(g_phase_cycles)));
    RT_HIP(hipDeviceSynchronize());
    if (out4) RT_HIP(hipMemcpy(out4, p, 128, hipMemcpyDeviceToHost));
    if (reset) RT_HIP(hipMemset(p, 0, 128));
    return RT_OK;
}
#endif

#ifdef RT_COUNT_SHADE
// development builds only (tools/shade_counts.py): the block kernel's shading census since the last reset
extern "C" int rt_debug_shade_counts(unsigned long long *out16, int reset)
{
    void *p = nullptr;
    RT_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(g_shade_counts)));
    RT_HIP(hipDeviceSynchronize());
    if (out16) RT_HIP(hipMemcpy(out16, p, 128, hipMemcpyDeviceToHost));
    if (reset) RT_HIP(hipMemset(p, 0, 128));
    return RT_OK;
}
#endif

extern "C" int rt_scene_get_info(rt_scene *s, int what, int *value)
{
    if (!s || !value) return rt_fail(RT_ERR_INVALID, "rt_scene_get_info: null argument");
    switch (what) {
    case RT_INFO_FAST_PATH_AVAILABLE: *value = (s->fast_ok || s->lanes_ok) ? 1 : 0; return RT_OK;
    case RT_INFO_LAST_RENDER_PATH: *value = s->last_render_path; return RT_OK;
    case RT_INFO_RECORDS: *value = s->fast_ok ? s->layout.lights_base + s->layout.n_lights : -1; return RT_OK;
    case RT_INFO_BVH_LEVELS: *value = s->lanes_ok ? s->bvh_levels : 0; return RT_OK;
    }
    return rt_fail(RT_ERR_INVALID, "rt_scene_get_info: unknown item %d", what);
}
