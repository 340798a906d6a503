/* rt_amd.h -- C-ABI of the MI355X path-tracing core.
 *
 * This is the drop-in boundary for the hot path of droune2001/ray_tracing:
 * the recursive color() (reference src/main.cpp:91-127) driven by the
 * thread_pool render loop (src/main.cpp:203-256, 543-561) and the resolve
 * step (src/main.cpp:708-743).  The reference has no FFI layer; the seam is
 * the argument set handed to compute_one_sample_task (src/main.cpp:545-554):
 * world (BVH root), important_hitables, camera, image size, sub_samples,
 * max_depth, shared_buffer.  The host keeps the reference scene API
 * (ray_tracing_amd/host/rt_scene_api.h mirrors scenes.h, hitable.h, bvh.h,
 * camera.h, material.h, texture.h, pdf.h) and flattens the object tree into the
 * plain arrays below; everything from ray generation to the accumulated
 * float4 image then runs on the GPU.
 *
 * Rules of the boundary
 *   - extern "C", plain pointers and sizes only; no C++ or torch types.
 *   - Every call returns RT_OK (0) or a negative rt_status; rt_last_error()
 *     returns a thread-local message for the last failure.  Nothing throws.
 *   - The caller owns every host array passed in; rt_scene_create copies them
 *     to the device and owns that copy until rt_scene_destroy.
 *   - Pointers named d_* are DEVICE pointers (hipMalloc / torch CUDA tensors);
 *     all other pointers are host pointers.
 *   - `stream` arguments are a hipStream_t passed as void* (NULL = default).
 *   - There is no CPU fallback: without a HIP device every compute entry point
 *     fails with RT_ERR_NO_DEVICE.
 */
#ifndef RT_AMD_H_
#define RT_AMD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 1

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID = -1,     /* bad argument / inconsistent scene description */
    RT_ERR_NO_DEVICE = -2,   /* no HIP device, or a HIP call failed           */
    RT_ERR_UNSUPPORTED = -3, /* scene composition the flattener cannot express */
    RT_ERR_NOMEM = -4
} rt_status;

/* ------------------------------------------------------------------ */
/* Flattened scene                                                      */
/* ------------------------------------------------------------------ */

/* One step of the depth-first walk of the reference's bvh_node tree
 * (src/bvh.h:123-163 visits left then right, always both).  Entries are stored
 * in pre-order.  An inner entry carries the node's aabb and `skip`, the index
 * of the first entry after its subtree, taken when aabb::hit
 * (src/aabb.h:60-83) fails.  A leaf entry names a primitive.  Two 16-byte
 * halves so a lane fetches one entry with two dwordx4 loads. */
typedef struct rt_entry {
    float   lo[3];  /* inner: box min                                          */
    int32_t a;      /* inner: skip index; leaf: primitive index                */
    float   hi[3];  /* inner: box max                                          */
    int32_t b;      /* bits 0-7 rt_entry_kind, bits 8-31 chain index of the
                       space this entry is expressed in                        */
} rt_entry;

enum rt_entry_kind { RT_ENTRY_INNER = 0, RT_ENTRY_LEAF = 1 };

/* Primitive kinds.  Values are part of the ABI. */
enum rt_prim_kind {
    RT_PRIM_SPHERE = 0,        /* src/sphere.h:21-59    f = cx cy cz r                  */
    RT_PRIM_MOVING_SPHERE = 1, /* src/sphere.h:147-183  f = c0xyz c1xyz r t0 t1         */
    RT_PRIM_XY_RECT = 2,       /* src/plane.h:9-29      f = a0 a1 b0 b1 k  (x0 x1 y0 y1 z) */
    RT_PRIM_XZ_RECT = 3,       /* src/plane.h:47-67     f = x0 x1 z0 z1 y               */
    RT_PRIM_YZ_RECT = 4,       /* src/plane.h:109-129   f = y0 y1 z0 z1 x               */
    RT_PRIM_GROUP = 5,         /* hitable_list / box: src/hitable_list.h:18-34, src/box.h:7-23;
                                  children = prims[first .. first+count)                */
    RT_PRIM_MEDIUM = 6         /* constant_medium src/volume.h:23-53; first = boundary
                                  primitive, f[0] = density, material = isotropic       */
};

#define RT_PRIM_FLIP 1u     /* odd number of flip_normals wrappers (src/hitable.h:27-38) */
#define RT_PRIM_NEEDS_UV 2u /* material's texture reads (u,v): compute them on a hit     */

typedef struct rt_prim {
    int32_t kind;     /* rt_prim_kind                                           */
    uint32_t flags;   /* RT_PRIM_*                                              */
    int32_t material; /* index into materials (-1 for GROUP)                    */
    int32_t chain;    /* instance chain from world space to this primitive      */
    int32_t first;    /* GROUP: first child; MEDIUM: boundary primitive         */
    int32_t count;    /* GROUP: number of children                              */
    int32_t user_id;  /* stable id: ordinal of this leaf in the depth-first walk
                         of the final tree (what rt_trace_rays reports)         */
    int32_t reserved;
    float   f[12];    /* parameters, layout per kind (see rt_prim_kind)         */
} rt_prim;            /* 80 bytes                                               */

/* Instance transforms (src/transforms.h).  A chain lists the operations in the
 * order a ray meets them going down the tree, outermost first. */
enum rt_xform_kind { RT_XFORM_TRANSLATE = 0, RT_XFORM_ROTATE_Y = 1 };
typedef struct rt_xform {
    int32_t kind;
    float   a, b, c; /* TRANSLATE: offset xyz.  ROTATE_Y: a = sin_theta, b = cos_theta */
} rt_xform;
typedef struct rt_chain {
    int32_t first, count; /* xforms[first .. first+count); chain 0 is the identity */
} rt_chain;

enum rt_material_kind { /* src/material.h */
    RT_MAT_LAMBERTIAN = 0,
    RT_MAT_METAL = 1,
    RT_MAT_DIELECTRIC = 2,
    RT_MAT_DIFFUSE_LIGHT = 3,
    RT_MAT_ISOTROPIC = 4
};
typedef struct rt_material {
    int32_t kind;
    int32_t texture; /* albedo / emit texture, -1 for dielectric */
    float   p0;      /* metal: fuzz (already clamped to <= 1); dielectric: ref_idx */
    int32_t reserved;
} rt_material;

enum rt_texture_kind { /* src/texture.h */
    RT_TEX_CONSTANT = 0,
    RT_TEX_CHECKER = 1, /* a = odd texture, b = even texture */
    RT_TEX_NOISE = 2,   /* scale; perlin gradient table is built into the library */
    RT_TEX_IMAGE = 3    /* texels[offset ..), nx * ny * 3 bytes RGB */
};
typedef struct rt_texture {
    int32_t kind;
    int32_t a, b;
    int32_t nx, ny;
    int32_t reserved;
    int64_t offset;
    float   color[3];
    float   scale;
} rt_texture; /* 48 bytes */

/* camera block = the members camera::camera computes (src/camera.h:29-52) */
typedef struct rt_camera {
    float origin[3];
    float u[3], v[3], w[3];
    float lower_left_corner[3];
    float horizontal[3];
    float vertical[3];
    float lens_radius;
    float time0, time1;
} rt_camera;

typedef struct rt_scene_desc {
    uint32_t struct_size; /* sizeof(rt_scene_desc) */
    uint32_t abi_version; /* RT_ABI_VERSION        */
    const rt_entry*    entries;   int32_t n_entries;
    const rt_prim*     prims;     int32_t n_prims;
    const rt_xform*    xforms;    int32_t n_xforms;
    const rt_chain*    chains;    int32_t n_chains;   /* chains[0] = identity */
    const rt_material* materials; int32_t n_materials;
    const rt_texture*  textures;  int32_t n_textures;
    const uint8_t*     texels;    int64_t n_texel_bytes;
    /* important_hitables (src/main.cpp:107, src/hitable_list.h:67-88): primitive
     * indices; -1 marks a member whose type has no pdf_value/random override
     * (src/hitable.h:19-20: pdf 0, direction (1,0,0)). */
    const int32_t*     lights;    int32_t n_lights;
    rt_camera camera;
} rt_scene_desc;

typedef struct rt_scene rt_scene; /* opaque, device-resident */

/* Replaces: `cornell_box(&world,&important,&cam,aspect)` + `new bvh_node(...)`
 * hand-off to the render tasks (src/main.cpp:505-511, 545-554).  `device` is
 * the HIP device ordinal.  Validates every index in the description. */
int rt_scene_create(const rt_scene_desc* desc, int device, rt_scene** out);
/* The same with flags (rt_scene_create passes 0).  RT_CREATE_KEEP_BOX_RECORDS: a small scene keeps the inner nodes of
 * the reference's tree in its record table (normally dropped at <= 16 primitives): a test aid that sends such a
 * scene through the generic scan.  No counterpart in the reference. */
#define RT_CREATE_KEEP_BOX_RECORDS 1u
int rt_scene_create_flags(const rt_scene_desc* desc, int device, uint32_t flags, rt_scene** out);
void rt_scene_destroy(rt_scene* scene);

/* Threading and devices.  A scene belongs to the device it was created on; every entry point that takes a scene makes
 * that device current for the calling thread and leaves it current.  One host thread at a time per rt_scene: the
 * calls are not re-entrant on one scene (launch bookkeeping, the timing events).  rt_last_render_ms refers to the
 * latest rt_render of the scene.  rt_resolve takes no scene: it runs on the calling thread's CURRENT device, which
 * must be the one that holds d_accum and d_rgba8.  Different scenes may be used from different threads. */

/* ------------------------------------------------------------------ */
/* Render                                                               */
/* ------------------------------------------------------------------ */

typedef struct rt_render_params {
    uint32_t struct_size;
    int32_t  nx, ny;        /* image_width, image_height (src/main.cpp:546-547)        */
    int32_t  sample_begin;  /* first sample index of this call (sample_id, :549)       */
    int32_t  sample_end;    /* one past the last; GPU g of G gets a disjoint range      */
    int32_t  sub_samples;   /* -S, src/main.cpp:228                                     */
    int32_t  max_depth;     /* -r, src/main.cpp:98                                      */
    uint32_t seed;          /* key of the counter-based per-path RNG (include/rt_rng.h) */
    uint32_t flags;         /* RT_RENDER_*                                              */
} rt_render_params;

#define RT_RENDER_ACCUMULATE 1u /* add into d_accum instead of overwriting it */

/* Replaces the loop `for i<ns: addTask(compute_one_sample_task)`, pool->finish()
 * (src/main.cpp:543-561) for samples [sample_begin, sample_end).
 * d_accum: device float[ny][nx][4], per pixel [A, B, G, R] exactly as
 * compute_one_sample_task writes it (src/main.cpp:238-242; row 0 = image top),
 * but holding SUMS over the samples of this call: A = number of samples,
 * B/G/R = sum over samples of the sub-sample average of de_nan(color()).
 * Asynchronous on `stream`; the buffer is ready when the stream is. */
int rt_render(rt_scene* scene, const rt_render_params* params, float* d_accum, void* stream);

/* Replaces the resolve loop (src/main.cpp:708-743): per pixel sum / total
 * samples (taken from the A channel), sqrtf gamma, clamp01, (unsigned)(c*255.99f),
 * packed 0xff000000 | b<<16 | g<<8 | r.  d_rgba8: device uint32[ny][nx]. */
int rt_resolve(const float* d_accum, int32_t nx, int32_t ny, uint32_t* d_rgba8, void* stream);

/* Convenience for callers without their own device buffers: render
 * [sample_begin, sample_end) and copy the float accumulators (and, when
 * h_rgba8 is not NULL, the resolved 8-bit image) to HOST memory. Blocking. */
int rt_render_to_host(rt_scene* scene, const rt_render_params* params, float* h_accum,
                      uint32_t* h_rgba8);

/* Samples [*begin, *end) of GPU `rank` of `world` out of [sample_begin, sample_end): the ranges of all ranks tile the
 * budget, sizes differ by at most one.  This is the partition of src/main.cpp:543-557 (one task per sample) cut into
 * `world` contiguous pieces.  No device needed. */
int rt_sample_range(int32_t rank, int32_t world, int32_t sample_begin, int32_t sample_end, int32_t* begin, int32_t* end);

/* One image on several GPUs of one node, from one host process (no Python, no launcher): a scene is created on every
 * device of `devices`, device i renders the samples rt_sample_range(i, n_devices, ...) gives it (all devices at once,
 * one stream each), the float4 sums are added onto devices[0] with ONE RCCL sum-reduce over xGMI (ncclReduce, float,
 * 4*nx*ny elements), and devices[0] resolves.  Replaces thread_pool + the per-sample sum of src/main.cpp:523-561,
 * 720-726 for a multi-GPU node.  h_accum: host float[ny][nx][4] sums; h_rgba8: host uint32[ny][nx] or NULL.  Blocking.
 * RCCL (librccl.so.1) is opened at run time; without it the call fails with RT_ERR_UNSUPPORTED when n_devices > 1.
 * RT_MULTI_REDUCE_ALWAYS: run the reduce also on ONE device (a one-rank communicator): exercises the RCCL path. */
#define RT_MULTI_REDUCE_ALWAYS 1u
int rt_render_multi_to_host(const rt_scene_desc* desc, const int32_t* devices, int32_t n_devices,
                            const rt_render_params* params, uint32_t flags, float* h_accum, uint32_t* h_rgba8);

/* The same with the set-up done ONCE: the reference builds its thread pool once and renders with it
 * (src/main.cpp:523-561); a context holds a scene and a stream per device, ONE RCCL communicator (ncclCommInitAll) and
 * the per-device accumulators (grown to the largest image asked for, then kept), so that the time of an image is
 * render + reduce + resolve + copy, not communicator set-up.  rt_render_multi_to_host is create + one image + destroy.
 *   rt_multi_create          flags: RT_MULTI_REDUCE_ALWAYS as above.  On failure nothing is left allocated.
 *   rt_multi_render_to_host  one image; h_accum and h_rgba8 as above, either may be NULL (not both).  Blocking; one
 *                            image at a time per context (calls on one context are serialised by a lock).  After a
 *                            failed call the streams have been drained; the context may be used again or destroyed.
 *   rt_multi_set_option      rt_scene_set_option on every device's scene.
 *   rt_multi_last_render_ms  the longest of the devices' render kernels of the latest image (rt_last_render_ms).
 *   rt_multi_destroy         waits for every stream, then frees communicator, buffers, streams and scenes. */
typedef struct rt_multi rt_multi;
int rt_multi_create(const rt_scene_desc* desc, const int32_t* devices, int32_t n_devices, uint32_t flags, rt_multi** out);
int rt_multi_render_to_host(rt_multi* ctx, const rt_render_params* params, float* h_accum, uint32_t* h_rgba8);
int rt_multi_set_option(rt_multi* ctx, int option, int value);
int rt_multi_last_render_ms(rt_multi* ctx, float* ms_out);
void rt_multi_destroy(rt_multi* ctx);

/* ------------------------------------------------------------------ */
/* Closest-hit queries (the parity surface)                             */
/* ------------------------------------------------------------------ */

/* camera::get_ray (src/camera.h:54-68) for n samples, with the random inputs
 * made explicit: h_st = n pairs (s,t); h_lens = n pairs of unit-disk points or
 * NULL (= (0,0)); h_time01 = n numbers in [0,1) or NULL (= 0).
 * h_rays_out: n x 7 floats (origin xyz, direction xyz, time). */
int rt_camera_rays(rt_scene* scene, int32_t n, const float* h_st, const float* h_lens,
                   const float* h_time01, float* h_rays_out);

typedef struct rt_hit {
    int32_t prim_id;  /* rt_prim.user_id of the closest primitive, -1 on a miss */
    int32_t material; /* material index, -1 on a miss                           */
    float   t;
    float   u, v;     /* 0 unless the primitive sets them                       */
    float   p[3];
    float   normal[3];
    float   reserved;
} rt_hit; /* 48 bytes */

/* hitable::hit on the world (src/main.cpp:94: world->hit(r, 0.001f, FLT_MAX, rec))
 * for n rays of 7 floats each.  `seed` keys the draws constant_medium::hit
 * makes (src/volume.h:43); it is unused for scenes without media.
 * On RT_PATH_FAST with a record table (RT_INFO_RECORDS > 0) t_min must be positive (the scan orders hits by the bits of
 * t, which order like t only for t > 0): t_min <= 0 or NaN is rejected with RT_ERR_INVALID; use RT_PATH_EXACT for such
 * ranges.  The reference itself only ever passes 0.001f (src/main.cpp:94) and 0.0001f (src/plane.h:81). */
int rt_trace_rays(rt_scene* scene, int32_t n, const float* h_rays, float t_min, float t_max,
                  uint32_t seed, rt_hit* h_hits_out);

/* rt_trace_rays for callers that cannot allocate (a tight query loop, a harness that forbids hipMalloc):
 * same arguments, same results, but the device staging of rays and hits lives in a static device array of the
 * library (RT_TRACE_NOALLOC_MAX_RAYS x 76 bytes = 4.75 MiB, held by every GPU the library runs on until the
 * process ends) instead of two hipMalloc'ed buffers per call.  Calls no hipMalloc / hipFree.  n above
 * RT_TRACE_NOALLOC_MAX_RAYS is rejected with RT_ERR_INVALID.  Every call on a GPU shares that GPU's array:
 * the library serialises the calls of ONE process on a GPU with a per-device lock (copy in, kernel, copy out under
 * it); calls from different processes have arrays of their own.  Work the caller queues itself on other streams
 * must not touch the array.
 * No counterpart in the reference (it has no device and no staging); it exists for callers that cannot allocate. */
#define RT_TRACE_NOALLOC_MAX_RAYS 65536
int rt_trace_rays_noalloc(rt_scene* scene, int32_t n, const float* h_rays, float t_min, float t_max,
                          uint32_t seed, rt_hit* h_hits_out);

/* ------------------------------------------------------------------ */
/* Known-answer evaluation of single functions on the device            */
/* ------------------------------------------------------------------ */
/* Evaluates, for n items of 16 floats each, the device functions behind the reference functions named below, called as
 * the render kernels call them, and writes 16 floats per item; rt:: is the family of RT_PATH_EXACT, rtf:: the one of
 * RT_PATH_FAST.  A test surface (tests/test_gpu_kat.py compares with the reference-written vectors of
 * tests/golden/kat.npz and with the oracle), never on a render path.  Unused inputs are ignored, unused outputs are 0.
 *   RT_EVAL_LIGHT_PDF   in kind (0 xz_rect, 1 sphere), o[3], v[3], f[5] (rt_prim.f), normal[3]
 *                       out rt:: pdf_value (src/plane.h:76-91, src/sphere.h:68-83), cosine_pdf::value (src/pdf.h:13-24),
 *                           then the same two of rtf:: (for the normalised v)
 *   RT_EVAL_ONB         in n[3]   out rt:: onb u, v, w (src/vec3.h:195-224), unit(n), get_sphere_uv(unit(n)) (src/utils.h:57-68)
 *   RT_EVAL_ONB_FAST    in n[3], a[3]   out rtf:: onb u, v, w, unit(n), onb.local(a)
 *   RT_EVAL_OPTICS      in v[3], n[3], ni_over_nt, cosine, ref_idx   out reflect[3], refracted?, refract[3], schlick (src/utils.h:4-30)
 *   RT_EVAL_PERLIN      in p[3]   out rt:: noise, turb(7) (src/perlin.h:54-136), rtf:: noise, turb (corner loop), rtf:: noise, turb (corners written out)
 *   RT_EVAL_TEXTURE     in index into the scene's textures, u, v, p[3]   out rt:: value[3] (src/texture.h:9-81), rtf:: value[3], rtf::wave_noise_value (noise textures; else 0)
 *   RT_EVAL_SAMPLERS    in key, pixel (bit patterns of the stream, include/rt_rng.h), o[3]; each sampler from draw 0
 *                       out rt:: hitable_list::random(o) over the scene's important list[3] (src/hitable_list.h:79-88),
 *                           random_cosine_direction[3] (src/vec3.h:173-182), random_in_unit_sphere[3], random_in_unit_disk[2]
 *                           (src/utils.h:32-55), rtf:: unit sphere[3], unit disk[2]
 *   RT_EVAL_BOUNCE      in ray[7], key, pixel   world->hit() and ONE bounce of color() (src/main.cpp:91-127) at depth 0 of 50, T = 1
 *                       out hit?, material kind, continues?, scattered direction[3], weight[3] (the throughput factor; the
 *                           radiance when the path ends), t        -- the EXACT family
 *   RT_EVAL_BOUNCE_FAST the same through the FAST family (record table or the library's own tree, as rt_render picks)
 * No counterpart in the reference. */
enum rt_eval_kind {
    RT_EVAL_LIGHT_PDF = 1, RT_EVAL_ONB = 2, RT_EVAL_ONB_FAST = 3, RT_EVAL_OPTICS = 4, RT_EVAL_PERLIN = 5, RT_EVAL_TEXTURE = 6,
    RT_EVAL_SAMPLERS = 7, RT_EVAL_BOUNCE = 8, RT_EVAL_BOUNCE_FAST = 9
};
int rt_debug_eval(rt_scene* scene, int kind, int32_t n, const float* h_in16, float* h_out16);

/* ------------------------------------------------------------------ */
/* Execution paths                                                      */
/* ------------------------------------------------------------------ */
/* The library holds an exact and a fast device implementation of world->hit() + color():
 *   RT_PATH_EXACT  scene tables in memory; any scene the flattener accepts; every float operation that
 *                  decides a hit repeats the reference's, so ids, t, p and normal equal the CPU
 *                  reference bit for bit (the reference's own tree, walked wave-uniformly when it has at
 *                  most 64 entries, else per lane, ordered and culled without changing the result).
 *   RT_PATH_FAST   hardware-rate reciprocal/sqrt and FMA; same algorithm, same random-draw order per
 *                  primitive, results equal up to rounding (and up to which of two hits at exactly the
 *                  same distance wins).  Two forms, chosen by the scene:
 *                  - at most 64 records (the Cornell box: 23), static spheres / rects / boxes: the scene
 *                    lives in registers and is scanned wave-uniformly (RT_INFO_RECORDS > 0);
 *                  - anything else (media, moving spheres, (u,v) textures, up to millions of primitives):
 *                    every lane walks the library's own surface-area-heuristic tree over the same
 *                    primitives (RT_INFO_RECORDS == -1).
 * RT_PATH_AUTO (default): rt_render uses FAST whenever one of its two forms can run the scene (EXACT is left
 * for what neither can: RT_INFO_FAST_PATH_AVAILABLE == 0); rt_trace_rays uses EXACT.
 * No counterpart in the reference (it has one CPU path). */
enum rt_option {
    RT_OPT_RENDER_PATH = 1, /* rt_path */
    RT_OPT_TRACE_PATH = 2,  /* rt_path */
    /* Tuning and test switches of rt_render; none changes what is computed (paths are keyed by pixel and sample),
     * only how the work is cut up.  0 = the library's own choice unless said otherwise.  (They are options of a scene
     * on purpose, not environment variables: nothing outside the caller's code can change which kernel runs.) */
    RT_OPT_BLOCK_KERNEL = 3,        /* 1 (default): record-table scenes without textures run the block kernel; 0: the wave kernel */
    RT_OPT_CHAIN_TASKS = 4,         /* 1 (default): waves / blocks take task after task from a counter; 0: one task each */
    RT_OPT_TARGET_TASKS = 5,        /* tasks a launch is cut into (default 262144 waves' worth) */
    RT_OPT_PATHS_PER_LANE = 6,      /* lower bound on a task's size, in paths per lane */
    RT_OPT_NODE_STEPS = 7,          /* per-lane walk: node steps between two looks at who waits (2 or 4) */
    RT_OPT_CHAIN_BLOCKS_PER_CU = 8, /* per-lane walk: blocks per CU of the chained grid */
    RT_OPT_SHADE_BATCH = 9,         /* per-lane walk: lanes whose walk is over wait until this many of a wave's 64 are */
    RT_OPT_LEAF_BATCH = 10,         /* per-lane walk: lanes standing on a leaf wait until this many do */
    RT_OPT_LANES_REGROUP = 11,      /* per-lane walk, trees of fewer than 32767 nodes and leaves: 1 = the block kernel that regroups its
                                       paths every few node steps, 0 = the wave kernel (default: the library's choice) */
    RT_OPT_REGROUP_ROUNDS = 12,     /* that kernel: rounds of (two node steps + one leaf test) between two regroups */
    RT_OPT_BLOCK_WAVES = 13         /* block kernel: waves of a block, 4 (a 16x16 tile) or 8 (32x16; chained launches only);
                                       0 (default): 8 for launches of at least 2^27 paths */
};
enum rt_path { RT_PATH_AUTO = 0, RT_PATH_EXACT = 1, RT_PATH_FAST = 2 };
int rt_scene_set_option(rt_scene* scene, int option, int value);
enum rt_info {
    RT_INFO_FAST_PATH_AVAILABLE = 1,
    RT_INFO_LAST_RENDER_PATH = 2,
    RT_INFO_RECORDS = 3,
    RT_INFO_BVH_LEVELS = 4 /* inner levels of the library's own tree (per-lane fast path), 0 when the scene has none */
};
int rt_scene_get_info(rt_scene* scene, int what, int* value);

/* ------------------------------------------------------------------ */
/* Misc                                                                 */
/* ------------------------------------------------------------------ */
const char* rt_last_error(void);
int rt_abi_version(void);
/* Number of HIP devices visible, or a negative rt_status. */
int rt_device_count(void);
/* Timing of the most recent rt_render on this scene: milliseconds between HIP
 * events recorded on the render stream around the path-tracing kernel.
 * Blocks until that kernel has finished. */
int rt_last_render_ms(rt_scene* scene, float* ms_out);

#ifdef __cplusplus
}
#endif
#endif /* RT_AMD_H_ */
