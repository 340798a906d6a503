/* rt_bvh.h -- the GPU-side tree of the fast per-lane path (rt_fastlanes.hpp).
 *
 * The reference's tree (src/bvh.h:79-121: random axis, median split, boxes stored in the node itself, visited
 * left-right-both) is kept as it is for the exact path.  For the fast path the library builds a second tree over
 * the SAME primitives: world-space boxes (instance chains folded in), binned surface-area-heuristic splits, and
 * nodes that hold the boxes of BOTH children, so one 32-byte fetch (half-float boxes) decides two boxes and a leaf needs no node of
 * its own.  The closest hit over a set of primitives does not depend on the tree that finds it (equal-distance
 * ties aside), so results are those of the reference's tree up to rounding; tests compare the two.
 * Internal to the library; not part of the ABI. */
#ifndef RT_BVH_H_
#define RT_BVH_H_

#include <stdint.h>

#include <vector>

#include "../../include/rt_amd.h"

/* child reference: >= 0 inner node index; < 0 leaf: ~ref is an rt_leaf index; RT_BVH_NONE: no child */
#define RT_BVH_NONE ((int32_t)0x80000000)
/* Pending far children per lane live in a stack in LDS, one level per inner level of the tree (rt_bvh_build reports how
 * many the tree has; the kernels size their dynamic LDS from it).  23 levels x 64 lanes x 4 B x 4 waves + the tile sums =
 * 26 KiB per block: six blocks per CU, six waves per SIMD -- the builder keeps to RT_BVH_MAX_DEPTH if it can (median
 * splits once the remaining levels only just fit a balanced tree; enough for 4M primitives).  28 levels leave it freer on
 * the 1M-sphere scene but hold a SIMD to five waves: measured 3-10 % slower on every scene.  Beyond 4M primitives it
 * falls back to RT_BVH_DEPTH_LIMIT. */
#ifndef RT_BVH_MAX_DEPTH
#define RT_BVH_MAX_DEPTH 23
#endif
#define RT_BVH_DEPTH_LIMIT 28

/* Inner node, 32 bytes (two dwordx4 loads from one cache line): the boxes of both children as IEEE half floats, rounded
 * outwards (lo down, hi up; beyond the half range they become -inf / +inf), and the two child references.  The walk
 * feeds the halves straight into v_fma_mix_f32 (t = half(plane) * (1/d) - o/d in one instruction), so the halved node
 * costs no decoding.  The two planes of an axis share a dword, lo in the low half: rotated by 16 bits when the ray's
 * direction is negative on that axis (one v_alignbit with a per-ray amount), the low half is the plane the ray meets first,
 * so entry and exit distances need no min / max per axis.  A box that is a little too large only costs a visit. */
typedef struct rt_bvh_node {
    uint16_t box0[3][2]; /* [axis]{lo, hi} */
    int32_t  ref0;
    uint16_t box1[3][2];
    int32_t  ref1;
} rt_bvh_node;

/* Leaf, 32 bytes, stored in the order the tree's leaves are met (neighbours in space are neighbours in memory):
 * what the intersection test reads, and the rt_prim the hit is built from afterwards.
 *   sphere: f = cx cy cz r    rect: f = a0 a1 b0 b1 k    moving sphere, constant_medium: the test reads prims[prim]
 *   box (a GROUP that rt_box.h recognises): f = p0.x p0.y p0.z p1.x p1.y and p1.z in the last word */
/* Leaf kinds beyond rt_prim_kind (the low byte of kind_chain): leaves that carry everything their test reads, so that the walk
 * does not follow the leaf with dependent fetches of the 80-byte primitive and of its boundary (round 3: on Next-Week final the
 * leaf tests were 38 % of the waves' time, most of it two media that every ray meets; on the book-1 scene most spheres move).
 *   RT_LEAF_MEDIUM_SPHERE  a constant_medium whose boundary is a static sphere: f = cx cy cz r density, visits as usual
 *   RT_LEAF_MEDIUM_BOX     ... whose boundary is a box (rt_box.h): f = p0.x p0.y p0.z p1.x p1.y, p1.z in the last word; TWO slots:
 *                          the next one holds f[0] = density, visits in its last word
 *   RT_LEAF_MOVING_SPHERE  f = c0.x c0.y c0.z r t0, 1 / (t1 - t0) in the last word; TWO slots: the next one holds f = c1 - c0
 * Any other constant_medium (a boundary that is neither) and anything else keeps its rt_prim_kind and reads prims[prim]. */
#define RT_LEAF_MEDIUM_SPHERE 16
#define RT_LEAF_MEDIUM_BOX 17
#define RT_LEAF_MOVING_SPHERE 18

/* rt_leaf.prim of a sphere, moving sphere or rect whose material has a noise texture carries this flag: a walk's best hit then
 * says by itself that shading will ask for perlin noise, and the render kernel serves the few such lanes of a shade phase
 * with the whole wave (rtf::wave_noise_value) before any of them has fetched anything. */
#define RT_LEAF_PRIM_NOISE 0x40000000
#define RT_LEAF_PRIM_MASK  0x3fffffff
typedef struct rt_leaf {
    float    f[5];
    uint32_t kind_chain; /* rt_prim_kind | instance chain << 8 (of the shape; 0 = world space) */
    int32_t  prim;       /* index into the primitive table returned by rt_bvh_build (| RT_LEAF_PRIM_NOISE) */
    int32_t  visits;     /* times the reference's tree visits this primitive (rt_bvh.cpp); box: the bits of p1.z */
} rt_leaf;

/* What shading needs about a primitive of that table, resolved on the host: the material and its texture in ONE 32-byte record
 * indexed like the primitive table, fetched beside the primitive instead of after it (primitive -> material -> texture were
 * three dependent fetches per shaded ray).  texture = -1: `color` is the material's constant colour; else an index into
 * rt_scene_desc.textures (noise, checker, image) that shading evaluates. */
typedef struct rt_shade_rec {
    uint32_t mat_kind; /* rt_material_kind */
    int32_t  texture;
    float    p0;       /* fuzz / ref_idx */
    float    color[3];
    uint32_t tex_kind; /* rt_texture_kind of `texture` (0 when texture = -1) and, for a noise texture, its scale: what the */
    float    tex_scale;/* wave-level noise evaluation reads without fetching the texture record */
} rt_shade_rec;

/* Builds the tree over the primitives the entry array names (each once; a list that is a box is one item, the members of
 * any other list are items of their own, so a leaf is one sphere, one rect, one box or one constant_medium).  `prims`
 * receives the primitive table rt_leaf.prim indexes: a copy of d->prims in which spheres under a rigid instance chain
 * are moved to world space (chain 0; not those whose texture reads (u,v), which is defined in the sphere's own frame)
 * and boxes carry RT_PRIM_IS_BOX, their corners and their faces in canonical order (rt_box.h).
 * `depth` receives the number of inner levels (= stack levels a walk needs).
 * Returns false when it cannot (not even RT_BVH_DEPTH_LIMIT levels suffice, non-finite boxes, more than 2^24 chains); the
 * caller then has no fast path for large scenes. */
bool rt_bvh_build(const rt_scene_desc *d, std::vector<rt_bvh_node> &nodes, std::vector<rt_leaf> &leaves, std::vector<rt_prim> &prims,
                  int *depth);

#endif
