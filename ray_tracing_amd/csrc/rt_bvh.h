/* rt_bvh.h -- the GPU-side tree of the fast per-lane path (rt_fastlanes.hpp).
 *
 * The reference's tree (src/bvh.h:79-121: random axis, median split, boxes stored in the node itself, visited
 * left-right-both) is kept as it is for the exact path.  For the fast path the library builds a second tree over
 * the SAME primitives: world-space boxes (instance chains folded in), binned surface-area-heuristic splits, and
 * nodes that hold the boxes of BOTH children, so one 64-byte fetch decides two boxes and a leaf needs no node of
 * its own.  The closest hit over a set of primitives does not depend on the tree that finds it (equal-distance
 * ties aside), so results are those of the reference's tree up to rounding; tests compare the two.
 * Internal to the library; not part of the ABI. */
#ifndef RT_BVH_H_
#define RT_BVH_H_

#include <stdint.h>

#include <vector>

#include "../../include/rt_amd.h"

/* child reference: >= 0 inner node index; < 0 leaf: ~ref is an rt_prim index; RT_BVH_NONE: no child */
#define RT_BVH_NONE ((int32_t)0x80000000)
#define RT_BVH_MAX_DEPTH 24 /* pending far children per lane: the stack in LDS */

typedef struct rt_bvh_node {
    float   lo0[3]; int32_t ref0;
    float   hi0[3]; int32_t ref1;
    float   lo1[3]; int32_t pad0;
    float   hi1[3]; int32_t pad1;
} rt_bvh_node; /* 64 bytes: four dwordx4 loads from one or two cache lines */

/* Builds the tree over the primitives the entry array names (each once; the members of a list -- box faces -- are
 * items of their own, so a leaf is always one sphere, one rect or one constant_medium).  `prims` receives the
 * primitive table the tree's leaf references index: a copy of d->prims in which spheres under a rigid instance chain
 * are moved to world space (chain 0; not those whose texture reads (u,v), which is defined in the sphere's own frame).
 * Returns false when it cannot (a depth bound it could not keep, non-finite boxes); the caller then has no fast
 * path for large scenes. */
bool rt_bvh_build(const rt_scene_desc *d, std::vector<rt_bvh_node> &nodes, std::vector<rt_prim> &prims, int *depth);

#endif
