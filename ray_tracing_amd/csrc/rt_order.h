/* rt_order.h -- side table for the ordered walk of the reference's tree (rt_device.hpp, world_hit_ordered).
 *
 * The entry array (include/rt_amd.h) is the reference's bvh_node tree in pre-order; every inner entry has exactly
 * two children (src/bvh.h:79-121; a one-primitive node holds the primitive twice).  For the ordered walk each
 * inner entry i gets two words:
 *     right[i]  index of its right child (the left child is i + 1)
 *     flags[i]  bits 0-1  axis along which the children's centres are furthest apart
 *               bit  2    the LEFT child is the lower one on that axis
 *               bit  3    the subtree holds a constant_medium: walk it in the reference's order, never cull it
 *                         (the medium draws a random number per visit, src/volume.h:43)
 *               bit  4    no preference (children live in different instance spaces): left first
 * Internal to the library; not part of the ABI. */
#ifndef RT_ORDER_H_
#define RT_ORDER_H_

#include <vector>

#include "../../include/rt_amd.h"

#define RT_ORD_AXIS_MASK 3
#define RT_ORD_LEFT_LOWER 4
#define RT_ORD_HAS_MEDIUM 8
#define RT_ORD_NO_PREFERENCE 16
#define RT_ORD_MAX_DEPTH 32 /* far children a lane may have pending: the per-lane stack in LDS */

/* Fills out[2*i] = right, out[2*i+1] = flags for every entry (zeros for leaves) and *max_depth with the depth of
 * the tree.  Returns false if the entries are not a binary tree in pre-order or the tree is deeper than
 * RT_ORD_MAX_DEPTH; the caller then keeps the stackless walk. */
bool rt_order_build(const rt_scene_desc *d, std::vector<int32_t> &out, int *max_depth);

#endif
