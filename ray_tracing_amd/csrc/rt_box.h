/* rt_box.h -- recognises the reference's box (src/box.h:19-31) in a scene description.  Internal to the library. */
#ifndef RT_BOX_H_
#define RT_BOX_H_

#include "../../include/rt_amd.h"

/* Internal flag on the library's own copy of a GROUP primitive (rt_bvh_build): the list is a box; f[0..2] = p0,
 * f[3..5] = p1 and the six members are stored in the order [2 * axis + side]. */
#define RT_PRIM_IS_BOX 0x100u

/* Is this list a box?  Six rects, two per axis, that span p0..p1, the one at p0 of each pair with its normal flipped
 * (outward normals), one material, no texture that reads (u,v).  face[2*axis + side] receives the index of the member
 * that is that face (axis x, y, z; side 0 at p0, 1 at p1). */
static inline bool rt_as_box(const rt_scene_desc *d, const rt_prim &g, float lo[3], float hi[3], int face[6])
{
    if (g.kind != RT_PRIM_GROUP || g.count != 6) return false;
    for (int k = 0; k < 6; ++k) face[k] = -1;
    for (int k = 0; k < 6; ++k) {
        const rt_prim &c = d->prims[g.first + k];
        int axis; /* the axis the rect is normal to */
        if (c.kind == RT_PRIM_YZ_RECT) axis = 0;
        else if (c.kind == RT_PRIM_XZ_RECT) axis = 1;
        else if (c.kind == RT_PRIM_XY_RECT) axis = 2;
        else return false;
        if (c.material != d->prims[g.first].material || (c.flags & RT_PRIM_NEEDS_UV) || c.chain != g.chain) return false;
        const int slot = 2 * axis + ((c.flags & RT_PRIM_FLIP) ? 0 : 1);
        if (face[slot] >= 0) return false;
        face[slot] = g.first + k;
    }
    for (int a = 0; a < 3; ++a) {
        lo[a] = d->prims[face[2 * a]].f[4], hi[a] = d->prims[face[2 * a + 1]].f[4];
        if (!(lo[a] < hi[a])) return false;
    }
    /* a yz rect spans (y, z), an xz rect (x, z), an xy rect (x, y): src/plane.h */
    static const int A[3] = {1, 0, 0}, B[3] = {2, 2, 1};
    for (int k = 0; k < 6; ++k) {
        const float *f = d->prims[face[k]].f;
        const int a = k / 2;
        if (f[0] != lo[A[a]] || f[1] != hi[A[a]] || f[2] != lo[B[a]] || f[3] != hi[B[a]]) return false;
    }
    return true;
}

#endif /* RT_BOX_H_ */
