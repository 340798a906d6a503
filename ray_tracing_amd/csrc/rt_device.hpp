// rt_device.hpp -- gfx950 device functions of the path tracer: vector math,
// intersection, the wave-uniform tree scan, textures, sampling and shading.
//
// Execution model.  One lane carries one path.  All lanes of a wave64 walk the
// flattened tree (include/rt_amd.h, rt_entry) TOGETHER: the entry index is a
// wave-uniform scalar, entry and primitive records are fetched with scalar loads
// into SGPRs and feed the VALU as scalar operands, and each lane only keeps the
// index at which it resumes after a missed box.  The reference's traversal order
// is fixed (left subtree, then right subtree, never front-to-back:
// src/bvh.h:128-129), which is what makes this possible; for the scenes the
// headline configurations use (tens of entries) it removes every per-lane
// address computation, stack and gather from the inner loop.
//
// Arithmetic.  Everything that decides WHICH primitive a ray hits and at what t
// repeats the reference's float operations one by one (cited per function) and
// the file is compiled with -ffp-contract=off, so primary hits agree with the
// CPU bit for bit (tests/test_gpu_parity.py).  hipcc's default IEEE-exact
// division and sqrt are kept for the same reason.
#ifndef RT_DEVICE_HPP_
#define RT_DEVICE_HPP_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rt_amd.h"
#include "../../include/rt_rng.h"
#include "rt_bvh.h"
#include "rt_order.h"

#define RT_PI 3.14159f           // src/main.cpp:54-56
#define RT_M_PI 3.14159265358979323846f // (float)M_PI, used only by get_sphere_uv (src/utils.h:61-66)
#define RT_FLT_MAX 3.402823466e+38f
#define RT_DEV __device__ __forceinline__

namespace rt {

// ---------------------------------------------------------------------------
// vec3 with the reference's operator shapes (src/vec3.h)
// ---------------------------------------------------------------------------
struct V3 {
    float x, y, z;
};
RT_DEV V3 v3(float a, float b, float c) { return V3{a, b, c}; }
RT_DEV V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_DEV V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_DEV V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_DEV V3 operator*(V3 a, float t) { return v3(t * a.x, t * a.y, t * a.z); }
RT_DEV V3 operator*(float t, V3 a) { return v3(t * a.x, t * a.y, t * a.z); }
RT_DEV V3 operator/(V3 a, float t) { return v3(a.x / t, a.y / t, a.z / t); }
RT_DEV V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
RT_DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_DEV float sqlen(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
RT_DEV float len(V3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
RT_DEV V3 unit(V3 a) { return a / len(a); }
RT_DEV V3 cross(V3 a, V3 b)
{
    return v3((a.y * b.z - a.z * b.y), (-(a.x * b.z - a.z * b.x)), (a.x * b.y - a.y * b.x));
}

struct Ray {
    V3 o, d;
    float time;
};
RT_DEV V3 at(const Ray &r, float t) { return r.o + t * r.d; }

struct Onb { // src/vec3.h:195-224
    V3 u, v, w;
    RT_DEV void build_from_w(V3 n)
    {
        w = unit(n);
        V3 a = fabsf(w.x) > 0.9f ? v3(0, 1, 0) : v3(1, 0, 0);
        v = unit(cross(w, a));
        u = cross(w, v);
    }
    RT_DEV V3 local(V3 a) const { return a.x * u + a.y * v + a.z * w; }
};

// per-path random stream, include/rt_rng.h
// rt_draw(key, pixel, n) hashes key + n * 0x9E3779B9 with the pixel index folded in; the sum is carried instead of
// recomputed.  `w` is the pixel index: a kernel that cannot spare the register between two places that draw sets it
// again (a pure function of the tile and the lane's place in it) instead of carrying it.
struct Rng {
    uint32_t x, w;
    RT_DEV void start(uint32_t key, uint32_t pixel) { x = key, w = pixel; }
    RT_DEV float next()
    {
        const uint32_t h = rt_mix32_w(x, w);
        x += 0x9E3779B9U;
        return (float)(h >> 8) * (1.0f / 16777216.0f);
    }
};

// ---------------------------------------------------------------------------
// device view of a scene (all pointers into one device allocation)
// ---------------------------------------------------------------------------
struct DevScene {
    const rt_entry *entries;
    const rt_prim *prims;
    const rt_xform *xforms;
    const rt_chain *chains;
    const rt_material *materials;
    const rt_texture *textures;
    const uint8_t *texels;
    const int32_t *lights;
    const ::rt_bvh_node *bvh; // rt_bvh.h
    const ::rt_leaf *leaves;  // that tree's leaves (rt_bvh.h), or null
    const rt_prim *fprims;    // the primitive table the leaves index (rt_bvh.h), or null: the fast per-lane path's own tree, or null
    const ::rt_shade_rec *fshade; // [like fprims] material and texture of each primitive in one record (rt_bvh.h), or null
    const int2 *ord; // rt_order.h: per inner entry {right child, flags}; only read when ordered != 0
    int32_t n_entries, n_lights;
    int32_t ordered; // the entries are a binary tree no deeper than RT_ORD_MAX_DEPTH: world_hit_ordered may be used
    rt_camera cam;
};

struct Hit {
    float t, u, v;
    V3 p, n;
    int32_t material, prim;
};

// ---------------------------------------------------------------------------
// intersection
// ---------------------------------------------------------------------------

// aabb::hit, src/aabb.h:60-83, with 1/d hoisted (same value every time) and the
// three per-axis early returns folded into one AND: a failed axis fails the test
// whatever later axes compute, and NaNs fall through the selects as in the source.
RT_DEV bool slab_test(const float *lo, const float *hi, const Ray &r, V3 inv, float t_min, float t_max)
{
    bool ok = true;
    {
        float t0 = (lo[0] - r.o.x) * inv.x, t1 = (hi[0] - r.o.x) * inv.x;
        float a = inv.x < 0.0f ? t1 : t0, b = inv.x < 0.0f ? t0 : t1;
        t_min = a > t_min ? a : t_min;
        t_max = b < t_max ? b : t_max;
        ok = ok && !(t_max <= t_min);
    }
    {
        float t0 = (lo[1] - r.o.y) * inv.y, t1 = (hi[1] - r.o.y) * inv.y;
        float a = inv.y < 0.0f ? t1 : t0, b = inv.y < 0.0f ? t0 : t1;
        t_min = a > t_min ? a : t_min;
        t_max = b < t_max ? b : t_max;
        ok = ok && !(t_max <= t_min);
    }
    {
        float t0 = (lo[2] - r.o.z) * inv.z, t1 = (hi[2] - r.o.z) * inv.z;
        float a = inv.z < 0.0f ? t1 : t0, b = inv.z < 0.0f ? t0 : t1;
        t_min = a > t_min ? a : t_min;
        t_max = b < t_max ? b : t_max;
        ok = ok && !(t_max <= t_min);
    }
    return ok;
}

// The same test, also returning the entry distance it ended with (>= the t_min passed in).
RT_DEV bool slab_test_entry(const float *lo, const float *hi, const Ray &r, V3 inv, float t_min, float t_max, float &entry)
{
    bool ok = true;
    {
        float t0 = (lo[0] - r.o.x) * inv.x, t1 = (hi[0] - r.o.x) * inv.x;
        float a = inv.x < 0.0f ? t1 : t0, b = inv.x < 0.0f ? t0 : t1;
        t_min = a > t_min ? a : t_min;
        t_max = b < t_max ? b : t_max;
        ok = ok && !(t_max <= t_min);
    }
    {
        float t0 = (lo[1] - r.o.y) * inv.y, t1 = (hi[1] - r.o.y) * inv.y;
        float a = inv.y < 0.0f ? t1 : t0, b = inv.y < 0.0f ? t0 : t1;
        t_min = a > t_min ? a : t_min;
        t_max = b < t_max ? b : t_max;
        ok = ok && !(t_max <= t_min);
    }
    {
        float t0 = (lo[2] - r.o.z) * inv.z, t1 = (hi[2] - r.o.z) * inv.z;
        float a = inv.z < 0.0f ? t1 : t0, b = inv.z < 0.0f ? t0 : t1;
        t_min = a > t_min ? a : t_min;
        t_max = b < t_max ? b : t_max;
        ok = ok && !(t_max <= t_min);
    }
    entry = t_min;
    return ok;
}

// get_sphere_uv, src/utils.h:57-68
RT_DEV void sphere_uv(V3 p, float &u, float &v)
{
    float phi = atan2f(p.z, p.x);
    float theta = asinf(p.y);
    float two_pi = (2.0f * RT_M_PI);
    float pi_over_two = (RT_M_PI / 2.0f);
    float phi_minus_pi = (phi + RT_M_PI);
    u = 1.0f - (phi_minus_pi / two_pi);
    v = (theta + pi_over_two) / RT_M_PI;
}

// sphere::hit src/sphere.h:21-59 / moving_sphere::hit src/sphere.h:147-183: t only
RT_DEV bool sphere_t(V3 center, float radius, const Ray &r, float t_min, float t_max, float &t)
{
    V3 oc = r.o - center;
    float a = dot(r.d, r.d);
    float b = dot(oc, r.d);
    float c = dot(oc, oc) - radius * radius;
    float discriminant = b * b - a * c;
    bool ok = false;
    if (discriminant > 0.0f) {
        float sq = sqrtf(discriminant);
        float t1 = (-b - sq) / a;
        float t2 = (-b + sq) / a;
        bool in1 = t1 < t_max && t1 > t_min;
        bool in2 = t2 < t_max && t2 > t_min;
        t = in1 ? t1 : t2;
        ok = in1 || in2;
    }
    return ok;
}

// xy/xz/yz_rect::hit, src/plane.h:9-29, 47-67, 109-129.  f = a0 a1 b0 b1 k; the
// kind is wave-uniform, so the axis selection costs nothing per lane.
RT_DEV bool rect_t(int kind, const float *f, const Ray &r, float t0, float t1, float &t, float &a, float &b)
{
    float ok_, dk, oa, da, ob, db;
    if (kind == RT_PRIM_XY_RECT) ok_ = r.o.z, dk = r.d.z, oa = r.o.x, da = r.d.x, ob = r.o.y, db = r.d.y;
    else if (kind == RT_PRIM_XZ_RECT) ok_ = r.o.y, dk = r.d.y, oa = r.o.x, da = r.d.x, ob = r.o.z, db = r.d.z;
    else ok_ = r.o.x, dk = r.d.x, oa = r.o.y, da = r.d.y, ob = r.o.z, db = r.d.z;
    t = (f[4] - ok_) / dk;
    a = oa + t * da;
    b = ob + t * db;
    return !(t < t0 || t > t1) && !(a < f[0] || a > f[1] || b < f[2] || b > f[3]);
}
RT_DEV V3 rect_normal(int kind)
{
    return kind == RT_PRIM_XY_RECT ? v3(0, 0, 1) : kind == RT_PRIM_XZ_RECT ? v3(0, 1, 0) : v3(1, 0, 0);
}

// translate::hit / rotate_y::hit going down, src/transforms.h:14-28, 102-130.
// `chain` is wave-uniform: the loop and every coefficient are scalar.
RT_DEV Ray chain_down(const DevScene &S, int chain, Ray r)
{
    rt_chain c = S.chains[chain];
    for (int i = 0; i < c.count; ++i) {
        rt_xform x = S.xforms[c.first + i];
        if (x.kind == RT_XFORM_TRANSLATE) {
            r.o = r.o - v3(x.a, x.b, x.c);
        } else {
            float s = x.a, co = x.b;
            V3 o = r.o, d = r.d;
            o.x = co * r.o.x - s * r.o.z;
            o.z = s * r.o.x + co * r.o.z;
            d.x = co * r.d.x - s * r.d.z;
            d.z = s * r.d.x + co * r.d.z;
            r.o = o, r.d = d;
        }
    }
    return r;
}
RT_DEV void chain_up(const DevScene &S, int chain, V3 &p, V3 &n)
{
    rt_chain c = S.chains[chain];
    for (int i = c.count - 1; i >= 0; --i) {
        rt_xform x = S.xforms[c.first + i];
        if (x.kind == RT_XFORM_TRANSLATE) {
            p = p + v3(x.a, x.b, x.c);
        } else {
            float s = x.a, co = x.b;
            V3 q = p, m = n;
            q.x = co * p.x + s * p.z;
            q.z = -s * p.x + co * p.z;
            m.x = co * n.x + s * n.z;
            m.z = -s * n.x + co * n.z;
            p = q, n = m;
        }
    }
}

// A plain primitive (sphere, moving sphere, rect) in its own space.  `pr` is read
// through a wave-uniform index.  Fills t, local p and n (flip applied), u, v.
RT_DEV bool plain_hit(const rt_prim &pr, const Ray &lr, float t_min, float t_max, bool full, float &t, V3 &p, V3 &n,
                      float &u, float &v)
{
    bool h;
    int kind = pr.kind;
    if (kind == RT_PRIM_SPHERE || kind == RT_PRIM_MOVING_SPHERE) {
        V3 c;
        float radius;
        if (kind == RT_PRIM_SPHERE) {
            c = v3(pr.f[0], pr.f[1], pr.f[2]);
            radius = pr.f[3];
        } else { // center(time), src/sphere.h:142-145
            V3 c0 = v3(pr.f[0], pr.f[1], pr.f[2]), c1 = v3(pr.f[3], pr.f[4], pr.f[5]);
            c = c0 + ((lr.time - pr.f[7]) / (pr.f[8] - pr.f[7])) * (c1 - c0);
            radius = pr.f[6];
        }
        h = sphere_t(c, radius, lr, t_min, t_max, t);
        if (full && h) {
            p = at(lr, t);
            n = (p - c) / radius;
            u = v = 0.0f;
            if (pr.flags & RT_PRIM_NEEDS_UV) sphere_uv(n, u, v); // before the flip, as sphere::hit does
        }
    } else {
        float a, b;
        h = rect_t(kind, pr.f, lr, t_min, t_max, t, a, b);
        if (full && h) {
            p = at(lr, t);
            n = rect_normal(kind);
            u = v = 0.0f;
            if (pr.flags & RT_PRIM_NEEDS_UV) {
                u = (a - pr.f[0]) / (pr.f[1] - pr.f[0]);
                v = (b - pr.f[2]) / (pr.f[3] - pr.f[2]);
            }
        }
    }
    if (full && h && (pr.flags & RT_PRIM_FLIP)) n = -n; // flip_normals::hit, src/hitable.h:27-38
    return h;
}

// Any non-medium primitive, `lr` already in the primitive's space.  Returns the
// hit in LOCAL coordinates and the index of the primitive that was hit (the
// child for groups).  hitable_list::hit semantics for groups
// (src/hitable_list.h:18-34): t_max shrinks from child to child.
RT_DEV bool simple_hit(const DevScene &S, int idx, const rt_prim &pr, const Ray &lr, float t_min, float t_max, bool on,
                       bool full, float &t, V3 &p, V3 &n, float &u, float &v, int &which)
{
    bool h = false;
    if (pr.kind == RT_PRIM_GROUP) {
        float closest = t_max;
        for (int k = 0; k < pr.count; ++k) {
            const rt_prim &cp = S.prims[pr.first + k];
            float ct, cu, cv;
            V3 cpnt, cn;
            bool ch = on && plain_hit(cp, lr, t_min, closest, full, ct, cpnt, cn, cu, cv);
            if (ch) {
                h = true;
                closest = ct;
                t = ct, which = pr.first + k;
                if (full) p = cpnt, n = cn, u = cu, v = cv;
            }
        }
    } else {
        h = on && plain_hit(pr, lr, t_min, t_max, full, t, p, n, u, v);
        which = idx;
    }
    return h;
}

// ---------------------------------------------------------------------------
// world->hit(r, t_min, t_max, rec): bvh_node::hit, src/bvh.h:123-163, as one
// wave-uniform pass over the pre-order entries.
//   * inner entry: lanes whose ray misses the box set `resume` to the entry after
//     the subtree; if no lane goes in, the whole wave jumps there.
//   * leaf entry: lanes not skipping test the primitive.  Both children of a node
//     see the same t_max in the reference and the nearer wins, the RIGHT one on a
//     tie: scanning left to right and replacing whenever !(best.t < t) is that rule.
// `alive` lanes take part; the others only follow.  rng feeds constant_medium.
// ---------------------------------------------------------------------------
RT_DEV bool world_hit(const DevScene &S, const Ray &wr, float t_min, float t_max, bool alive, Rng &rng, Hit &best)
{
    bool any = false;
    int resume = 0;
    int cur_chain = 0;
    Ray lr = wr;
    V3 inv = v3(1.0f / lr.d.x, 1.0f / lr.d.y, 1.0f / lr.d.z);
    best.t = 0.0f, best.u = best.v = 0.0f, best.p = best.n = v3(0, 0, 0), best.material = -1, best.prim = -1;

    for (int i = 0; i < S.n_entries;) {
        const rt_entry e = S.entries[i];
        bool on = alive && resume <= i;
        if (!__any(on)) { // every lane is skipping: nothing to do here
            ++i;
            continue;
        }
        int chain = e.b >> 8;
        if ((e.b & 255) == RT_ENTRY_INNER) {
            if (chain != cur_chain) {
                lr = chain_down(S, chain, wr);
                inv = v3(1.0f / lr.d.x, 1.0f / lr.d.y, 1.0f / lr.d.z);
                cur_chain = chain;
            }
            bool in = slab_test(e.lo, e.hi, lr, inv, t_min, t_max);
            if (on && !in) resume = e.a;
            i = __any(on && in) ? i + 1 : e.a;
            continue;
        }
        // leaf
        const rt_prim &pr = S.prims[e.a];
        if (pr.chain != cur_chain) {
            lr = chain_down(S, pr.chain, wr);
            inv = v3(1.0f / lr.d.x, 1.0f / lr.d.y, 1.0f / lr.d.z);
            cur_chain = pr.chain;
        }
        float t = 0.0f, u = 0.0f, v = 0.0f;
        V3 p = v3(0, 0, 0), n = v3(0, 0, 0);
        int which = e.a, material;
        bool h;
        if (pr.kind == RT_PRIM_MEDIUM) { // constant_medium::hit, src/volume.h:23-53
            const rt_prim &bp = S.prims[pr.first];
            Ray br = bp.chain == cur_chain ? lr : chain_down(S, bp.chain, wr);
            float t1 = 0.0f, t2 = 0.0f;
            int dummy;
            bool h1 = simple_hit(S, pr.first, bp, br, -RT_FLT_MAX, RT_FLT_MAX, on, false, t1, p, n, u, v, dummy);
            bool h2 = simple_hit(S, pr.first, bp, br, t1 + 0.0001f, RT_FLT_MAX, h1, false, t2, p, n, u, v, dummy);
            h = false;
            if (h2) {
                if (t1 < t_min) t1 = t_min;
                if (t1 > t_max) t1 = t_max;
                if (!(t1 >= t2)) {
                    if (t1 < 0) t1 = 0;
                    // no exit test: src/volume.h:41 assigns rec2.t = rec1.t and drops the distance
                    float hit_distance = -(1.0f / pr.f[0]) * logf(rng.next());
                    t = t1 + hit_distance / len(lr.d);
                    p = at(lr, t);
                    n = v3(1, 0, 0);
                    u = v = 0.0f;
                    h = true;
                }
            }
            material = pr.material;
        } else {
            h = simple_hit(S, e.a, pr, lr, t_min, t_max, on, true, t, p, n, u, v, which);
            material = pr.material;
            if (h && pr.kind == RT_PRIM_GROUP) material = S.prims[which].material; // members may differ
        }
        if (h && (!any || !(best.t < t))) {
            chain_up(S, cur_chain, p, n);
            best.t = t, best.u = u, best.v = v, best.p = p, best.n = n;
            best.prim = which;
            best.material = material;
            any = true;
        }
        ++i;
    }
    return any;
}

// One leaf of the tree for ONE lane (per-lane walks): the primitive `leaf` against the lane's ray.
// Keeps (lr, inv, cur_chain), the ray in the space of the last chain visited.  Returns the hit in local
// coordinates of cur_chain; the caller applies chain_up if it keeps the hit.
RT_DEV bool leaf_hit_lane(const DevScene &S, int leaf, const Ray &wr, Ray &lr, V3 &inv, int &cur_chain, float t_min, float t_max,
                          Rng &rng, float &t, float &u, float &v, V3 &p, V3 &nn, int &which, int &material)
{
    const rt_prim &pr = S.prims[leaf];
    if (pr.chain != cur_chain) {
        lr = chain_down(S, pr.chain, wr);
        inv = v3(1.0f / lr.d.x, 1.0f / lr.d.y, 1.0f / lr.d.z);
        cur_chain = pr.chain;
    }
    bool h;
    if (pr.kind == RT_PRIM_MEDIUM) { // constant_medium::hit, src/volume.h:23-53
        const rt_prim &bp = S.prims[pr.first];
        Ray br = bp.chain == cur_chain ? lr : chain_down(S, bp.chain, wr);
        float t1 = 0.0f, t2 = 0.0f;
        int dummy;
        bool h1 = simple_hit(S, pr.first, bp, br, -RT_FLT_MAX, RT_FLT_MAX, true, false, t1, p, nn, u, v, dummy);
        bool h2 = simple_hit(S, pr.first, bp, br, t1 + 0.0001f, RT_FLT_MAX, h1, false, t2, p, nn, u, v, dummy);
        h = false;
        if (h2) {
            if (t1 < t_min) t1 = t_min;
            if (t1 > t_max) t1 = t_max;
            if (!(t1 >= t2)) {
                if (t1 < 0) t1 = 0;
                float hit_distance = -(1.0f / pr.f[0]) * logf(rng.next());
                t = t1 + hit_distance / len(lr.d);
                p = at(lr, t);
                nn = v3(1, 0, 0);
                u = v = 0.0f;
                h = true;
            }
        }
        material = pr.material;
    } else {
        h = simple_hit(S, leaf, pr, lr, t_min, t_max, true, true, t, p, nn, u, v, which);
        material = pr.material;
        if (h && pr.kind == RT_PRIM_GROUP) material = S.prims[which].material;
    }
    return h;
}

// ---------------------------------------------------------------------------
// The same walk with a PER-LANE entry index, for scenes too large for the wave to
// visit together (hundreds of entries and up: with incoherent rays the union of the
// 64 lanes' subtrees is most of the tree).  Order, box tests, tie rule and the
// order of constant_medium's draws are those of world_hit(); only the mapping to
// the machine differs: each lane follows its own skip links and fetches its own
// entries (two dwordx4 loads) and primitives.  No stack is needed because the
// reference's order is fixed (left, right, both always; src/bvh.h:128-129).
// Classic while-while shape: lanes first run down boxes until each stands on a
// leaf, then the wave tests leaves together.
// ---------------------------------------------------------------------------
RT_DEV bool world_hit_lanes(const DevScene &S, const Ray &wr, float t_min, float t_max, bool alive, Rng &rng, Hit &best)
{
    bool any = false;
    const int n = S.n_entries;
    int i = alive ? 0 : n;
    int cur_chain = 0;
    Ray lr = wr;
    V3 inv = v3(1.0f / lr.d.x, 1.0f / lr.d.y, 1.0f / lr.d.z);
    best.t = 0.0f, best.u = best.v = 0.0f, best.p = best.n = v3(0, 0, 0), best.material = -1, best.prim = -1;

    while (__any(i < n)) {
        int leaf = -1;
        while (i < n) { // boxes
            const rt_entry e = S.entries[i];
            if ((e.b & 255) != RT_ENTRY_INNER) {
                leaf = e.a;
                break;
            }
            const int chain = e.b >> 8;
            if (chain != cur_chain) {
                lr = chain_down(S, chain, wr);
                inv = v3(1.0f / lr.d.x, 1.0f / lr.d.y, 1.0f / lr.d.z);
                cur_chain = chain;
            }
            i = slab_test(e.lo, e.hi, lr, inv, t_min, t_max) ? i + 1 : e.a;
        }
        if (leaf >= 0) {
            float t = 0.0f, u = 0.0f, v = 0.0f;
            V3 p = v3(0, 0, 0), nn = v3(0, 0, 0);
            int which = leaf, material;
            const bool h = leaf_hit_lane(S, leaf, wr, lr, inv, cur_chain, t_min, t_max, rng, t, u, v, p, nn, which, material);
            if (h && (!any || !(best.t < t))) {
                chain_up(S, cur_chain, p, nn);
                best.t = t, best.u = u, best.v = v, best.p = p, best.n = nn;
                best.prim = which;
                best.material = material;
                any = true;
            }
            ++i;
        }
    }
    return any;
}

// ---------------------------------------------------------------------------
// The per-lane walk, ORDERED and CULLED, with the reference's results.
//
// bvh_node::hit visits both children with the same t_max and keeps the nearer hit, the right one on a tie
// (src/bvh.h:128-151).  The winner of the whole tree is therefore the hit with the smallest t and, among
// equal t, the one latest in depth-first order -- a rule that does not depend on the order of the visits.
// So a lane may (a) visit the nearer child first, (b) skip a subtree whose box the ray enters behind the
// best hit so far, and still end with the reference's hit, bit for bit:
//   * a box is entered iff the reference's aabb::hit accepts it (same arithmetic, slab_test_entry) AND its
//     entry distance is not beyond best.t by more than a margin.  Box and rect distances carry a few ulps of
//     relative error; a sphere's t can be off by ~sqrt(eps) of the sphere's distance for grazing hits, so
//     the margin is 2^-9 of (best.t + the box's largest side / |d|).  A skipped box cannot hold a winner.
//   * primitives see the reference's t_min, t_max (not best.t): which hits exist is unchanged;
//   * ties go to the larger entry index (= later in depth-first order);
//   * subtrees that hold a constant_medium are walked left to right and never skipped: the medium draws a
//     random number per visit (src/volume.h:43) and those draws must happen in the reference's order.
// (A hit whose t is NaN -- a ray lying exactly in a rect's plane -- is ordered by the reference through
// `!(best.t < t)`; here it never replaces a hit.  It needs an exactly zero direction component and an
// origin exactly on the plane.)
// Pending far children live in a per-lane stack in LDS: stack[level * 64], at most RT_ORD_MAX_DEPTH levels
// (rt_order.cpp checks the depth of the tree on the host).
// ---------------------------------------------------------------------------
RT_DEV bool world_hit_ordered(const DevScene &S, const Ray &wr, float t_min, float t_max, bool alive, Rng &rng, Hit &best,
                              int *stack)
{
    bool any = false;
    int best_ord = -1;
    int sp = 0;
    int cur = alive ? 0 : -1;
    int cur_chain = 0;
    Ray lr = wr;
    V3 inv = v3(1.0f / lr.d.x, 1.0f / lr.d.y, 1.0f / lr.d.z);
    const float inv_len = 1.0f / len(wr.d); // chains are rigid: |d| is the same in every space
    best.t = 0.0f, best.u = best.v = 0.0f, best.p = best.n = v3(0, 0, 0), best.material = -1, best.prim = -1;

    while (__any(cur >= 0)) {
        int leaf = -1, leaf_ord = -1;
        while (cur >= 0) { // boxes
            const rt_entry e = S.entries[cur];
            if ((e.b & 255) != RT_ENTRY_INNER) {
                leaf = e.a, leaf_ord = cur;
                break;
            }
            const int chain = e.b >> 8;
            if (chain != cur_chain) {
                lr = chain_down(S, chain, wr);
                inv = v3(1.0f / lr.d.x, 1.0f / lr.d.y, 1.0f / lr.d.z);
                cur_chain = chain;
            }
            const int2 o = S.ord[cur];
            float entry;
            bool in = slab_test_entry(e.lo, e.hi, lr, inv, t_min, t_max, entry);
            if (any && !(o.y & RT_ORD_HAS_MEDIUM)) {
                const float side = fmaxf(fmaxf(e.hi[0] - e.lo[0], e.hi[1] - e.lo[1]), e.hi[2] - e.lo[2]);
                in = in && !(entry > best.t + 0.001953125f * (best.t + side * inv_len));
            }
            if (!in) {
                cur = sp > 0 ? stack[--sp * 64] : -1;
                continue;
            }
            const int axis = o.y & RT_ORD_AXIS_MASK;
            const float da = axis == 0 ? lr.d.x : axis == 1 ? lr.d.y : lr.d.z;
            const bool left_first =
                (o.y & (RT_ORD_HAS_MEDIUM | RT_ORD_NO_PREFERENCE)) != 0 || ((da >= 0.0f) == ((o.y & RT_ORD_LEFT_LOWER) != 0));
            stack[sp++ * 64] = left_first ? o.x : cur + 1;
            cur = left_first ? cur + 1 : o.x;
        }
        if (leaf >= 0) {
            float t = 0.0f, u = 0.0f, v = 0.0f;
            V3 p = v3(0, 0, 0), nn = v3(0, 0, 0);
            int which = leaf, material;
            const bool h = leaf_hit_lane(S, leaf, wr, lr, inv, cur_chain, t_min, t_max, rng, t, u, v, p, nn, which, material);
            if (h && (!any || t < best.t || (t == best.t && leaf_ord > best_ord))) {
                chain_up(S, cur_chain, p, nn);
                best.t = t, best.u = u, best.v = v, best.p = p, best.n = nn;
                best.prim = which;
                best.material = material;
                best_ord = leaf_ord;
                any = true;
            }
            cur = sp > 0 ? stack[--sp * 64] : -1;
        }
    }
    return any;
}

// Entry counts up to this use the wave-uniform walk, larger scenes the per-lane walk.
#define RT_UNIFORM_WALK_MAX 64

// ---------------------------------------------------------------------------
// textures: src/texture.h, src/perlin.h
// ---------------------------------------------------------------------------
// the 256 gradient vectors perlin::noise indexes (src/perlin.h:114-115); see tools/gen_perlin_table.py
static __device__ const float4 k_perlin_gradients[256] = {
#include "rt_perlin_table.inc"
};

RT_DEV float perlin_noise(const DevScene &S, V3 p) // src/perlin.h:54-121
{
    float fx = floorf(p.x), fy = floorf(p.y), fz = floorf(p.z);
    float u = p.x - fx, v = p.y - fy, w = p.z - fz;
    int i = (int)fx, j = (int)fy, k = (int)fz;
    float uu = (u * u * (3.0f - 2.0f * u)), vv = (v * v * (3.0f - 2.0f * v)), ww = (w * w * (3.0f - 2.0f * w));
    float accum = 0.0f;
#pragma unroll
    for (int di = 0; di < 2; ++di)
#pragma unroll
        for (int dj = 0; dj < 2; ++dj)
#pragma unroll
            for (int dk = 0; dk < 2; ++dk) {
                uint32_t h = 1619u * (uint32_t)(i + di) + 31337u * (uint32_t)(j + dj) + 6971u * (uint32_t)(k + dk) + 1013u;
                int index = (int)h;
                index ^= (index >> 8);
                index &= 0xff;
                float4 g = k_perlin_gradients[index];
                V3 wv = v3(u - di, v - dj, w - dk);
                accum += (di * uu + (1.0f - di) * (1.0f - uu)) * (dj * vv + (1.0f - dj) * (1.0f - vv)) *
                         (dk * ww + (1.0f - dk) * (1.0f - ww)) * dot(v3(g.x, g.y, g.z), wv);
            }
    return accum;
}
RT_DEV float perlin_turb(const DevScene &S, V3 p) // src/perlin.h:123-136, depth 7
{
    float accum = 0.0f, weight = 1.0f;
    for (int i = 0; i < 7; ++i) {
        accum += weight * perlin_noise(S, p);
        weight *= 0.5f;
        p = p * 2.0f;
    }
    return fabsf(accum);
}

RT_DEV V3 tex_value(const DevScene &S, int idx, float u, float v, V3 p)
{
    rt_texture t = S.textures[idx];
    for (int guard = 0; guard < 8 && t.kind == RT_TEX_CHECKER; ++guard) { // src/texture.h:26-37
        float sines = sinf(10.0f * p.x) * sinf(10.0f * p.y) * sinf(10.0f * p.z);
        t = S.textures[sines < 0.0f ? t.a : t.b];
    }
    if (t.kind == RT_TEX_CONSTANT) return v3(t.color[0], t.color[1], t.color[2]);
    if (t.kind == RT_TEX_NOISE) // src/texture.h:47-53
        return v3(1.0f, 1.0f, 1.0f) * 0.5f * (1.0f + sinf(t.scale * p.z + 10.0f * perlin_turb(S, t.scale * p)));
    if (t.kind == RT_TEX_IMAGE) { // src/texture.h:63-77
        int i = (int)(u * (float)t.nx);
        int j = (int)((1.0f - v) * (float)t.ny - 0.001f);
        i = i < 0 ? 0 : i, j = j < 0 ? 0 : j;
        i = i > t.nx - 1 ? t.nx - 1 : i, j = j > t.ny - 1 ? t.ny - 1 : j;
        const uint8_t *d = S.texels + t.offset + 3 * i + 3 * t.nx * j;
        return v3((float)d[0] / 255.0f, (float)d[1] / 255.0f, (float)d[2] / 255.0f);
    }
    return v3(0, 0, 0); // a checker nested deeper than 8 levels
}

// ---------------------------------------------------------------------------
// sampling helpers.  Draw order follows the oracle (oracle/rt_oracle.cpp), which
// follows g++'s right-to-left evaluation of the reference's constructor arguments.
// ---------------------------------------------------------------------------
RT_DEV V3 random_in_unit_sphere(Rng &rng) // src/utils.h:32-42
{
    V3 p;
    do {
        float c = rng.next(), b = rng.next(), a = rng.next();
        p = 2.0f * v3(a, b, c) - v3(1.0f, 1.0f, 1.0f);
    } while (sqlen(p) >= 1.0f);
    return p;
}
RT_DEV V3 random_in_unit_disk(Rng &rng) // src/utils.h:44-55
{
    V3 p;
    do {
        float b = rng.next(), a = rng.next();
        p = 2.0f * v3(a, b, 0.0f) - v3(1.0f, 1.0f, 0.0f);
    } while (dot(p, p) >= 1.0f);
    return p;
}
RT_DEV V3 random_cosine_direction(Rng &rng) // src/vec3.h:173-182 (x and y carry the factor 2)
{
    float r1 = rng.next();
    float r2 = rng.next();
    float z = sqrtf(1.0f - r2);
    float phi = 2.0f * RT_PI * r1;
    float s, c;
    sincosf(phi, &s, &c);
    float x = c * 2.0f * sqrtf(r2);
    float y = s * 2.0f * sqrtf(r2);
    return v3(x, y, z);
}
RT_DEV float schlick(float cosine, float ref_idx) // src/utils.h:4-9; powf(x,5) written as products
{
    float r0 = (1 - ref_idx) / (1 + ref_idx);
    r0 = r0 * r0;
    float x = 1 - cosine, x2 = x * x;
    return r0 + (1 - r0) * (x2 * x2 * x);
}
RT_DEV V3 reflect(V3 v, V3 n) { return v - 2.0f * dot(v, n) * n; }
RT_DEV bool refract(V3 v, V3 n, float ni_over_nt, V3 &refracted) // src/utils.h:16-30
{
    V3 uv = unit(v);
    float dt = dot(uv, n);
    float discriminant = 1.0f - ni_over_nt * ni_over_nt * (1.0f - dt * dt);
    if (discriminant > 0.0f) {
        refracted = ni_over_nt * (uv - n * dt) - n * sqrtf(discriminant);
        return true;
    }
    return false;
}

// important_hitables: hitable_list::pdf_value / random, src/hitable_list.h:67-88,
// over xz_rect (src/plane.h:76-98) and sphere (src/sphere.h:68-103).  The light
// index is wave-uniform in pdf_value (a loop over all lights) and per-lane in random.
RT_DEV float light_pdf_value(const rt_prim &pr, V3 o, V3 v)
{
    Ray r{o, v, 0.0f};
    if (pr.kind == RT_PRIM_XZ_RECT) {
        float t, a, b;
        if (rect_t(RT_PRIM_XZ_RECT, pr.f, r, 0.0001f, RT_FLT_MAX, t, a, b)) {
            float area = (pr.f[1] - pr.f[0]) * (pr.f[3] - pr.f[2]);
            float distance_squared = t * t * sqlen(v);
            float cosine = fabsf(dot(v, v3(0, 1, 0)) / len(v));
            return distance_squared / (cosine * area);
        }
        return 0.0f;
    }
    V3 center = v3(pr.f[0], pr.f[1], pr.f[2]);
    float radius = pr.f[3], t;
    if (sphere_t(center, radius, r, 0.001f, RT_FLT_MAX, t)) {
        float cos_theta_max = sqrtf(1.0f - radius * radius / sqlen(center - o));
        float solid_angle = 2.0f * RT_PI * (1.0f - cos_theta_max);
        return 1.0f / solid_angle;
    }
    return 0.0f;
}
RT_DEV float lights_pdf_value(const DevScene &S, V3 o, V3 v)
{
    float weight = 1.0f / S.n_lights;
    float sum = 0.0f;
    for (int i = 0; i < S.n_lights; ++i) {
        int l = S.lights[i];
        float pdf = l < 0 ? 0.0f : light_pdf_value(S.prims[l], o, v); // src/hitable.h:19 for the rest
        sum += weight * pdf;
    }
    return sum;
}
RT_DEV V3 lights_random(const DevScene &S, V3 o, Rng &rng)
{
    int index = (int)(rng.next() * S.n_lights);
    if (index == S.n_lights) index--;
    int l = S.lights[index];
    if (l < 0) return v3(1, 0, 0); // src/hitable.h:20
    const rt_prim pr = S.prims[l];
    if (pr.kind == RT_PRIM_XZ_RECT) {
        float rz = rng.next(), rx = rng.next();
        V3 random_point = v3(pr.f[0] + rx * (pr.f[1] - pr.f[0]), pr.f[4], pr.f[2] + rz * (pr.f[3] - pr.f[2]));
        return random_point - o;
    }
    V3 center = v3(pr.f[0], pr.f[1], pr.f[2]);
    float radius = pr.f[3];
    V3 direction = center - o;
    float distance_squared = sqlen(direction);
    Onb uvw;
    uvw.build_from_w(direction);
    float r1 = rng.next(); // random_to_sphere, src/sphere.h:85-94
    float r2 = rng.next();
    float z = 1.0f + r2 * (sqrtf(1.0f - radius * radius / distance_squared) - 1.0f);
    float phi = 2.0f * RT_PI * r1;
    float s, c;
    sincosf(phi, &s, &c);
    float x = c * sqrtf(1.0f - z * z);
    float y = s * sqrtf(1.0f - z * z);
    return uvw.local(v3(x, y, z));
}

// cosine_pdf::value, src/pdf.h:13-24
RT_DEV float cosine_pdf_value(const Onb &uvw, V3 dir)
{
    float cosine = dot(unit(dir), uvw.w);
    return cosine > 0.0f ? cosine / RT_PI : 0.0f;
}

// camera::get_ray, src/camera.h:54-68
RT_DEV Ray camera_ray(const rt_camera &c, float s, float t, V3 disk, float time01)
{
    V3 cu = v3(c.u[0], c.u[1], c.u[2]), cv = v3(c.v[0], c.v[1], c.v[2]);
    V3 origin = v3(c.origin[0], c.origin[1], c.origin[2]);
    V3 llc = v3(c.lower_left_corner[0], c.lower_left_corner[1], c.lower_left_corner[2]);
    V3 hor = v3(c.horizontal[0], c.horizontal[1], c.horizontal[2]), ver = v3(c.vertical[0], c.vertical[1], c.vertical[2]);
    V3 rd = c.lens_radius * disk;
    V3 offset = cu * rd.x + cv * rd.y;
    V3 oo = origin + offset;
    float time = c.time0 + time01 * (c.time1 - c.time0);
    return Ray{oo, (llc + s * hor + t * ver) - oo, time};
}

// ---------------------------------------------------------------------------
// One bounce of color(), src/main.cpp:91-127, in throughput form.
// The recursion  emitted + albedo*spdf*color(next)/pdf  (diffuse) and
// albedo*color(next) (specular, no emitted term) unrolls to: radiance = T * emitted at
// the hit that ends the path, T accumulating albedo*spdf/pdf or albedo.  Only
// diffuse_light emits and it never scatters, so nothing is lost by adding emission
// only when the path ends.  Returns true if the path continues along `r`.
//   L    : set to the path's radiance when it ends (the caller applies de_nan)
// A weight of exactly zero ends the path early: every later factor multiplies it.
// ---------------------------------------------------------------------------
RT_DEV bool shade(const DevScene &S, bool hit_any, const Hit &h, Ray &r, V3 &T, int &depth, int max_depth, Rng &rng, V3 &L)
{
    if (!hit_any) { // src/main.cpp:123-126: no sky
        L = v3(0, 0, 0);
        return false;
    }
    const rt_material m = S.materials[h.material];
    if (m.kind == RT_MAT_DIFFUSE_LIGHT) { // emits from both faces, never scatters (src/material.h:158-174)
        L = T * tex_value(S, m.texture, h.u, h.v, h.p);
        return false;
    }
    if (!(depth < max_depth)) { // src/main.cpp:98
        L = v3(0, 0, 0);
        return false;
    }
    ++depth;
    if (m.kind == RT_MAT_LAMBERTIAN) { // src/material.h:24-44, src/main.cpp:105-116, src/pdf.h
        V3 albedo = tex_value(S, m.texture, h.u, h.v, h.p);
        Onb uvw;
        uvw.build_from_w(h.n);
        V3 dir;
        if (rng.next() < 0.5f) dir = unit(lights_random(S, h.p, rng));
        else dir = unit(uvw.local(random_cosine_direction(rng)));
        float cos_val = cosine_pdf_value(uvw, dir);
        float pdf_val = 0.5f * lights_pdf_value(S, h.p, dir) + 0.5f * cos_val;
        float sc_cos = dot(h.n, dir);
        if (sc_cos < 0.0f) sc_cos = 0.0f;
        float spdf = sc_cos / RT_PI;
        T = T * (albedo * spdf) / pdf_val;
        r = Ray{h.p, dir, r.time};
        // 0/0 here is the NaN the reference's de_nan zeroes (src/main.cpp:234); a zero weight is final
        if (!(T.x == T.x) || !(T.y == T.y) || !(T.z == T.z) || (T.x == 0.0f && T.y == 0.0f && T.z == 0.0f)) {
            L = v3(0, 0, 0);
            return false;
        }
        return true;
    }
    if (m.kind == RT_MAT_METAL) { // src/material.h:65-78; at fuzz 0 the sphere sample is not drawn (include/rt_rng.h)
        V3 reflected = reflect(unit(r.d), h.n);
        V3 fuzz = v3(0, 0, 0);
        if (m.p0 != 0.0f) fuzz = m.p0 * random_in_unit_sphere(rng);
        T = T * tex_value(S, m.texture, h.u, h.v, h.p);
        r = Ray{h.p, reflected + fuzz, r.time};
        L = v3(0, 0, 0);
        return !(T.x == 0.0f && T.y == 0.0f && T.z == 0.0f); // a zero weight is final (a black albedo)
    }
    if (m.kind == RT_MAT_DIELECTRIC) { // src/material.h:92-145
        V3 outward_normal;
        V3 reflected = reflect(unit(r.d), h.n);
        float ni_over_nt, reflect_prob, cosine, ref_idx = m.p0;
        V3 refracted = v3(0, 0, 0);
        if (dot(r.d, h.n) > 0.0f) {
            outward_normal = -h.n;
            ni_over_nt = ref_idx;
            cosine = ref_idx * dot(r.d, h.n) / len(r.d);
        } else {
            outward_normal = h.n;
            ni_over_nt = 1.0f / ref_idx;
            cosine = -dot(r.d, h.n) / len(r.d);
        }
        if (refract(r.d, outward_normal, ni_over_nt, refracted)) reflect_prob = schlick(cosine, ref_idx);
        else reflect_prob = 1.0f;
        r = Ray{h.p, rng.next() < reflect_prob ? reflected : refracted, r.time};
        return true; // albedo (1,1,1)
    }
    // isotropic, src/material.h:187-196: new ray has time 0
    V3 dir = random_in_unit_sphere(rng);
    T = T * tex_value(S, m.texture, h.u, h.v, h.p);
    r = Ray{h.p, dir, 0.0f};
    L = v3(0, 0, 0);
    return !(T.x == 0.0f && T.y == 0.0f && T.z == 0.0f); // black smoke (cornell_box_volumes): the path carries nothing further
}

} // namespace rt
#endif // RT_DEVICE_HPP_
