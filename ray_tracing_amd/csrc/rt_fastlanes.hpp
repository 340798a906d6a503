// rt_fastlanes.hpp -- the fast render path for scenes that do NOT fit the record table (rt_records.h):
// hundreds to millions of primitives, media, moving spheres, textures.
//
// Same algorithm and same random-draw order per primitive as the exact path, different machine mapping and
// arithmetic, like rt_fast.hpp for small scenes:
//   * every lane walks a tree on its own -- the library's tree over the same primitives (rt_bvh.h: world-space
//     boxes, surface-area splits, both children's boxes in one 32-byte node), nearer child first, a box entered
//     only if the ray reaches it before the best hit so far, pending children in a per-lane stack in LDS.
//     The closest hit over a set of primitives does not depend on the tree or on the order of the visits
//     (rt_device.hpp, world_hit_ordered, explains why); which of two hits at EQUAL distance wins may differ
//     here, and a constant_medium that lies behind the best hit is not visited, so it draws no random number:
//     its hit would have lost anyway, and the draws are independent, so the image has the same distribution
//     (tests compare the two paths hit by hit and statistically).
//   * during the walk a lane keeps (t, primitive index); point, normal, (u,v) and material are built once per
//     ray from the winner.
//   * hardware-rate reciprocal / sqrt / log, FMA contraction, 1/d hoisted.
#ifndef RT_FASTLANES_HPP_
#define RT_FASTLANES_HPP_

#include <hip/hip_fp16.h>

#include "rt_box.h"
#include "rt_bvh.h"
#include "rt_fast.hpp"

namespace rtf {
#pragma clang fp contract(fast)

RTF_DEV V3 rcp3(V3 d) { return v3(rcp(d.x), rcp(d.y), rcp(d.z)); }

// b and the discriminant of sphere::hit (src/sphere.h:26-32) with the reference's operations one by one (products and sums
// rounded separately, in its order).  WHETHER a ray hits a small sphere far away is decided by the sign of b*b - a*c, a
// difference of two numbers that agree in their first five or six digits (a sphere of radius 0.25 seen from 200 units
// away): with fused operations the sign comes out differently on a few per cent of the grazing rays -- no less valid, but
// not the reference's answer (the 1M-sphere scene: 1.7 % of camera rays found another sphere).  The roots themselves keep
// the hardware-rate sqrt and reciprocal: an ulp in t only matters where two surfaces coincide.
RTF_DEV void sphere_disc(V3 oc, V3 d, float dd, float radius, float &b, float &disc)
{
    b = ref_dot(oc, d);
    const float cc = sub_rn(ref_dot(oc, oc), mul_rn(radius, radius));
    disc = sub_rn(mul_rn(b, b), mul_rn(dd, cc));
}

// sphere / moving_sphere / rect in its own space: t only.  src/sphere.h:21-59, 147-183; src/plane.h:9-29, 47-67, 109-129
RTF_DEV bool plain_t(const rt_prim &pr, const Ray &lr, V3 inv, float dd, float inv_dd, float t_min, float t_max, float &t)
{
    const int kind = pr.kind;
    if (kind <= RT_PRIM_MOVING_SPHERE) {
        V3 c;
        float radius;
        if (kind == RT_PRIM_SPHERE) {
            c = v3(pr.f[0], pr.f[1], pr.f[2]);
            radius = pr.f[3];
        } else { // center(time), src/sphere.h:142-145
            const float k = (lr.time - pr.f[7]) * rcp(pr.f[8] - pr.f[7]);
            c = v3(fmaf(k, pr.f[3] - pr.f[0], pr.f[0]), fmaf(k, pr.f[4] - pr.f[1], pr.f[1]), fmaf(k, pr.f[5] - pr.f[2], pr.f[2]));
            radius = pr.f[6];
        }
        const V3 oc = v3(lr.o.x - c.x, lr.o.y - c.y, lr.o.z - c.z);
        float b, disc;
        sphere_disc(oc, lr.d, dd, radius, b, disc);
        const float sq = fsqrt(disc);
        const float t1 = (-b - sq) * inv_dd, t2 = (-b + sq) * inv_dd;
        const bool in1 = (t1 < t_max) & (t1 > t_min), in2 = (t2 < t_max) & (t2 > t_min);
        t = in1 ? t1 : t2;
        return (disc > 0.0f) & (in1 | in2);
    }
    const bool xy = kind == RT_PRIM_XY_RECT, xz = kind == RT_PRIM_XZ_RECT;
    const float ok = xy ? lr.o.z : xz ? lr.o.y : lr.o.x, ik = xy ? inv.z : xz ? inv.y : inv.x;
    const float oa = (xy | xz) ? lr.o.x : lr.o.y, da = (xy | xz) ? lr.d.x : lr.d.y;
    const float ob = xy ? lr.o.y : lr.o.z, db = xy ? lr.d.y : lr.d.z;
    t = (pr.f[4] - ok) * ik;
    const float a = fmaf(t, da, oa), b = fmaf(t, db, ob);
    return (t >= t_min) & (t <= t_max) & (a >= pr.f[0]) & (a <= pr.f[1]) & (b >= pr.f[2]) & (b <= pr.f[3]);
}

// a box (rt_box.h) against the ray of its chain: where the ray enters and leaves the intersection of the three slabs.
// The nearest of the six faces hit inside a range is tn if it lies in the range, else tf (src/box.h, hitable_list::hit).
RTF_DEV bool box_range(V3 lo, V3 hi, const Ray &lr, V3 inv, float &tn, float &tf)
{
    const float x0 = (lo.x - lr.o.x) * inv.x, x1 = (hi.x - lr.o.x) * inv.x;
    const float y0 = (lo.y - lr.o.y) * inv.y, y1 = (hi.y - lr.o.y) * inv.y;
    const float z0 = (lo.z - lr.o.z) * inv.z, z1 = (hi.z - lr.o.z) * inv.z;
    tn = fmax3(fminf(x0, x1), fminf(y0, y1), fminf(z0, z1));
    tf = fmin3(fmaxf(x0, x1), fmaxf(y0, y1), fmaxf(z0, z1));
    return tn <= tf;
}

// a plain primitive or a list (box): hitable_list::hit, src/hitable_list.h:18-34
RTF_DEV bool simple_t(const DevScene &S, int idx, const rt_prim &pr, const Ray &lr, V3 inv, float dd, float inv_dd, float t_min,
                      float t_max, float &t, int &which)
{
    if (pr.kind != RT_PRIM_GROUP) {
        which = idx;
        return plain_t(pr, lr, inv, dd, inv_dd, t_min, t_max, t);
    }
    bool h = false;
    float closest = t_max;
    for (int k = 0; k < pr.count; ++k) {
        float ct;
        if (plain_t(S.fprims[pr.first + k], lr, inv, dd, inv_dd, t_min, closest, ct)) {
            h = true;
            closest = ct;
            t = ct, which = pr.first + k;
        }
    }
    return h;
}

// ---------------------------------------------------------------------------
// world->hit() over the library's own tree (rt_bvh.h), as a per-lane state machine.
// A lane's walk is (cur, sp, best_t, best_prim): cur >= 0 an inner node to open, cur < 0 a leaf to test
// (~cur is the rt_prim index), RT_BVH_NONE when the walk is over.  The steps below advance the lanes that are in
// the matching state; the callers decide how to interleave them (hit_bvh: one ray per lane until every lane is
// done; render_fastlanes_kernel: lanes that finish are shaded and restarted while the others keep walking).
// Boxes are in world space; a leaf under an instance chain takes the ray down its chain
// (src/transforms.h:17, 107-110) when the lane gets there.
// ---------------------------------------------------------------------------
struct Walk {
    int cur, sp;
    float best_t;
    int best_prim; // a MEDIUM primitive for a hit inside a constant_medium
    V3 winv, noi;  // 1/d and -o/d: t = lo * winv + noi, one FMA per plane.  Where 1/d overflows the two terms can
                   // cancel to NaN; min/max drop a NaN, which only ever makes a box easier to enter.
    uint32_t rx, ry, rz; // 16 where the direction is negative on that axis, else 0: how far walk_node rotates a node's plane
                         // pairs so that the plane met first is in the low half (rt_bvh.h)
    int tos;             // `int *` stacks: the top of the stack, in a register (walk_push / walk_pop below)
};
// what a walk keeps about its ray (also made again when a walk under way moves to another lane)
RTF_DEV void walk_ray_constants(Walk &w, const Ray &wr)
{
    w.winv = rcp3(wr.d);
    w.noi = v3(-wr.o.x * w.winv.x, -wr.o.y * w.winv.y, -wr.o.z * w.winv.z);
    w.rx = (__float_as_uint(w.winv.x) >> 31) << 4, w.ry = (__float_as_uint(w.winv.y) >> 31) << 4, w.rz = (__float_as_uint(w.winv.z) >> 31) << 4;
}
RTF_DEV void walk_start(Walk &w, const Ray &wr, float t_max, bool go)
{
    w.cur = go ? 0 : RT_BVH_NONE;
    w.sp = 0;
    w.tos = RT_BVH_NONE;
    w.best_t = t_max;
    w.best_prim = -1;
    walk_ray_constants(w, wr);
}
// Where a lane keeps its pending children.  The walk only pushes and pops, so the stack is a policy:
//   int *                one dword per level, [level][lane] of the lane's wave (the lane's place in its wave never changes)
//   rtf::HomeStack16     16 bits per level, [level][home] of the block: the stack stays where it is when the PATH moves to
//                        another lane (render_fastlanes_block_kernel: `home` travels with the path); trees of fewer than
//                        32768 nodes and leaves
RTF_DEV void stack_push(int *stack, int &sp, int ref) { stack[sp++ * 64] = ref; }
RTF_DEV int stack_pop(const int *stack, int &sp) { return sp > 0 ? stack[--sp * 64] : RT_BVH_NONE; }
struct HomeStack16 {
    short *base; // the block's stacks + the path's home
    int stride;  // homes per level (threads of the block)
};
RTF_DEV void stack_push(HomeStack16 st, int &sp, int ref) { st.base[sp++ * st.stride] = (short)ref; }
RTF_DEV int stack_pop(HomeStack16 st, int &sp) { return sp > 0 ? (int)st.base[--sp * st.stride] : RT_BVH_NONE; }
// The walk's own push and pop.  With an `int *` stack the TOP of the stack lives in a register (w.tos): a pop hands it over at once
// and refills it from LDS while the popped node is being fetched from memory -- read straight from LDS, every pop put an LDS
// round trip in front of that fetch.  What the first push writes to LDS is the empty top, RT_BVH_NONE, so the stack's floor is
// its own sentinel and the levels needed stay the same (the entry in the register is the one the sentinel takes the place of).
// A pop from an empty stack returns RT_BVH_NONE and leaves w.tos to be reset by the next walk_start.
RTF_DEV void walk_push(Walk &w, int *stack, int ref)
{
    stack[w.sp++ * 64] = w.tos;
    w.tos = ref;
}
RTF_DEV int walk_pop(Walk &w, const int *stack)
{
    int r = w.tos;
    asm volatile("" : "+v"(r)); // (the copy is made before the read is issued, so that the read can land in w.tos's register and nothing waits for it here)
    w.sp = (int)__builtin_elementwise_sub_sat((unsigned)w.sp, 1u);
    w.tos = stack[w.sp * 64];
    return r;
}
RTF_DEV void walk_push(Walk &w, HomeStack16 stack, int ref) { stack_push(stack, w.sp, ref); }
RTF_DEV int walk_pop(Walk &w, HomeStack16 stack) { return stack_pop(stack, w.sp); }

// t of the plane stored as a half in the low / high 16 bits of w: ONE v_fma_mix_f32 (half * float + float)
RTF_DEV float plane_lo(uint32_t w, float inv, float noi) { return fmaf(__half2float(__ushort_as_half((unsigned short)(w & 0xffffu))), inv, noi); }
RTF_DEV float plane_hi(uint32_t w, float inv, float noi) { return fmaf(__half2float(__ushort_as_half((unsigned short)(w >> 16))), inv, noi); }

// lanes with cur >= 0: both children's boxes of node cur (32 bytes: half-float boxes, rt_bvh.h), far bound = best hit
// so far; nearer child next, the other pushed
template <class ST>
RTF_DEV void walk_node(const DevScene &S, Walk &w, float t_min, ST stack)
{
    // (a 32-bit byte offset from the table's base: the load takes the base from scalar registers, the lane computes one shift;
    // trees are far below 4 GiB -- 4M primitives are 256 MiB)
    const uint4 *q = reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(S.bvh) + ((uint32_t)w.cur << 5));
    const uint4 a = q[0], b = q[1]; // {x lo|hi, y lo|hi, z lo|hi, ref} per child
    const V3 i = w.winv, n = w.noi;
    // the plane the ray meets first into the low half of each pair; then entry = max of the three near planes, exit = min of
    // the three far ones (a NaN -- a plane through the origin of a ray parallel to it -- is dropped by max3 / min3)
    const uint32_t ax = __builtin_amdgcn_alignbit(a.x, a.x, w.rx), ay = __builtin_amdgcn_alignbit(a.y, a.y, w.ry), az = __builtin_amdgcn_alignbit(a.z, a.z, w.rz);
    const uint32_t bx = __builtin_amdgcn_alignbit(b.x, b.x, w.rx), by = __builtin_amdgcn_alignbit(b.y, b.y, w.ry), bz = __builtin_amdgcn_alignbit(b.z, b.z, w.rz);
    const float tn0 = fmaxf(fmax3(plane_lo(ax, i.x, n.x), plane_lo(ay, i.y, n.y), plane_lo(az, i.z, n.z)), t_min);
    const float tf0 = fminf(fmin3(plane_hi(ax, i.x, n.x), plane_hi(ay, i.y, n.y), plane_hi(az, i.z, n.z)), w.best_t);
    const float tn1 = fmaxf(fmax3(plane_lo(bx, i.x, n.x), plane_lo(by, i.y, n.y), plane_lo(bz, i.z, n.z)), t_min);
    const float tf1 = fminf(fmin3(plane_hi(bx, i.x, n.x), plane_hi(by, i.y, n.y), plane_hi(bz, i.z, n.z)), w.best_t);
    const int ref0 = (int)a.w, ref1 = (int)b.w;
    // "First" is the box to open now: the only one that was hit, or of two the nearer.  It is to be opened if either was hit
    // and the other one pushed if both were.  Said with LANE MASKS in scalar registers (ballots of the comparisons, scalar
    // logic, and back to a condition with inverse_ballot): left to itself hipcc turns selects between booleans, and booleans
    // that live across the push, into 0 / 1 vector registers and and-s, or-s and compares them there -- a dozen vector
    // instructions per step at first, seven after the first rewrite, none now.  (A select on the distances instead, "a missed
    // box is infinitely far", costs the kernels with box leaves 30 bytes of spills per lane.)
    // A ray that starts inside both boxes enters both at t_min: the box it leaves first -- the tighter one -- goes first
    // (random spheres +2 %, 1M spheres +1 %).
    const unsigned long long m0 = __builtin_amdgcn_ballot_w64(tf0 >= tn0);
    const unsigned long long m1 = __builtin_amdgcn_ballot_w64(tf1 >= tn1) & __builtin_amdgcn_ballot_w64(ref1 != RT_BVH_NONE);
    const unsigned long long nearer0 = __builtin_amdgcn_ballot_w64(tn0 < tn1) | (__builtin_amdgcn_ballot_w64(tn0 == tn1) & __builtin_amdgcn_ballot_w64(tf0 <= tf1));
    const bool first0 = __builtin_amdgcn_inverse_ballot_w64(m0 & (~m1 | nearer0));
    const int near = first0 ? ref0 : ref1, far = first0 ? ref1 : ref0;
    const bool hit_near = __builtin_amdgcn_inverse_ballot_w64(m0 | m1), hit_far = __builtin_amdgcn_inverse_ballot_w64(m0 & m1);
    if (hit_far) walk_push(w, stack, far);
    int next = near;
    if (!hit_near) next = walk_pop(w, stack);
    w.cur = next;
}

// lanes with cur < 0 (and not NONE): leaf ~cur (rt_bvh.h: 32 bytes, what the test of a sphere or rect reads) against
// the lane's ray.  best_prim receives the leaf's rt_prim index.
// BOX: which leaves the tree may hold besides static spheres and rects IN WORLD SPACE -- 0 none, 1 moving spheres (in world
// space), 2 anything (boxes, media, leaves under an instance chain).
// Trees without them run kernels compiled without that code: fewer registers, no spills in the node loop.
#ifdef RT_TIME_PHASES // development (tools/phase_times.py): the leaf phase by branch, in the caller's counters 5..11
// (tp: sixteen counters of the WAVE in LDS, [15] the time of the last stamp; the first lane that is on writes them, so a stamp
// inside a branch counts whichever lanes took it.  The stamps themselves cost: with them the leaf phase weighs twice what it
// weighs without, so only the split between the branches is to be read from them.)
RTF_DEV void phase_stamp(unsigned long long *tp, int k)
{
    const unsigned long long now = clock64(), on = __ballot(true);
    if (__builtin_amdgcn_mbcnt_hi((uint32_t)(on >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)on, 0u)) == 0) tp[k] += now - tp[15], tp[15] = now;
}
#define RT_LEAF_CLOCK(k) do { if (tp) phase_stamp(tp, k); } while (0)
#else
#define RT_LEAF_CLOCK(k)
#endif
template <int BOX, class ST>
RTF_DEV void walk_leaf(const DevScene &S, Walk &w, const Ray &wr, float t_min, Rng &rng, ST stack, unsigned long long *tp = nullptr)
{
    RT_LEAF_CLOCK(2);
    const uint4 *q = reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(S.leaves) + ((uint32_t)~w.cur << 5));
    const uint4 la = q[0], lb = q[1]; // f0 f1 f2 f3 | f4 kind_chain prim visits
    const int kind = (int)(lb.y & 255u), chain = (int)(lb.y >> 8);
    // d.d as src/sphere.h:27 rounds it (it enters the sign of the discriminant, sphere_disc) and its reciprocal: made here, per
    // leaf test, rather than carried through the walk in two registers (chains are rigid: the same in every space)
    const float w_dd = ref_dot(wr.d, wr.d), w_inv_dd = rcp(w_dd);
    Ray lr = wr;
    V3 inv = w.winv;
    if (BOX == 2 && chain != 0) { // (BOX < 2: every leaf is in world space, and the ray is not copied: eight moves per leaf test)
        lr = rt::chain_down(S, chain, wr);
        inv = rcp3(lr.d);
    }
    float t = 0.0f;
    bool h;
    RT_LEAF_CLOCK(5); // fetch, decode, chain
    if (kind == RT_PRIM_SPHERE) { // src/sphere.h:21-59
        const V3 oc = v3(lr.o.x - __uint_as_float(la.x), lr.o.y - __uint_as_float(la.y), lr.o.z - __uint_as_float(la.z));
        const float radius = __uint_as_float(la.w);
        float b, disc;
        sphere_disc(oc, lr.d, w_dd, radius, b, disc);
        const float sq = fsqrt(disc);
        const float t1 = (-b - sq) * w_inv_dd, t2 = (-b + sq) * w_inv_dd;
        const bool in1 = (t1 < w.best_t) & (t1 > t_min), in2 = (t2 < w.best_t) & (t2 > t_min);
        t = in1 ? t1 : t2;
        h = (disc > 0.0f) & (in1 | in2);
        RT_LEAF_CLOCK(6);
    } else if (kind >= RT_PRIM_XY_RECT && kind <= RT_PRIM_YZ_RECT) { // src/plane.h:9-29, 47-67, 109-129
        const bool xy = kind == RT_PRIM_XY_RECT, xz = kind == RT_PRIM_XZ_RECT;
        const float ok = xy ? lr.o.z : xz ? lr.o.y : lr.o.x, ik = xy ? inv.z : xz ? inv.y : inv.x;
        const float oa = (xy | xz) ? lr.o.x : lr.o.y, da = (xy | xz) ? lr.d.x : lr.d.y;
        const float ob = xy ? lr.o.y : lr.o.z, db = xy ? lr.d.y : lr.d.z;
        t = (__uint_as_float(lb.x) - ok) * ik;
        const float a = fmaf(t, da, oa), b = fmaf(t, db, ob);
        h = (t >= t_min) & (t <= w.best_t) & (a >= __uint_as_float(la.x)) & (a <= __uint_as_float(la.y)) & (b >= __uint_as_float(la.z)) &
            (b <= __uint_as_float(la.w));
        RT_LEAF_CLOCK(7);
    } else if (BOX == 0) { // (a tree of static spheres and rects only: the kernel is compiled without the rest)
        h = false;
    } else if (kind == RT_LEAF_MOVING_SPHERE) { // src/sphere.h:142-183, from the leaf's two slots (rt_bvh.h)
        const uint4 le = q[2]; // c1 - c0
        const float k = (lr.time - __uint_as_float(lb.x)) * __uint_as_float(lb.w); // center(time), src/sphere.h:142-145
        const V3 c = v3(fmaf(k, __uint_as_float(le.x), __uint_as_float(la.x)), fmaf(k, __uint_as_float(le.y), __uint_as_float(la.y)),
                        fmaf(k, __uint_as_float(le.z), __uint_as_float(la.z)));
        const V3 oc = v3(lr.o.x - c.x, lr.o.y - c.y, lr.o.z - c.z);
        float b, disc;
        sphere_disc(oc, lr.d, w_dd, __uint_as_float(la.w), b, disc);
        const float sq = fsqrt(disc);
        const float t1 = (-b - sq) * w_inv_dd, t2 = (-b + sq) * w_inv_dd;
        const bool in1 = (t1 < w.best_t) & (t1 > t_min), in2 = (t2 < w.best_t) & (t2 > t_min);
        t = in1 ? t1 : t2;
        h = (disc > 0.0f) & (in1 | in2);
        RT_LEAF_CLOCK(8);
    } else if (BOX < 2) { // (no boxes or media in the tree: the kernel is compiled without them)
        h = false;
    } else if (kind == RT_PRIM_GROUP) { // a box: one slab test for its six faces
        float tn, tf;
        const bool cut = box_range(v3(__uint_as_float(la.x), __uint_as_float(la.y), __uint_as_float(la.z)),
                                   v3(__uint_as_float(la.w), __uint_as_float(lb.x), __uint_as_float(lb.w)), lr, inv, tn, tf);
        t = tn >= t_min ? tn : tf;
        h = cut & (t >= t_min) & (t <= w.best_t);
        RT_LEAF_CLOCK(9);
    } else { // constant_medium::hit, src/volume.h:23-53 (with its line-41 assignment); or a primitive the leaf only names
        // the boundary's first hit on the whole line, then the first one after it
        float t1 = 0.0f, t2 = 0.0f, density = 1.0f;
        int visits = 1;
        bool both = false, medium = true;
        if (kind == RT_LEAF_MEDIUM_SPHERE) { // sphere::hit twice: (-inf, inf) gives the smaller root, (t1 + 0.0001, inf) the larger
            const V3 oc = v3(lr.o.x - __uint_as_float(la.x), lr.o.y - __uint_as_float(la.y), lr.o.z - __uint_as_float(la.z));
            float b, disc;
            sphere_disc(oc, lr.d, w_dd, __uint_as_float(la.w), b, disc);
            const float sq = fsqrt(disc);
            t1 = (-b - sq) * w_inv_dd, t2 = (-b + sq) * w_inv_dd;
            both = (disc > 0.0f) & (t1 > -RT_FLT_MAX) & (t2 < RT_FLT_MAX) & (t2 > t1 + 0.0001f);
            density = __uint_as_float(lb.x), visits = (int)lb.w;
        } else if (kind == RT_LEAF_MEDIUM_BOX) {
            const uint4 le = q[2], lf = q[3]; // density | ... visits
            both = box_range(v3(__uint_as_float(la.x), __uint_as_float(la.y), __uint_as_float(la.z)),
                             v3(__uint_as_float(la.w), __uint_as_float(lb.x), __uint_as_float(lb.w)), lr, inv, t1, t2) & (t2 > t1 + 0.0001f);
            density = __uint_as_float(le.x), visits = (int)lf.w;
        } else {
            const rt_prim &pr = S.fprims[(int)lb.z];
            medium = kind == RT_PRIM_MEDIUM;
            if (medium) { // a boundary of another kind: its own hit function, twice
                const rt_prim &bp = S.fprims[pr.first];
                int dummy;
                both = simple_t(S, pr.first, bp, lr, inv, w_dd, w_inv_dd, -RT_FLT_MAX, RT_FLT_MAX, t1, dummy) &&
                       simple_t(S, pr.first, bp, lr, inv, w_dd, w_inv_dd, t1 + 0.0001f, RT_FLT_MAX, t2, dummy);
                density = pr.f[0], visits = (int)lb.w;
            } else {
                h = plain_t(pr, lr, inv, w_dd, w_inv_dd, t_min, w.best_t, t);
            }
        }
        if (medium) {
            h = false;
            if (both) {
                t1 = fminf(fmaxf(t1, t_min), w.best_t);
                if (t1 < t2) {
                    t1 = fmaxf(t1, 0.0f);
                    // one draw per visit the reference pays this medium (rt_bvh.cpp: a lone child of a bvh_node is visited
                    // twice); the nearest of the candidates is what its tree keeps
                    float u = rng.next();
                    for (int v = 1; v < visits; ++v) u = fmaxf(u, rng.next());
                    const float hit_distance = -rcp(density) * __logf(u);
                    t = fmaf(hit_distance, rsq(w_dd), t1);
                    h = true;
                }
            }
        }
        RT_LEAF_CLOCK(10);
    }
    RT_LEAF_CLOCK(11); // the join
    const bool take = h & (t <= w.best_t);
    w.best_t = take ? t : w.best_t;
    w.best_prim = take ? (int)lb.z : w.best_prim;
    w.cur = walk_pop(w, stack);
}

// one ray per lane, to the end
RTF_DEV bool hit_bvh(const DevScene &S, const Ray &wr, float t_min, float t_max, bool alive, Rng &rng, float &best_t, int &best_prim,
                     int *stack)
{
    Walk w;
    walk_start(w, wr, t_max, alive);
    while (__any(w.cur != RT_BVH_NONE)) {
        while (w.cur >= 0) walk_node(S, w, t_min, stack);
        if (w.cur != RT_BVH_NONE) walk_leaf<2>(S, w, wr, t_min, rng, stack);
    }
    best_t = w.best_t, best_prim = w.best_prim;
    return best_prim >= 0;
}

// point, normal, (u,v), material of the winner; every lane for itself.  BOX as walk_leaf's: the leaf kinds the tree may hold.
// The primitive's record and its shade record (rt_bvh.h) are fetched in ONE round: read field by field, each behind a branch on
// the one before (kind, then chain, then f, then flags, then the shade record), a shaded ray waited for memory four or five times
// in a row.
template <int BOX>
RTF_DEV void build_surface_lanes(const DevScene &S, const Ray &wr, float t, int prim, Surface &s)
{
    prim = prim < 0 ? 0 : prim & RT_LEAF_PRIM_MASK;
    const uint4 *pq = reinterpret_cast<const uint4 *>(S.fprims + prim);
    const uint4 h0 = pq[0], g0 = pq[2], g1 = pq[3]; // kind flags material chain | f[0..3] | f[4..7]
    const uint4 *sq = reinterpret_cast<const uint4 *>(S.fshade + prim);
    uint4 sa = sq[0], sb = sq[1];
    const int kind = (int)h0.x, chain = (int)h0.w;
    const uint32_t flags = h0.y;
    const float f0 = __uint_as_float(g0.x), f1 = __uint_as_float(g0.y), f2 = __uint_as_float(g0.z), f3 = __uint_as_float(g0.w);
    const float f4 = __uint_as_float(g1.x), f5 = __uint_as_float(g1.y), f6 = __uint_as_float(g1.z), f7 = __uint_as_float(g1.w);
    s.p = fma3(t, wr.d, wr.o);
    s.u = s.v = 0.0f;
    s.face = prim;
    V3 n = v3(1, 0, 0); // constant_medium: src/volume.h:47
    if (BOX == 2 && kind == RT_PRIM_GROUP) { // a box: the face whose plane holds the point, its outward normal, its rect's material
        const int first = (int)pq[1].x;
        const Ray lr = chain == 0 ? wr : rt::chain_down(S, chain, wr);
        const V3 pl = fma3(t, lr.d, lr.o);
        const V3 lo = v3(f0, f1, f2), hi = v3(f3, f4, f5);
        const V3 q = v3(pl.x - 0.5f * (lo.x + hi.x), pl.y - 0.5f * (lo.y + hi.y), pl.z - 0.5f * (lo.z + hi.z));
        const float ex = fabsf(q.x) - 0.5f * (hi.x - lo.x), ey = fabsf(q.y) - 0.5f * (hi.y - lo.y), ez = fabsf(q.z) - 0.5f * (hi.z - lo.z);
        const bool is_x = (ex >= ey) & (ex >= ez), is_y = !is_x & (ey >= ez);
        const int axis = is_x ? 0 : is_y ? 1 : 2;
        const float qa = is_x ? q.x : is_y ? q.y : q.z;
        const float sg = qa < 0.0f ? -1.0f : 1.0f;
        n = v3(is_x ? sg : 0.0f, is_y ? sg : 0.0f, (is_x | is_y) ? 0.0f : sg);
        if (chain != 0) {
            V3 dummy = pl;
            rt::chain_up(S, chain, dummy, n);
        }
        s.face = first + 2 * axis + (qa < 0.0f ? 0 : 1);
        const uint4 *fq = reinterpret_cast<const uint4 *>(S.fshade + s.face); // the face rect's material: the one fetch that waits for another
        sa = fq[0], sb = fq[1];
    } else if (kind != RT_PRIM_MEDIUM) {
        const bool chained = BOX == 2 && chain != 0; // (BOX < 2: every leaf is in world space)
        const Ray lr = !chained ? wr : rt::chain_down(S, chain, wr);
        const V3 pl = fma3(t, lr.d, lr.o);
        if (kind <= RT_PRIM_MOVING_SPHERE) {
            V3 c = v3(f0, f1, f2);
            float radius = f3;
            if (BOX >= 1 && kind == RT_PRIM_MOVING_SPHERE) { // f = c0 c1 r t0 t1 (include/rt_amd.h)
                const float f8 = __uint_as_float(pq[4].x);
                const float k = (lr.time - f7) * rcp(f8 - f7);
                c = v3(fmaf(k, f3 - f0, f0), fmaf(k, f4 - f1, f1), fmaf(k, f5 - f2, f2));
                radius = f6;
            }
            const float ir = rcp(radius);
            n = v3((pl.x - c.x) * ir, (pl.y - c.y) * ir, (pl.z - c.z) * ir);
            if (flags & RT_PRIM_NEEDS_UV) rt::sphere_uv(n, s.u, s.v); // before the flip, as sphere::hit does
        } else {
            const bool xy = kind == RT_PRIM_XY_RECT, xz = kind == RT_PRIM_XZ_RECT;
            n = v3((xy | xz) ? 0.0f : 1.0f, xz ? 1.0f : 0.0f, xy ? 1.0f : 0.0f);
            if (flags & RT_PRIM_NEEDS_UV) {
                const float a = (xy | xz) ? pl.x : pl.y, b = xy ? pl.y : pl.z;
                s.u = (a - f0) * rcp(f1 - f0);
                s.v = (b - f2) * rcp(f3 - f2);
            }
        }
        if (flags & RT_PRIM_FLIP) n = v3(-n.x, -n.y, -n.z);
        if (chained) {
            V3 dummy = pl;
            rt::chain_up(S, chain, dummy, n);
        }
    }
    s.n = n;
    // material and texture, resolved on the host into one record per primitive (rt_bvh.h)
    s.mat_kind = sa.x;
    s.texture = (int32_t)sa.y;
    s.p0 = __uint_as_float(sa.z);
    s.color = v3(__uint_as_float(sa.w), __uint_as_float(sb.x), __uint_as_float(sb.y));
}

} // namespace rtf
#endif // RT_FASTLANES_HPP_
