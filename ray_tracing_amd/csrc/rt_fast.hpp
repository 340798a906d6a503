// rt_fast.hpp -- the render path for scenes that fit the record table (rt_records.h): same algorithm and same
// random-draw order as rt_device.hpp, different machine mapping.
//
//   * The scene is a table of at most 64 records of 32 bytes.  The scan walks it with a wave-uniform index, so
//     scene data is the same for all 64 lanes and no lane computes an address: small scenes (stored flat and in
//     runs, rt_records.cpp) read their records from a copy in LDS with broadcast reads and test a rect with one
//     block of v_cmpx instructions (rect_update_v); larger ones use scalar loads with the next record in flight.
//   * Lane L of every wave also holds record L in 8 VGPRs.  During the scan a lane keeps only (t, record index)
//     of its closest hit; the hit point, normal and material are built once per ray afterwards: the lane fetches
//     the winning record and its material record with ds_bpermute.
//   * Spheres are stored in world space (a rigid chain is folded into the centre), rect records carry their world
//     normal, p = o + t*d is evaluated on the world ray: instance chains cost one ray transform per chain.
//   * Arithmetic is hardware-rate: v_rcp/v_sqrt/v_rsq/v_sin/v_cos/v_log (about 1 ulp) and FMA contraction, 1/d
//     hoisted per ray.  Results therefore differ from the reference in the last bits; the bit-exact path
//     (rt_device.hpp) remains the one that rt_trace_rays uses by default, and tests/test_gpu_parity.py compares the
//     two paths with each other, with the oracle on the same random stream, and with the reference's converged image.
//   * Comparisons are written in their positive form (t >= t_min && t <= t_max ...) so that a NaN never counts as
//     a hit; the reference's negative forms accept a NaN t, which only arises for a ray origin exactly on a
//     primitive's plane with a direction exactly parallel to it.
#ifndef RT_FAST_HPP_
#define RT_FAST_HPP_

#include "rt_device.hpp"
#include "rt_records.h"

namespace rtf {

using rt::DevScene;
using rt::Ray;
using rt::Rng;
using rt::V3;
using rt::v3;

#define RTF_DEV __device__ __forceinline__
#pragma clang fp contract(fast)

RTF_DEV float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
RTF_DEV float rsq(float x) { return __builtin_amdgcn_rsqf(x); }
RTF_DEV float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
RTF_DEV float fmin3(float a, float b, float c) { return fminf(fminf(a, b), c); }
RTF_DEV float fmax3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
RTF_DEV V3 fma3(float t, V3 d, V3 o) { return v3(fmaf(t, d.x, o.x), fmaf(t, d.y, o.y), fmaf(t, d.z, o.z)); }
RTF_DEV float fdot(V3 a, V3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
RTF_DEV V3 funit(V3 a)
{
    float k = rsq(fdot(a, a));
    return v3(a.x * k, a.y * k, a.z * k);
}
RTF_DEV V3 fcross(V3 a, V3 b)
{
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// sin and cos of 2*RT_PI*r (RT_PI = 3.14159f, src/main.cpp:54): the hardware takes revolutions
RTF_DEV void sincos_2pi(float r, float &s, float &c)
{
    const float rev = r * (RT_PI / 3.14159265358979323846f);
    s = __builtin_amdgcn_sinf(rev);
    c = __builtin_amdgcn_cosf(rev);
}

struct Regs { // the record table, one record per lane
    uint32_t d[8];
};
RTF_DEV float rl_f(uint32_t v, int lane) { return __uint_as_float(__builtin_amdgcn_readlane(v, lane)); }
RTF_DEV uint32_t rl_u(uint32_t v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
RTF_DEV uint32_t sh_u(uint32_t v, int lane) { return (uint32_t)__shfl((int)v, lane, 64); }
RTF_DEV float sh_f(uint32_t v, int lane) { return __uint_as_float(sh_u(v, lane)); }

struct Layout { // wave-uniform description of the table
    int32_t n_stream, materials_base, lights_base, n_lights;
    uint32_t run[RT_REC_MAX_RUNS]; // rt_records.h: the runs of a flat stream
    int32_t n_runs, n_spheres;
    uint32_t rect_lights; // bit k: member k of the important list is a rect (the first 32 members)
    float inv_n_lights;   // 1 / n_lights (hitable_list::pdf_value's weight, src/hitable_list.h:69), from the host
};

// Arithmetic that decides WHICH primitive is hit (north_star check 1: ids equal to the reference's hitable::hit).
// The plane distance t = (k - o) / d of a rect (src/plane.h:11,49,111) is the reference's correctly rounded quotient,
// at the price of two FMAs instead of an IEEE division: with y = 1/d refined once per ray (Newton step on v_rcp_f32),
// q = n*y, r = fma(-q, d, n) is the exact remainder and fma(r, y, q) rounds to the quotient (Markstein); the in-plane
// point is o + t*d with the product and the sum rounded one after the other, as g++ compiles it (no contraction).
// -DRT_FAST_APPROX_T gives the round-1 arithmetic back (v_rcp and FMA; t within an ulp or two): a development switch
// that prices the difference (tools/build_variant.sh).
RTF_DEV float mul_rn(float a, float b) { return __fmul_rn(a, b); } // never contracted into an FMA
RTF_DEV float add_rn(float a, float b) { return __fadd_rn(a, b); }
RTF_DEV float sub_rn(float a, float b) { return __fsub_rn(a, b); }
RTF_DEV float refined_rcp(float d)
{
    const float y = rcp(d);
#ifdef RT_FAST_APPROX_T
    return y;
#else
    return fmaf(fmaf(-d, y, 1.0f), y, y); // d = 0: NaN, and a NaN t is never a hit
#endif
}
// n / d, correctly rounded, from y = refined_rcp(d)
RTF_DEV float div_rn(float n, float d, float y)
{
    const float q = n * y;
#ifdef RT_FAST_APPROX_T
    return q;
#else
    return fmaf(fmaf(-q, d, n), y, q);
#endif
}

// the ray in the space of the current chain, with what the slab and rect tests need
struct LocalRay {
    V3 o, d, inv;
};
RTF_DEV void set_local(LocalRay &l, V3 o, V3 d)
{
    l.o = o, l.d = d;
    l.inv = v3(refined_rcp(d.x), refined_rcp(d.y), refined_rcp(d.z));
}

// One stream record in SGPRs.  The scan reads the record table with wave-uniform scalar loads
// (s_load_dwordx8 from the 2 KiB table, which stays in the scalar cache) and always has the NEXT
// record's load in flight while it works on the current one, so the loads cost the VALU nothing.
// (v_readlane from the register-resident copy would cost a VALU slot per dword: six per record.)
struct SRec {
    uint32_t d[8];
};
RTF_DEV SRec load_rec(const uint32_t *__restrict__ recs, int i)
{
    const uint4 a = reinterpret_cast<const uint4 *>(recs)[2 * i], b = reinterpret_cast<const uint4 *>(recs)[2 * i + 1];
    SRec r;
    r.d[0] = a.x, r.d[1] = a.y, r.d[2] = a.z, r.d[3] = a.w, r.d[4] = b.x, r.d[5] = b.y, r.d[6] = b.z, r.d[7] = b.w;
    return r;
}
RTF_DEV float sf(const SRec &r, int k) { return __uint_as_float(r.d[k]); }

// translate::hit / rotate_y::hit on the way down (src/transforms.h:17, 107-110), operations from records
// (translate and rotate_y leave a direction's y alone, so 1/d.y is the world ray's: one reciprocal and its refinement less)
RTF_DEV void apply_chain(const uint32_t *__restrict__ recs, uint32_t chain_bits, const Ray &wr, float world_inv_y, LocalRay &l)
{
    V3 o = wr.o, d = wr.d;
    const int first = chain_bits & 63, count = chain_bits >> 6;
    for (int k = 0; k < count; ++k) {
        const SRec x = load_rec(recs, first + k);
        const float a = sf(x, 0), b = sf(x, 1);
        if ((x.d[7] & 15u) == RT_REC_XFORM_T) {
            o = v3(o.x - a, o.y - b, o.z - sf(x, 2));
        } else { // a = sin, b = cos; products and sums rounded one by one, as the reference's (src/transforms.h:107-110)
            const float ox = sub_rn(mul_rn(b, o.x), mul_rn(a, o.z)), oz = add_rn(mul_rn(a, o.x), mul_rn(b, o.z));
            const float dx = sub_rn(mul_rn(b, d.x), mul_rn(a, d.z)), dz = add_rn(mul_rn(a, d.x), mul_rn(b, d.z));
            o.x = ox, o.z = oz, d.x = dx, d.z = dz;
        }
    }
    l.o = o, l.d = d;
    l.inv = v3(refined_rcp(d.x), world_inv_y, refined_rcp(d.z));
}

// ---------------------------------------------------------------------------
// world->hit(): bvh_node::hit (src/bvh.h:123-163) over the record stream.
// Returns the closest t and the stream index of the record that was hit (-1: miss).
// Every primitive record competes with the best so far and replaces it unless the best is
// strictly nearer (src/bvh.h:132-139).  Members of lists (box faces) are stored as leaves of
// their own: for rects that is hitable_list::hit (rt_records.cpp explains why).
// best_t starts at t_max, so `t <= best_t` is also the upper range test of the primitive.
// FLAT: the stream holds no box records (small scenes), so no lane ever skips.
// ---------------------------------------------------------------------------
// One rect record against the ray of its chain.  Straight-line code per axis (the axis is
// wave-uniform); bitwise & on purpose: && would make hipcc branch around every comparison.
RTF_DEV bool rect_in(const SRec &c, float ok, float dk, float ik, float oa, float da, float ob, float db, float t_min, float bound,
                     float &t)
{
    t = div_rn(sf(c, 4) - ok, dk, ik);
#ifdef RT_FAST_APPROX_T
    const float a = fmaf(t, da, oa), b = fmaf(t, db, ob);
#else
    const float a = add_rn(oa, mul_rn(t, da)), b = add_rn(ob, mul_rn(t, db));
#endif
    return (t >= t_min) & (t <= bound) & (a >= sf(c, 0)) & (a <= sf(c, 1)) & (b >= sf(c, 2)) & (b <= sf(c, 3));
}

// The lane's best hit so far is ONE 64-bit number (rt_records.h, RT_REC_KEY): t's bits above the record's key.  A hit
// replaces it iff its number is smaller -- nearer, or exactly as near and met later by the reference's depth-first walk
// (src/bvh.h:132-139: the right child wins a tie) -- so the order of the stream decides nothing.  t > 0 for every
// candidate (t >= t_min is tested first), so the bits of t order like t.
typedef unsigned long long Best;
RTF_DEV Best best_start(float t_max) { return ((Best)__float_as_uint(t_max) << 32) | RT_REC_KEY_NONE; }
RTF_DEV Best best_of(float t, uint32_t key) { return ((Best)__float_as_uint(t) << 32) | key; }
RTF_DEV float best_t_of(Best b) { return __uint_as_float((uint32_t)(b >> 32)); }
RTF_DEV int best_rec_of(Best b) { return (uint32_t)b == RT_REC_KEY_NONE ? -1 : (int)RT_KEY_INDEX((uint32_t)b); }
RTF_DEV uint32_t best_key_of(Best b) { return (uint32_t)b; } // RT_REC_KEY of the record that was hit, RT_REC_KEY_NONE on a miss

// One rect record against the ray of its chain, and the update of the lane's best hit.
// What the compiler makes of rect_in() is 6 v_cmp + 5 s_and_b64 + 2 v_cndmask; the SIMD issues one scalar instruction
// per turn just like one vector instruction, and with the loop's own scalar work the scan was bound by the scalar pipe
// (35 scalar against 13 vector instructions per record).  Here the comparisons narrow EXEC one after the other
// (v_cmpx), one 64-bit move runs under what is left, and EXEC is restored: 8 vector instructions for t and the in-plane
// point (intrinsics: nothing is contracted or reordered), 7 for the decision, 2 scalar.
// Comparisons in positive form, so a NaN is never a hit.
RTF_DEV void rect_update_v(float k, uint32_t key, float a0, float a1, float b0, float b1, float t_min, float ok, float dk, float ik,
                           float oa, float da, float ob, float db, Best &best)
{
    const float t = div_rn(sub_rn(k, ok), dk, ik);
#ifdef RT_FAST_APPROX_T
    const float a = fmaf(t, da, oa), b = fmaf(t, db, ob);
#else
    const float a = add_rn(oa, mul_rn(t, da)), b = add_rn(ob, mul_rn(t, db));
#endif
    const Best cand = best_of(t, key);
    unsigned long long saved;
    asm volatile("s_mov_b64 %[sv], exec\n\t"
                 "v_cmpx_le_f32 %[tmin], %[t]\n\t"
                 "v_cmpx_le_f32 %[a0], %[a]\n\t"
                 "v_cmpx_ge_f32 %[a1], %[a]\n\t"
                 "v_cmpx_le_f32 %[b0], %[b]\n\t"
                 "v_cmpx_ge_f32 %[b1], %[b]\n\t"
                 "v_cmpx_gt_u64 %[best], %[cand]\n\t"
                 "v_mov_b64 %[best], %[cand]\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [sv] "=&s"(saved), [best] "+v"(best)
                 : [t] "v"(t), [a] "v"(a), [b] "v"(b), [cand] "v"(cand), [a0] "v"(a0), [a1] "v"(a1), [b0] "v"(b0), [b1] "v"(b1),
                   [tmin] "s"(t_min)
                 : "vcc");
}

// `count` rect records of one chain and axis from record `index` on: nothing to decode, the loop is fetch + test.
// The records come from the block's copy of the table in LDS with broadcast reads (every lane the same address): the
// LDS pipe is idle otherwise, the scalar pipe was the busy one, and the next record's read is in flight during the test.
// In that copy (stage_records) a stream record carries its key and, beside what the scan reads, what shading reads:
//   rect    d0-3 a0 a1 b0 b1 | d4 key, d5 the plane | d6 d7 world normal x, z
//   sphere  d0-2 centre, d3 radius | d4 key, d5 1 / radius
//   box     d0-2 p0, d3 key | d4-6 p1          (its second record, BOX_AUX, as in memory)
struct LRec {
    uint4 a;   // a0 a1 b0 b1 (rect) | cx cy cz r (sphere)
    uint2 kk;  // key; rect: the plane
};
RTF_DEV LRec lds_rec(const uint32_t *lrecs, int i)
{
    LRec r;
    r.a = *reinterpret_cast<const uint4 *>(lrecs + 8 * i);
    r.kk = *reinterpret_cast<const uint2 *>(lrecs + 8 * i + 4);
    return r;
}
// the block's copy of the record table (all RT_REC_MAX + 1 records), stream records re-arranged as above;
// every thread of the block calls it, the caller's barrier follows
RTF_DEV void stage_records(const uint32_t *__restrict__ recs, uint32_t *lrecs, int n_stream, int tid, int n_threads)
{
    for (int k = tid; k < (RT_REC_MAX + 1) * 8; k += n_threads) {
        const int i = k >> 3, j = k & 7;
        uint32_t v = recs[k];
        if (i < n_stream) {
            const uint32_t kw = recs[8 * i + 7], kind = kw & 15u;
            const uint32_t key = RT_REC_KEY(kw, i);
            if (kind == RT_REC_RECT) v = j == 4 ? key : j >= 5 ? recs[8 * i + j - 1] : v;
            if (kind == RT_REC_SPHERE) v = j == 4 ? key : j == 5 ? recs[8 * i + 4] : v;
            if (kind == RT_REC_BOX) v = j == 3 ? key : v;
        }
        lrecs[k] = v;
    }
}
RTF_DEV void rect_run(const uint32_t *lrecs, int &index, int count, float t_min, float ok, float dk, float ik, float oa, float da,
                      float ob, float db, Best &best)
{
    // Two records per turn, each read while the other is tested (the table is padded: reading one record past the run is
    // harmless).  Written out by hand: a rotating "next record" variable costs four register moves per record.
    if (count == 0) return;
    LRec a = lds_rec(lrecs, index);
    int k = 0;
    for (; k + 2 <= count; k += 2, index += 2) {
        const LRec b = lds_rec(lrecs, index + 1);
        rect_update_v(__uint_as_float(a.kk.y), a.kk.x, __uint_as_float(a.a.x), __uint_as_float(a.a.y), __uint_as_float(a.a.z),
                      __uint_as_float(a.a.w), t_min, ok, dk, ik, oa, da, ob, db, best);
        a = lds_rec(lrecs, index + 2);
        rect_update_v(__uint_as_float(b.kk.y), b.kk.x, __uint_as_float(b.a.x), __uint_as_float(b.a.y), __uint_as_float(b.a.z),
                      __uint_as_float(b.a.w), t_min, ok, dk, ik, oa, da, ob, db, best);
    }
    if (k < count) {
        rect_update_v(__uint_as_float(a.kk.y), a.kk.x, __uint_as_float(a.a.x), __uint_as_float(a.a.y), __uint_as_float(a.a.z),
                      __uint_as_float(a.a.w), t_min, ok, dk, ik, oa, da, ob, db, best);
        ++index;
    }
}

// One box record (src/box.h: a list of six rects with outward normals) against the ray of its chain, and the update of
// the lane's best hit.  The nearest face hit inside (t_min, best] is the entry into the intersection of the three slabs,
// or the exit when the entry lies before t_min (the origin is inside, or on the face it leaves): one slab test, 27
// vector instructions, instead of six rect tests (72).  Which face it was is worked out after the scan (box_normal).
RTF_DEV float slab_div(float n, float d, float y)
{
#ifdef RT_FAST_APPROX_T
    return n * y;
#else
    return __builtin_amdgcn_div_fixupf(div_rn(n, d, y), d, n);
#endif
}
RTF_DEV void slab_range(V3 lo, V3 hi, const LocalRay &l, float &tn, float &tf)
{
    // each plane's distance is the t the reference's face rect computes, so the t of a box hit is its face's t.  A
    // direction component of exactly 0 makes refined_rcp (and so div_rn) a NaN; v_div_fixup puts the IEEE special
    // cases back (n/0 = +-inf: the slab on that axis is everything or nothing), which min/max need here.
    const float x0 = slab_div(lo.x - l.o.x, l.d.x, l.inv.x), x1 = slab_div(hi.x - l.o.x, l.d.x, l.inv.x);
    const float y0 = slab_div(lo.y - l.o.y, l.d.y, l.inv.y), y1 = slab_div(hi.y - l.o.y, l.d.y, l.inv.y);
    const float z0 = slab_div(lo.z - l.o.z, l.d.z, l.inv.z), z1 = slab_div(hi.z - l.o.z, l.d.z, l.inv.z);
    tn = fmax3(fminf(x0, x1), fminf(y0, y1), fminf(z0, z1));
    tf = fmin3(fmaxf(x0, x1), fmaxf(y0, y1), fmaxf(z0, z1));
}
RTF_DEV void box_update_v(V3 lo, V3 hi, uint32_t key, float t_min, const LocalRay &l, Best &best)
{
    float tn, tf;
    slab_range(lo, hi, l, tn, tf);
    const float t = t_min <= tn ? tn : tf;
    const Best cand = best_of(t, key);
    unsigned long long saved;
    asm volatile("s_mov_b64 %[sv], exec\n\t"
                 "v_cmpx_le_f32 %[tn], %[tf]\n\t"
                 "v_cmpx_le_f32 %[tmin], %[t]\n\t"
                 "v_cmpx_gt_u64 %[best], %[cand]\n\t"
                 "v_mov_b64 %[best], %[cand]\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [sv] "=&s"(saved), [best] "+v"(best)
                 : [tn] "v"(tn), [tf] "v"(tf), [t] "v"(t), [cand] "v"(cand), [tmin] "s"(t_min)
                 : "vcc");
}
RTF_DEV void box_run(const uint32_t *lrecs, int &index, int count, float t_min, const LocalRay &l, Best &best)
{
    for (int k = 0; k < count; ++k, index += 2) { // the record after a box is its second half
        const uint4 a = *reinterpret_cast<const uint4 *>(lrecs + 8 * index), b = *reinterpret_cast<const uint4 *>(lrecs + 8 * index + 4);
        box_update_v(v3(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z)),
                     v3(__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z)), a.w, t_min, l, best);
    }
}

// sphere::hit's discriminant and roots (src/sphere.h:26-47) with the reference's operations one by one: dot products as
// ((x*x + y*y) + z*z), IEEE square root, quotients correctly rounded (div_rn with 1/a refined once per ray).
// dd = ref_dot(d, d), inv_dd = refined_rcp(dd).
RTF_DEV float ref_dot(V3 a, V3 b) { return add_rn(add_rn(mul_rn(a.x, b.x), mul_rn(a.y, b.y)), mul_rn(a.z, b.z)); }
RTF_DEV void sphere_roots(V3 o, V3 d, float dd, float inv_dd, float cx, float cy, float cz, float radius, float &disc, float &t1,
                          float &t2)
{
    const V3 oc = v3(o.x - cx, o.y - cy, o.z - cz);
#ifdef RT_FAST_APPROX_T
    const float b = fdot(oc, d);
    const float cc = fmaf(-radius, radius, fdot(oc, oc));
    disc = fmaf(b, b, -(dd * cc));
    const float sq = fsqrt(disc);
    t1 = (-b - sq) * inv_dd, t2 = (sq - b) * inv_dd;
#else
    const float b = ref_dot(oc, d);
    const float cc = sub_rn(ref_dot(oc, oc), mul_rn(radius, radius));
    disc = sub_rn(mul_rn(b, b), mul_rn(dd, cc));
    const float sq = sqrtf(disc); // correctly rounded (hipcc's default for sqrtf)
    t1 = div_rn(sub_rn(-b, sq), dd, inv_dd), t2 = div_rn(add_rn(-b, sq), dd, inv_dd);
#endif
}

// The scan of a flat stream stored in runs (rt_records.h): per chain three loops of rects (one per axis) and one of
// boxes, then the spheres; run lengths come from the kernel arguments, so a record costs its fetch, its test and the loop's counter.
// `lrecs`: the record stream in LDS; `recs`: the table in memory (instance chains, read with scalar loads).
RTF_DEV Best scan_runs(const uint32_t *__restrict__ recs, const uint32_t *lrecs, const Layout &Y, const Ray &wr, float t_min, float t_max)
{
    Best best = best_start(t_max);
    LocalRay l;
    set_local(l, wr.o, wr.d);
    const float world_inv_y = l.inv.y;
    int index = 0;
    for (int r = 0; r < Y.n_runs; ++r) {
        const uint32_t w = Y.run[r];
        if ((w >> 15) & RT_KW_CHAIN_MASK) apply_chain(recs, (w >> 15) & RT_KW_CHAIN_MASK, wr, world_inv_y, l);
        rect_run(lrecs, index, (int)((w >> 10) & 31u), t_min, l.o.z, l.d.z, l.inv.z, l.o.x, l.d.x, l.o.y, l.d.y, best);
        rect_run(lrecs, index, (int)((w >> 5) & 31u), t_min, l.o.y, l.d.y, l.inv.y, l.o.x, l.d.x, l.o.z, l.d.z, best);
        rect_run(lrecs, index, (int)(w & 31u), t_min, l.o.x, l.d.x, l.inv.x, l.o.y, l.d.y, l.o.z, l.d.z, best);
        box_run(lrecs, index, (int)(w >> 24), t_min, l, best);
    }
    // spheres, world space: src/sphere.h:21-59 (strict range test: t_min < t < t_max).  The smaller root is the hit if it
    // lies behind t_min, else only the larger one can be; a negative discriminant makes every comparison false.
#ifdef RT_FAST_APPROX_T
    const float w_dd = fdot(wr.d, wr.d), w_inv_dd = rcp(w_dd);
#else
    const float w_dd = ref_dot(wr.d, wr.d), w_inv_dd = refined_rcp(w_dd);
#endif
    for (int k = 0; k < Y.n_spheres; ++k, ++index) {
        const LRec c = lds_rec(lrecs, index);
        float disc, t1, t2;
        sphere_roots(wr.o, wr.d, w_dd, w_inv_dd, __uint_as_float(c.a.x), __uint_as_float(c.a.y), __uint_as_float(c.a.z), __uint_as_float(c.a.w),
                     disc, t1, t2);
        const float t = t_min < t1 ? t1 : t2;
        const Best cand = best_of(t, c.kk.x);
        unsigned long long saved;
        asm volatile("s_mov_b64 %[sv], exec\n\t"
                     "v_cmpx_lt_f32 0, %[disc]\n\t"
                     "v_cmpx_lt_f32 %[tmin], %[t]\n\t"
                     "v_cmpx_gt_f32 %[tmax], %[t]\n\t"
                     "v_cmpx_gt_u64 %[best], %[cand]\n\t"
                     "v_mov_b64 %[best], %[cand]\n\t"
                     "s_mov_b64 exec, %[sv]"
                     : [sv] "=&s"(saved), [best] "+v"(best)
                     : [t] "v"(t), [cand] "v"(cand), [disc] "v"(disc), [tmin] "s"(t_min), [tmax] "s"(t_max)
                     : "vcc");
    }
    return best;
}

// the lane's best hit as one number (best_t_of, best_rec_of, best_key_of); a lane that is not alive reports a miss
template <bool FLAT>
RTF_DEV Best scan_best(const uint32_t *__restrict__ recs, const uint32_t *lrecs, const Layout &Y, const Ray &wr, float t_min, float t_max,
                       bool alive)
{
    if (FLAT) { // the stream is stored in runs (the host only asks for FLAT when it is)
        const Best b = scan_runs(recs, lrecs, Y, wr, t_min, t_max);
        return alive ? b : best_start(t_max);
    }
    Best best = best_start(t_max);
    float best_t = t_max; // the bound the tests below compare with: always the t of `best`
    int resume = 0;
    uint32_t cur_chain = 0;
    LocalRay w, l; // world ray, ray of the current chain
    set_local(w, wr.o, wr.d);
    l = w;
#ifdef RT_FAST_APPROX_T
    const float w_dd = fdot(wr.d, wr.d), w_inv_dd = rcp(w_dd);
#else
    const float w_dd = ref_dot(wr.d, wr.d), w_inv_dd = refined_rcp(w_dd);
#endif

    // A plain counted loop keeps i in an SGPR.  A lane that missed a box carries `resume`, the index
    // after that subtree; when every lane is skipping, a record costs one compare and one branch.
    SRec nxt = load_rec(recs, 0);
    for (int i = 0; i < Y.n_stream; ++i) {
        const SRec c = nxt;
        nxt = load_rec(recs, i + 1); // the table is padded: record n_stream exists
        bool on = alive;
        if (!FLAT) {
            on = alive & (resume <= i);
            if (__ballot(on) == 0ull) continue;
        }
        const uint32_t kw = c.d[7];
        const uint32_t kind = kw & 15u;
        if (kind == RT_REC_BOX_AUX) continue; // the second half of a box record
        if (kind != RT_REC_SPHERE) {
            const uint32_t chain = (kw >> RT_KW_CHAIN_SHIFT) & RT_KW_CHAIN_MASK;
            if (chain != cur_chain) {
                if (chain == 0) l = w;
                else apply_chain(recs, chain, wr, w.inv.y, l);
                cur_chain = chain;
            }
        }
        if (!FLAT && kind == RT_REC_INNER) { // aabb::hit, src/aabb.h:60-83
            const float x0 = (sf(c, 0) - l.o.x) * l.inv.x, x1 = (sf(c, 4) - l.o.x) * l.inv.x;
            const float y0 = (sf(c, 1) - l.o.y) * l.inv.y, y1 = (sf(c, 5) - l.o.y) * l.inv.y;
            const float z0 = (sf(c, 2) - l.o.z) * l.inv.z, z1 = (sf(c, 6) - l.o.z) * l.inv.z;
            const float tn = fmaxf(fmax3(fminf(x0, x1), fminf(y0, y1), fminf(z0, z1)), t_min);
            const float tf = fminf(fmin3(fmaxf(x0, x1), fmaxf(y0, y1), fmaxf(z0, z1)), t_max);
            const int skip = (int)c.d[3];
            resume = (on & !(tf > tn)) ? skip : resume;
            continue;
        }
        float t;
        bool in;
        if (kind == RT_REC_RECT) { // src/plane.h:9-29, 47-67, 109-129
            const uint32_t axis = (kw >> RT_KW_AXIS_SHIFT) & 3u;
            if (axis == 0) in = rect_in(c, l.o.z, l.d.z, l.inv.z, l.o.x, l.d.x, l.o.y, l.d.y, t_min, best_t, t);
            else if (axis == 1) in = rect_in(c, l.o.y, l.d.y, l.inv.y, l.o.x, l.d.x, l.o.z, l.d.z, t_min, best_t, t);
            else in = rect_in(c, l.o.x, l.d.x, l.inv.x, l.o.y, l.d.y, l.o.z, l.d.z, t_min, best_t, t);
        } else if (kind == RT_REC_BOX) { // src/box.h as one slab test (box_update_v)
            float tn, tf;
            slab_range(v3(sf(c, 0), sf(c, 1), sf(c, 2)), v3(sf(c, 4), sf(c, 5), sf(c, 6)), l, tn, tf);
            t = tn >= t_min ? tn : tf;
            in = (tn <= tf) & (t >= t_min) & (t <= best_t);
        } else { // sphere, world space: src/sphere.h:21-59 (strict range test), then the tie rule of the tree
            float disc, t1, t2;
            sphere_roots(wr.o, wr.d, w_dd, w_inv_dd, sf(c, 0), sf(c, 1), sf(c, 2), sf(c, 3), disc, t1, t2);
            const bool in1 = (t1 < t_max) & (t1 > t_min), in2 = (t2 < t_max) & (t2 > t_min);
            t = in1 ? t1 : t2;
            in = (disc > 0.0f) & (in1 | in2) & (t <= best_t);
        }
        // `in` holds t <= best_t; of two hits at the same t the one the reference's walk meets later wins (Best)
        const Best cand = best_of(t, RT_REC_KEY(kw, i));
        const bool take = in & on & (cand < best);
        best = take ? cand : best;
        best_t = take ? t : best_t;
    }
    return best;
}
template <bool FLAT>
RTF_DEV void scan(const uint32_t *__restrict__ recs, const uint32_t *lrecs, const Layout &Y, const Ray &wr, float t_min, float t_max,
                  bool alive, float &best_t, int &best_rec)
{
    const Best b = scan_best<FLAT>(recs, lrecs, Y, wr, t_min, t_max, alive);
    best_t = best_t_of(b), best_rec = best_rec_of(b);
}

// what shading needs about the hit, built after the scan from the winning record
struct Surface {
    V3 p, n;
    int face;   // BOX records: 2*axis + side of the face that was hit (rt_records.h); per-lane path: the rt_prim that was hit (a box's face rect)
    float u, v; // only where a texture reads them
    uint32_t mat_kind;
    int32_t texture;
    float p0;
    V3 color;
};
// The face of a box record that holds the hit point, and its outward normal in world space.  In the box's own axes
// (second record: centre and the image of the x axis) the hit point lies on the face whose plane it is nearest to.
RTF_DEV V3 box_normal(const Regs &R, int rec, V3 lo, V3 hi, V3 p, int &face)
{
    const int aux = rec + 1;
    const V3 c = v3(sh_f(R.d[0], aux), sh_f(R.d[1], aux), sh_f(R.d[2], aux));
    const float ax = sh_f(R.d[3], aux), az = sh_f(R.d[4], aux); // x axis -> (ax, 0, az), z axis -> (-az, 0, ax)
    const V3 q = v3(p.x - c.x, p.y - c.y, p.z - c.z);
    const float qx = fmaf(q.x, ax, q.z * az), qy = q.y, qz = fmaf(q.z, ax, -(q.x * az));
    const float ex = fabsf(qx) - 0.5f * (hi.x - lo.x), ey = fabsf(qy) - 0.5f * (hi.y - lo.y), ez = fabsf(qz) - 0.5f * (hi.z - lo.z);
    const bool is_x = (ex >= ey) & (ex >= ez), is_y = !is_x & (ey >= ez);
    const float sx = qx < 0.0f ? -1.0f : 1.0f, sy = qy < 0.0f ? -1.0f : 1.0f, sz = qz < 0.0f ? -1.0f : 1.0f;
    face = is_x ? (qx < 0.0f ? 0 : 1) : is_y ? (qy < 0.0f ? 2 : 3) : (qz < 0.0f ? 4 : 5);
    return is_x ? v3(sx * ax, 0.0f, sx * az) : is_y ? v3(0.0f, sy, 0.0f) : v3(-sz * az, 0.0f, sz * ax);
}
// normal of the record that was hit (f0..f6: its dwords 0-2 and 4-6).  Called by whole waves: the box part fetches
// across lanes, so it comes first, outside anything a lane decides for itself.
RTF_DEV V3 record_normal(const Regs &R, uint32_t kw, int rec, V3 p, float f0, float f1, float f2, float f4, float f5, float f6, int &face)
{
    const uint32_t kind = kw & 15u;
    V3 nb = v3(0, 0, 0);
    face = 0;
    if (__any(kind == RT_REC_BOX)) nb = box_normal(R, rec, v3(f0, f1, f2), v3(f4, f5, f6), p, face);
    face = kind == RT_REC_BOX ? face : 0;
    const float k = (kw & RT_KW_FLIP) ? -f4 : f4; // sphere: (p - c) / r, src/sphere.h:41, then flip_normals
    const V3 ns = v3((p.x - f0) * k, (p.y - f1) * k, (p.z - f2) * k);
    const bool xz = ((kw >> RT_KW_AXIS_SHIFT) & 3u) == 1u;
    const V3 nr = v3(f5, xz ? ((kw & RT_KW_FLIP) ? -1.0f : 1.0f) : 0.0f, f6);
    return kind == RT_REC_SPHERE ? ns : kind == RT_REC_BOX ? nb : nr;
}
// The same from the block's copy of the table in LDS (stage_records' layout), every lane for itself: the two halves of
// the record it hit (`key`: the low word of its best hit) and of its material record, four 16-byte reads where
// build_surface makes a dozen cross-lane fetches.  `p` is the hit point.
RTF_DEV void build_surface_lds(const uint32_t *lrecs, const Layout &Y, uint32_t key, V3 p, Surface &s)
{
    const int rec = (int)RT_KEY_INDEX(key);
    const uint32_t kind = key & 15u;
    const uint4 ra = *reinterpret_cast<const uint4 *>(lrecs + 8 * rec), rb = *reinterpret_cast<const uint4 *>(lrecs + 8 * rec + 4);
    const int m = Y.materials_base + (int)((key >> RT_KW_MAT_SHIFT) & RT_KW_MAT_MASK);
    const uint4 ma = *reinterpret_cast<const uint4 *>(lrecs + 8 * m);
    const uint2 mb = *reinterpret_cast<const uint2 *>(lrecs + 8 * m + 4);
    s.p = p, s.u = s.v = 0.0f, s.face = 0;
    const bool flip = (key & RT_KW_FLIP) != 0;
    V3 n;
    if (kind == RT_REC_SPHERE) { // (p - c) / r, src/sphere.h:41, then flip_normals
        const float k = flip ? -__uint_as_float(rb.y) : __uint_as_float(rb.y);
        n = v3((p.x - __uint_as_float(ra.x)) * k, (p.y - __uint_as_float(ra.y)) * k, (p.z - __uint_as_float(ra.z)) * k);
    } else if (kind == RT_REC_BOX) { // box_normal with the second record read from LDS
        const uint4 xa = *reinterpret_cast<const uint4 *>(lrecs + 8 * (rec + 1));
        const float az = __uint_as_float(lrecs[8 * (rec + 1) + 4]), ax = __uint_as_float(xa.w);
        const V3 lo = v3(__uint_as_float(ra.x), __uint_as_float(ra.y), __uint_as_float(ra.z));
        const V3 hi = v3(__uint_as_float(rb.x), __uint_as_float(rb.y), __uint_as_float(rb.z));
        const V3 q = v3(p.x - __uint_as_float(xa.x), p.y - __uint_as_float(xa.y), p.z - __uint_as_float(xa.z));
        const float qx = fmaf(q.x, ax, q.z * az), qy = q.y, qz = fmaf(q.z, ax, -(q.x * az));
        const float ex = fabsf(qx) - 0.5f * (hi.x - lo.x), ey = fabsf(qy) - 0.5f * (hi.y - lo.y), ez = fabsf(qz) - 0.5f * (hi.z - lo.z);
        const bool is_x = (ex >= ey) & (ex >= ez), is_y = !is_x & (ey >= ez);
        const float sx = qx < 0.0f ? -1.0f : 1.0f, sy = qy < 0.0f ? -1.0f : 1.0f, sz = qz < 0.0f ? -1.0f : 1.0f;
        s.face = is_x ? (qx < 0.0f ? 0 : 1) : is_y ? (qy < 0.0f ? 2 : 3) : (qz < 0.0f ? 4 : 5);
        n = is_x ? v3(sx * ax, 0.0f, sx * az) : is_y ? v3(0.0f, sy, 0.0f) : v3(-sz * az, 0.0f, sz * ax);
    } else { // a rect: its world normal (y is +-1 for an xz rect, 0 otherwise)
        const bool xz = ((key >> RT_KW_AXIS_SHIFT) & 3u) == 1u;
        n = v3(__uint_as_float(rb.z), xz ? (flip ? -1.0f : 1.0f) : 0.0f, __uint_as_float(rb.w));
    }
    s.n = n;
    s.mat_kind = ma.x;
    s.texture = (int32_t)ma.y;
    s.p0 = __uint_as_float(ma.z);
    s.color = v3(__uint_as_float(ma.w), __uint_as_float(mb.x), __uint_as_float(mb.y));
}
RTF_DEV void build_surface(const Regs &R, const Layout &Y, const Ray &wr, float t, int rec, Surface &s)
{
    const int src = rec < 0 ? 0 : rec;
    const uint32_t kw = sh_u(R.d[7], src);
    const float f0 = sh_f(R.d[0], src), f1 = sh_f(R.d[1], src), f2 = sh_f(R.d[2], src);
    const float f4 = sh_f(R.d[4], src), f5 = sh_f(R.d[5], src), f6 = sh_f(R.d[6], src);
    s.p = fma3(t, wr.d, wr.o);
    s.u = s.v = 0.0f; // the record table holds no texture that reads (u,v)
    s.n = record_normal(R, kw, src, s.p, f0, f1, f2, f4, f5, f6, s.face);
    const int m = Y.materials_base + (int)((kw >> RT_KW_MAT_SHIFT) & RT_KW_MAT_MASK);
    s.mat_kind = sh_u(R.d[0], m);
    s.texture = (int32_t)sh_u(R.d[1], m);
    s.p0 = sh_f(R.d[2], m);
    s.color = v3(sh_f(R.d[3], m), sh_f(R.d[4], m), sh_f(R.d[5], m));
}

// ---------------------------------------------------------------------------
// sampling (draw order as oracle/rt_oracle.cpp)
// ---------------------------------------------------------------------------
RTF_DEV V3 rnd_unit_sphere(Rng &rng) // src/utils.h:32-42
{
    V3 p;
    do {
        const float c = rng.next(), b = rng.next(), a = rng.next();
        p = v3(2.0f * a - 1.0f, 2.0f * b - 1.0f, 2.0f * c - 1.0f);
    } while (fdot(p, p) >= 1.0f);
    return p;
}
RTF_DEV V3 rnd_unit_disk(Rng &rng) // src/utils.h:44-55
{
    V3 p;
    do {
        const float b = rng.next(), a = rng.next();
        p = v3(2.0f * a - 1.0f, 2.0f * b - 1.0f, 0.0f);
    } while (fdot(p, p) >= 1.0f);
    return p;
}
struct Onb {
    V3 u, v, w;
    // src/vec3.h:209-224 with the cross products against (0,1,0) / (1,0,0) written out:
    // w x (0,1,0) = (-w.z, 0, w.x) and w x (1,0,0) = (0, w.z, -w.y)
    RTF_DEV void build_from_w(V3 n)
    {
        w = funit(n);
        const bool big = fabsf(w.x) > 0.9f;
        const float other = big ? w.x : w.y;
        const float k = rsq(fmaf(other, other, w.z * w.z));
        v = v3(big ? -w.z * k : 0.0f, big ? 0.0f : w.z * k, (big ? w.x : -w.y) * k);
        u = fcross(w, v);
    }
    RTF_DEV V3 local(V3 a) const
    {
        return v3(fmaf(a.z, w.x, fmaf(a.y, v.x, a.x * u.x)), fmaf(a.z, w.y, fmaf(a.y, v.y, a.x * u.y)),
                  fmaf(a.z, w.z, fmaf(a.y, v.z, a.x * u.z)));
    }
};

// hitable_list::pdf_value over the light records, src/hitable_list.h:67-77, for a direction of UNIT length (the only
// caller has just normalised it: dot(v,v), its root and its reciprocal drop out of the reference's formulas)
RTF_DEV bool within(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi) == x; } // lo <= x <= hi, false for NaN
// one light record (rt_records.cpp: emit_lights) for a direction of unit length
RTF_DEV float light_rec_pdf(const SRec &c, V3 o, V3 v)
{
    const uint32_t kind = c.d[7] & 15u;
    float pdf = 0.0f;
    if (kind == RT_REC_LIGHT_RECT) { // xz_rect::pdf_value, src/plane.h:76-91
        const float iy = rcp(v.y);
        const float t = (sf(c, 4) - o.y) * iy;
        const float x = fmaf(t, v.x, o.x), z = fmaf(t, v.z, o.z);
        const bool in = (t >= 0.0001f) & within(x, sf(c, 0), sf(c, 1)) & within(z, sf(c, 2), sf(c, 3));
        pdf = in ? t * t * (fabsf(iy) * sf(c, 6)) : 0.0f; // t^2 / (|v.y| area): the record carries 1 / area beside the area
    } else if (kind == RT_REC_LIGHT_SPHERE) { // sphere::pdf_value, src/sphere.h:68-83
        const V3 oc = v3(o.x - sf(c, 0), o.y - sf(c, 1), o.z - sf(c, 2));
        const float radius = sf(c, 3);
        const float b = fdot(oc, v), occ = fdot(oc, oc);
        const float disc = fmaf(b, b, fmaf(radius, radius, -occ));
        const bool in = (disc > 0.0f) & (fsqrt(disc) - b > 0.001f); // the larger root decides (t1 <= t2)
        const float cos_theta_max = fsqrt(1.0f - radius * radius * rcp(occ));
        pdf = in ? rcp(2.0f * RT_PI * (1.0f - cos_theta_max)) : 0.0f;
    }
    return pdf;
}
RTF_DEV float lights_pdf_value(const uint32_t *__restrict__ recs, const Layout &Y, V3 o, V3 v)
{
    const float weight = Y.inv_n_lights;
    float sum = 0.0f;
    for (int k = 0; k < Y.n_lights; ++k) sum = fmaf(weight, light_rec_pdf(load_rec(recs, Y.lights_base + k), o, v), sum);
    return sum;
}
// hitable_list::random, src/hitable_list.h:79-88, first half: the member is chosen per lane.  The (few) light
// records are read with wave-uniform scalar loads and each lane keeps the one it drew.  (ds_bpermute from the
// register-resident table must not be used here: under a divergent branch it returns 0 for records held by
// inactive lanes.)
struct LightPick {
    uint32_t kind;
    float f0, f1, f2, f3, f4;
};
RTF_DEV int lights_index(const Layout &Y, Rng &rng)
{
    int index = (int)(rng.next() * (float)Y.n_lights);
    if (index == Y.n_lights) index--;
    return index;
}
RTF_DEV LightPick light_of_index(const uint32_t *__restrict__ recs, const Layout &Y, int index)
{
    LightPick p{RT_REC_LIGHT_NONE, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    for (int k = 0; k < Y.n_lights; ++k) {
        const SRec c = load_rec(recs, Y.lights_base + k);
        const bool mine = index == k;
        p.kind = mine ? (c.d[7] & 15u) : p.kind;
        p.f0 = mine ? sf(c, 0) : p.f0, p.f1 = mine ? sf(c, 1) : p.f1, p.f2 = mine ? sf(c, 2) : p.f2;
        p.f3 = mine ? sf(c, 3) : p.f3, p.f4 = mine ? sf(c, 4) : p.f4;
    }
    return p;
}
// the same from the block's copy of the table in LDS: every lane reads the record it drew
RTF_DEV LightPick light_of_index_lds(const uint32_t *lrecs, const Layout &Y, int index)
{
    const uint32_t *r = lrecs + 8 * (Y.lights_base + index);
    const uint4 a = *reinterpret_cast<const uint4 *>(r);
    return LightPick{r[7] & 15u, __uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w), __uint_as_float(r[4])};
}
RTF_DEV LightPick lights_pick(const uint32_t *__restrict__ recs, const Layout &Y, Rng &rng)
{
    return light_of_index(recs, Y, lights_index(Y, rng));
}

// Textures for the fast paths: rt::tex_value (src/texture.h, src/perlin.h) with hardware sine and with the perlin
// loops kept as loops (fully unrolled, the eight corner gradients of seven octaves cost the kernel 50 more registers).
// UNROLLED: the eight corners written out (below); kernels that are short of registers keep the loop.
template <bool UNROLLED = false>
RTF_DEV float fast_perlin_noise(V3 p) // src/perlin.h:54-121
{
    const float fx = floorf(p.x), fy = floorf(p.y), fz = floorf(p.z);
    const float u = p.x - fx, v = p.y - fy, w = p.z - fz;
    const int i = (int)fx, j = (int)fy, k = (int)fz;
    const float uu = u * u * (3.0f - 2.0f * u), vv = v * v * (3.0f - 2.0f * v), ww = w * w * (3.0f - 2.0f * w);
    if (!UNROLLED) { // one corner per turn of a loop, its index decoded at run time
    float accum = 0.0f;
#pragma nounroll
    for (int c = 0; c < 8; ++c) {
        const int di = c >> 2, dj = (c >> 1) & 1, dk = c & 1;
        const uint32_t h = 1619u * (uint32_t)(i + di) + 31337u * (uint32_t)(j + dj) + 6971u * (uint32_t)(k + dk) + 1013u;
        int index = (int)h;
        index ^= (index >> 8);
        index &= 0xff;
        const float4 g = rt::k_perlin_gradients[index];
        const float wi = di ? uu : 1.0f - uu, wj = dj ? vv : 1.0f - vv, wk = dk ? ww : 1.0f - ww;
        accum = fmaf(wi * wj * wk, fmaf(g.z, w - (float)dk, fmaf(g.y, v - (float)dj, g.x * (u - (float)di))), accum);
    }
    return accum;
    }
    // The eight corners written out: three multiplies for the hash of the cell, the corners' hashes by additions, weights and
    // offsets shared between corners, and the eight gradient fetches in flight together -- 120 instructions and one memory
    // latency per octave where the loop took 200 and eight.
    const uint32_t hx0 = 1619u * (uint32_t)i, hy0 = 31337u * (uint32_t)j, hz0 = 6971u * (uint32_t)k + 1013u;
    const uint32_t hxy[4] = {hx0 + hy0, hx0 + hy0 + 31337u, hx0 + 1619u + hy0, hx0 + 1619u + hy0 + 31337u}; // [2 * di + dj]
    const uint32_t hz[2] = {hz0, hz0 + 6971u};
    const float wx[2] = {1.0f - uu, uu}, wy[2] = {1.0f - vv, vv}, wz[2] = {1.0f - ww, ww};
    const float ox[2] = {u, u - 1.0f}, oy[2] = {v, v - 1.0f}, oz[2] = {w, w - 1.0f};
    float4 g[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const uint32_t h = hxy[c >> 1] + hz[c & 1];
        g[c] = rt::k_perlin_gradients[(h ^ (uint32_t)((int)h >> 8)) & 0xffu];
    }
    float accum = 0.0f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int di = c >> 2, dj = (c >> 1) & 1, dk = c & 1;
        accum = fmaf(wx[di] * wy[dj] * wz[dk], fmaf(g[c].z, oz[dk], fmaf(g[c].y, oy[dj], g[c].x * ox[di])), accum);
    }
    return accum;
}
template <bool UNROLLED = false>
RTF_DEV float fast_perlin_turb(V3 p) // src/perlin.h:123-136, depth 7
{
    float accum = 0.0f, weight = 1.0f;
#pragma nounroll
    for (int i = 0; i < 7; ++i) {
        accum = fmaf(weight, fast_perlin_noise<UNROLLED>(p), accum);
        weight *= 0.5f;
        p = v3(p.x * 2.0f, p.y * 2.0f, p.z * 2.0f);
    }
    return fabsf(accum);
}
// The value of a noise texture (src/texture.h:47-53) at ONE point, by the whole wave: the 7 octaves x 8 corners of the
// turbulence are 56 independent terms; lane L < 56 computes term (octave L >> 3, corner L & 7) and the wave adds them
// up.  About 60 vector instructions for the wave, where fast_perlin_turb costs the wave 1400 whatever the number of lanes
// that ask (a perlin sphere that covers 3 % of the image is in three of four shade phases of 48 lanes).  Called by all
// 64 lanes with wave-uniform arguments; the terms are those of fast_perlin_noise, summed in another order.
RTF_DEV float wave_noise_value(float scale, float px, float py, float pz, int lane)
{
    // what follows depends on the lane alone (octave, weight, corner): the compiler must not make it once per kernel and
    // hold it in a dozen registers through every walk
    asm volatile("" : "+v"(lane));
    const int o = lane >> 3;
    const float f = __uint_as_float((uint32_t)(127 + o) << 23);                         // 2^octave: p doubles per octave,
    const float weight = lane < 56 ? __uint_as_float((uint32_t)(127 - o) << 23) : 0.0f; // its weight halves
    const float x = scale * px * f, y = scale * py * f, z = scale * pz * f;
    const float fx = floorf(x), fy = floorf(y), fz = floorf(z);
    const float u = x - fx, v = y - fy, w = z - fz;
    const float uu = u * u * (3.0f - 2.0f * u), vv = v * v * (3.0f - 2.0f * v), ww = w * w * (3.0f - 2.0f * w);
    const int di = (lane >> 2) & 1, dj = (lane >> 1) & 1, dk = lane & 1;
    const uint32_t h = 1619u * (uint32_t)((int)fx + di) + 31337u * (uint32_t)((int)fy + dj) + 6971u * (uint32_t)((int)fz + dk) + 1013u;
    int index = (int)h;
    index ^= (index >> 8);
    index &= 0xff;
    const float4 g = rt::k_perlin_gradients[index];
    const float wi = di ? uu : 1.0f - uu, wj = dj ? vv : 1.0f - vv, wk = dk ? ww : 1.0f - ww;
    float term = weight * (wi * wj * wk) * fmaf(g.z, w - (float)dk, fmaf(g.y, v - (float)dj, g.x * (u - (float)di)));
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) term += __shfl_xor(term, m, 64);
    return 0.5f * (1.0f + __sinf(fmaf(10.0f, fabsf(term), scale * pz)));
}
template <bool UNROLLED = false>
RTF_DEV V3 fast_tex_value(const DevScene &S, int idx, float u, float v, V3 p)
{
    rt_texture t = S.textures[idx];
    for (int guard = 0; guard < 8 && t.kind == RT_TEX_CHECKER; ++guard) { // src/texture.h:26-37
        const float sines = __sinf(10.0f * p.x) * __sinf(10.0f * p.y) * __sinf(10.0f * p.z);
        t = S.textures[sines < 0.0f ? t.a : t.b];
    }
    if (t.kind == RT_TEX_CONSTANT) return v3(t.color[0], t.color[1], t.color[2]);
    if (t.kind == RT_TEX_NOISE) { // src/texture.h:47-53
        const float c = 0.5f * (1.0f + __sinf(fmaf(10.0f, fast_perlin_turb<UNROLLED>(v3(t.scale * p.x, t.scale * p.y, t.scale * p.z)), t.scale * p.z)));
        return v3(c, c, c);
    }
    if (t.kind == RT_TEX_IMAGE) { // src/texture.h:63-77
        int i = (int)(u * (float)t.nx);
        int j = (int)((1.0f - v) * (float)t.ny - 0.001f);
        i = i < 0 ? 0 : i, j = j < 0 ? 0 : j;
        i = i > t.nx - 1 ? t.nx - 1 : i, j = j > t.ny - 1 ? t.ny - 1 : j;
        const uint8_t *d = S.texels + t.offset + 3 * i + 3 * t.nx * j;
        return v3((float)d[0] * (1.0f / 255.0f), (float)d[1] * (1.0f / 255.0f), (float)d[2] * (1.0f / 255.0f));
    }
    return v3(0, 0, 0); // a checker nested deeper than 8 levels
}

// lambertian::scatter + the mixture pdf of color() (src/material.h:35-50, src/main.cpp:104-121), after the coin
// (`to_light`) and, for a lane that samples the important list, the member (`lp`) have been drawn.
// mixture_pdf::generate (src/pdf.h:71-79): half the lanes sample the important list, half the cosine lobe.  A sphere
// light (sphere::random + random_to_sphere, src/sphere.h:85-103) and the cosine lobe (cosine_pdf::generate,
// random_cosine_direction with its factor 2, src/vec3.h:179-180) are the same computation around different axes -- an
// orthonormal basis, two draws, a point on a cap -- so those lanes run it together; a rect light (xz_rect::random,
// src/plane.h:94-98, z takes the first draw) needs no basis.
RTF_DEV float cos_over_pi(V3 wn, V3 dir) { return fmaxf(fdot(wn, dir), 0.0f) * (1.0f / RT_PI); } // both of unit length
RTF_DEV void scatter_lambert(const uint32_t *__restrict__ recs, const Layout &Y, const Surface &h, V3 albedo, bool to_light,
                             const LightPick &lp, Rng &rng, V3 &nd, V3 &mul, bool &go)
{
    const V3 wn = h.n; // of unit length: rect and box normals by construction, a sphere's (p - c) / r to rounding
    V3 dir = v3(1, 0, 0); // hitable::random, src/hitable.h:20
    if (to_light & (lp.kind == RT_REC_LIGHT_RECT)) {
        const float rz = rng.next(), rx = rng.next();
        dir = v3(fmaf(rx, lp.f1 - lp.f0, lp.f0) - h.p.x, lp.f4 - h.p.y, fmaf(rz, lp.f3 - lp.f2, lp.f2) - h.p.z);
    }
    const bool sphere_light = to_light & (lp.kind == RT_REC_LIGHT_SPHERE);
    if (!to_light | sphere_light) {
        const V3 axis = sphere_light ? v3(lp.f0 - h.p.x, lp.f1 - h.p.y, lp.f2 - h.p.z) : wn;
        Onb uvw;
        uvw.build_from_w(axis);
        const float r1 = rng.next();
        const float r2 = rng.next();
        float s, c;
        sincos_2pi(r1, s, c);
        float z, q;
        if (sphere_light) {
            z = fmaf(r2, fsqrt(1.0f - lp.f3 * lp.f3 * rcp(fdot(axis, axis))) - 1.0f, 1.0f);
            q = fsqrt(1.0f - z * z);
        } else {
            z = fsqrt(1.0f - r2);
            q = 2.0f * fsqrt(r2);
        }
        dir = uvw.local(v3(c * q, s * q, z));
    }
    dir = funit(dir);
    // cosine_pdf::value (src/pdf.h:23-30) and lambertian::scattering_pdf (src/material.h:35-41) are the same number
    // once the normal has unit length: max(0, cos) / pi
    const float sc = cos_over_pi(wn, dir);
    const float pdf_val = 0.5f * lights_pdf_value(recs, Y, h.p, dir) + 0.5f * sc;
    const float wgt = sc * rcp(pdf_val);
    nd = dir;
    mul = v3(albedo.x * wgt, albedo.y * wgt, albedo.z * wgt);
    go = wgt == wgt; // 0/0 is the NaN the reference's de_nan zeroes
}

// metal / dielectric / isotropic ::scatter (src/material.h:65-78, 92-145, 187-196); `mul` comes in as the albedo
RTF_DEV void scatter_specular(const Surface &h, V3 rd, Rng &rng, V3 &nd, V3 &mul, float &ntime)
{
    if (h.mat_kind == RT_MAT_METAL) {
        const V3 ud = funit(rd);
        const float k = 2.0f * fdot(ud, h.n);
        V3 fz = v3(0, 0, 0); // at fuzz 0 the sphere sample is not drawn (include/rt_rng.h)
        if (h.p0 != 0.0f) fz = rnd_unit_sphere(rng);
        nd = v3(fmaf(h.p0, fz.x, ud.x - k * h.n.x), fmaf(h.p0, fz.y, ud.y - k * h.n.y), fmaf(h.p0, fz.z, ud.z - k * h.n.z));
    } else if (h.mat_kind == RT_MAT_DIELECTRIC) {
        const float inv_len = rsq(fdot(rd, rd));
        const V3 ud = v3(rd.x * inv_len, rd.y * inv_len, rd.z * inv_len);
        const float dn = fdot(rd, h.n);
        const bool inside = dn > 0.0f;
        const float ref_idx = h.p0;
        const float ni_over_nt = inside ? ref_idx : rcp(ref_idx);
        const V3 on = inside ? v3(-h.n.x, -h.n.y, -h.n.z) : h.n;
        const float cosine = (inside ? ref_idx * dn : -dn) * inv_len;
        const float dt = fdot(ud, on);
        const float disc = 1.0f - ni_over_nt * ni_over_nt * (1.0f - dt * dt);
        float reflect_prob = 1.0f;
        if (disc > 0.0f) { // schlick, src/utils.h:4-9
            float r0 = (1.0f - ref_idx) * rcp(1.0f + ref_idx);
            r0 = r0 * r0;
            const float x = 1.0f - cosine, x2 = x * x;
            reflect_prob = fmaf(1.0f - r0, x2 * x2 * x, r0);
        }
        // reflect (src/utils.h:11-14) and refract (src/utils.h:16-30) as one expression a*ud + b*n:
        const bool reflect = rng.next() < reflect_prob;
        const float k = 2.0f * fdot(ud, h.n);
        const float sq = fsqrt(fmaxf(disc, 0.0f));
        const float ca = reflect ? 1.0f : ni_over_nt;
        const float cb = reflect ? -k : -(ni_over_nt * dt + sq);
        const V3 nn = reflect ? h.n : on;
        nd = v3(fmaf(ca, ud.x, cb * nn.x), fmaf(ca, ud.y, cb * nn.y), fmaf(ca, ud.z, cb * nn.z));
        mul = v3(1, 1, 1); // attenuation (1,1,1), src/material.h:100
    } else { // isotropic
        nd = rnd_unit_sphere(rng);
        ntime = 0.0f;
    }
}

// One bounce of color() (src/main.cpp:91-127) in throughput form; see rt::shade for the derivation.
// TEX: the scene has a non-constant texture (noise / checker / image); those are evaluated from the memory tables.
// Every branch only PRODUCES (next direction, throughput factor); ray and throughput are written once, after the
// branches have joined.  (Writing them inside the divergent branches makes the compiler shuttle the whole path state
// through copies at every join.)
template <bool TEX, bool NOISE_UNROLLED = false>
RTF_DEV bool shade(const DevScene &S, const uint32_t *__restrict__ recs, const Layout &Y, bool hit_any, const Surface &h, Ray &r, V3 &T,
                   int &depth, int max_depth, Rng &rng, V3 &L)
{
    // what happens to the path is settled first, with selects; then come FLAT branches, one per fate (nested
    // branches make the compiler re-initialise every value that leaves them at every level of the nest)
    enum { END = 0, LIGHT = 1, DIFFUSE = 2, SPECULAR = 3 };
    const bool is_light = h.mat_kind == RT_MAT_DIFFUSE_LIGHT;
    const int fate = !hit_any ? END : is_light ? LIGHT : !(depth < max_depth) ? END : h.mat_kind == RT_MAT_LAMBERTIAN ? DIFFUSE : SPECULAR;
    V3 albedo = h.color;
    if (TEX && hit_any && h.texture >= 0) albedo = fast_tex_value<NOISE_UNROLLED>(S, h.texture, h.u, h.v, h.p);
    L = fate == LIGHT ? v3(T.x * albedo.x, T.y * albedo.y, T.z * albedo.z) : v3(0, 0, 0);
    bool go = fate >= DIFFUSE;
    V3 nd = r.d, mul = albedo;
    float ntime = r.time;
    if (fate == DIFFUSE) {
        const bool to_light = rng.next() < 0.5f;
        LightPick lp{RT_REC_LIGHT_NONE, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        if (to_light) lp = lights_pick(recs, Y, rng);
        scatter_lambert(recs, Y, h, albedo, to_light, lp, rng, nd, mul, go);
    }
    if (fate == SPECULAR) scatter_specular(h, r.d, rng, nd, mul, ntime);
    depth += go ? 1 : 0;
    r = Ray{h.p, nd, ntime};
    T = v3(T.x * mul.x, T.y * mul.y, T.z * mul.z);
    // a zero weight is final: every later factor multiplies it
    return go && (T.x != 0.0f || T.y != 0.0f || T.z != 0.0f);
}

// camera::get_ray, src/camera.h:54-68
RTF_DEV Ray camera_ray(const rt_camera &c, float s, float t, V3 disk, float time01)
{
    const float rx = c.lens_radius * disk.x, ry = c.lens_radius * disk.y;
    const V3 oo = v3(fmaf(c.v[0], ry, fmaf(c.u[0], rx, c.origin[0])), fmaf(c.v[1], ry, fmaf(c.u[1], rx, c.origin[1])),
                     fmaf(c.v[2], ry, fmaf(c.u[2], rx, c.origin[2])));
    const V3 d = v3(fmaf(t, c.vertical[0], fmaf(s, c.horizontal[0], c.lower_left_corner[0])) - oo.x,
                    fmaf(t, c.vertical[1], fmaf(s, c.horizontal[1], c.lower_left_corner[1])) - oo.y,
                    fmaf(t, c.vertical[2], fmaf(s, c.horizontal[2], c.lower_left_corner[2])) - oo.z);
    return Ray{oo, d, fmaf(time01, c.time1 - c.time0, c.time0)};
}

} // namespace rtf
#endif // RT_FAST_HPP_
