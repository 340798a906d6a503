// rt_oracle.cpp -- TEST INFRASTRUCTURE: CPU restatement of the reference's hot path.
//
// This file restates, for the flattened scene description of include/rt_amd.h,
// the algorithm of droune2001/ray_tracing's per-pixel Monte Carlo loop:
// color() (src/main.cpp:91-127), compute_one_sample_task::run
// (src/main.cpp:217-245), the resolve loop (src/main.cpp:708-743) and everything
// they call.  Each function cites the reference lines it follows.  It exists to
// CHECK the HIP path; it is never the thing measured or shipped.  Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
//
// Parity status: PINNED.  In RNG mode ORC_RNG_MINSTD this code reproduces the
// compiled reference (oracle/_ref, g++ -O2, same libstdc++) BIT FOR BIT: the
// per-sample float buffers, the primary-hit records and the per-function
// known-answer vectors in tests/golden/ were all written by the reference's own
// code (oracle/ref_harness_tail.cpp, oracle/make_golden.py) and
// tests/test_oracle_pinned.py compares against them exactly.  That requires
// the same expression shapes, the same operation order, no FMA contraction
// (-ffp-contract=off; the reference build has no FMA either) and the same order
// of RAN01() draws, including g++'s right-to-left evaluation of constructor
// arguments such as vec3(RAN01(), RAN01(), RAN01()).
//
// In mode ORC_RNG_COUNTER the same code draws from the per-path stream of
// include/rt_rng.h instead, in the same order; that is the stream the GPU
// kernels use, so GPU images can be compared with this oracle pixel by pixel.
#include <math.h>

#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "rt_amd.h"
#include "rt_rng.h"

#define ORC_PI 3.14159f // src/main.cpp:54-56

namespace {

// ---------------------------------------------------------------------------
// vec3 (src/vec3.h).  Operator shapes kept: v*t is (t*x, t*y, t*z), dot is
// left-associated, cross has the negated middle term.
// ---------------------------------------------------------------------------
struct V3 {
    float x, y, z;
};
inline V3 v3(float a, float b, float c) { return V3{a, b, c}; }
inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline V3 operator*(V3 a, float t) { return v3(t * a.x, t * a.y, t * a.z); }
inline V3 operator*(float t, V3 a) { return v3(t * a.x, t * a.y, t * a.z); }
inline V3 operator/(V3 a, float t) { return v3(a.x / t, a.y / t, a.z / t); }
inline V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float sqlen(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
inline float len(V3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
inline V3 unit(V3 a) { return a / len(a); } // src/vec3.h:157-160
inline V3 cross(V3 a, V3 b)                 // src/vec3.h:102-107
{
    return v3((a.y * b.z - a.z * b.y), (-(a.x * b.z - a.z * b.x)), (a.x * b.y - a.y * b.x));
}
inline float comp(V3 a, int i) { return i == 0 ? a.x : i == 1 ? a.y : a.z; }

struct Ray {
    V3 o, d;
    float time;
};
inline V3 at(const Ray &r, float t) { return r.o + t * r.d; } // src/ray.h:14

struct Onb { // src/vec3.h:195-224
    V3 u, v, w;
    void build_from_w(V3 n)
    {
        w = unit(n);
        V3 a = fabsf(w.x) > 0.9f ? v3(0, 1, 0) : v3(1, 0, 0);
        v = unit(cross(w, a));
        u = cross(w, v);
    }
    V3 local(V3 a) const { return a.x * u + a.y * v + a.z * w; }
};

// ---------------------------------------------------------------------------
// random numbers
// ---------------------------------------------------------------------------
enum { ORC_RNG_MINSTD = 0, ORC_RNG_COUNTER = 1 };

struct Rng {
    int mode = ORC_RNG_COUNTER;
    // MINSTD: the reference's global engine + distribution (src/main.cpp:62-63)
    std::default_random_engine *engine = nullptr;
    std::uniform_real_distribution<float> *dist = nullptr;
    // COUNTER: include/rt_rng.h
    uint32_t key = 0, pixel = 0, n = 0;
    float next()
    {
        if (mode == ORC_RNG_MINSTD) return (*dist)(*engine);
        return rt_draw(key, pixel, n++);
    }
};

// work counters (what SURVEY.md 3.3 measured on the reference)
struct Counters {
    uint64_t rays = 0, slab_tests = 0, rect_tests = 0, sphere_tests = 0, scatters = 0, paths = 0;
    uint64_t end_miss = 0, end_light = 0, end_depth = 0;
};
thread_local Counters tl_counters;
Counters g_counters;
const float *g_perlin_table = nullptr; // 256 x 3 floats, set by orc_set_perlin_table

// ---------------------------------------------------------------------------
// geometry
// ---------------------------------------------------------------------------
struct Rec { // hit_record, src/hitable.h:6-13, plus the ids the flattener adds
    float t = 0, u = 0, v = 0;
    V3 p{0, 0, 0}, n{0, 0, 0};
    int material = -1, prim = -1;
};

// aabb::hit, src/aabb.h:60-83
bool aabb_hit(const float lo[3], const float hi[3], const Ray &r, float t_min, float t_max)
{
    ++tl_counters.slab_tests;
    for (int a = 0; a < 3; ++a) {
        float invD = 1.0f / comp(r.d, a);
        float t0 = (lo[a] - comp(r.o, a)) * invD;
        float t1 = (hi[a] - comp(r.o, a)) * invD;
        if (invD < 0.0f) {
            float s = t0;
            t0 = t1;
            t1 = s;
        }
        t_min = t0 > t_min ? t0 : t_min;
        t_max = t1 < t_max ? t1 : t_max;
        if (t_max <= t_min) return false;
    }
    return true;
}

// get_sphere_uv, src/utils.h:57-68 (note (float)M_PI here, not PI)
void sphere_uv(V3 p, float &u, float &v)
{
    float phi = atan2(p.z, p.x);
    float theta = asin(p.y);
    float two_pi = (2.0f * (float)M_PI);
    float pi_over_two = ((float)M_PI / 2.0f);
    float phi_minus_pi = (phi + (float)M_PI);
    float phi_minus_pi_over_two_pi = (phi_minus_pi / two_pi);
    u = 1.0f - phi_minus_pi_over_two_pi;
    v = (theta + pi_over_two) / (float)M_PI;
}

// sphere::hit src/sphere.h:21-59 and moving_sphere::hit src/sphere.h:147-183
bool sphere_hit(V3 center, float radius, bool with_uv, const Ray &r, float t_min, float t_max, Rec &rec)
{
    ++tl_counters.sphere_tests;
    V3 oc = r.o - center;
    float a = dot(r.d, r.d);
    float b = dot(oc, r.d);
    float c = dot(oc, oc) - radius * radius;
    float discriminant = b * b - a * c;
    if (discriminant > 0.0f) {
        float temp = (-b - sqrtf(discriminant)) / a;
        if (!(temp < t_max && temp > t_min)) {
            temp = (-b + sqrtf(discriminant)) / a;
            if (!(temp < t_max && temp > t_min)) return false;
        }
        rec.t = temp;
        rec.p = at(r, rec.t);
        rec.n = (rec.p - center) / radius;
        rec.u = rec.v = 0.0f;
        if (with_uv) sphere_uv(rec.n, rec.u, rec.v);
        return true;
    }
    return false;
}

// xy_rect / xz_rect / yz_rect ::hit, src/plane.h:9-29, 47-67, 109-129.
// f = a0 a1 b0 b1 k.  kind selects which axes a,b,k are.
bool rect_hit(int kind, const float *f, const Ray &r, float t0, float t1, Rec &rec)
{
    ++tl_counters.rect_tests;
    int ka, kb, kk; // axis of the first range, the second range, the plane
    if (kind == RT_PRIM_XY_RECT) ka = 0, kb = 1, kk = 2;
    else if (kind == RT_PRIM_XZ_RECT) ka = 0, kb = 2, kk = 1;
    else ka = 1, kb = 2, kk = 0;
    float t = (f[4] - comp(r.o, kk)) / comp(r.d, kk);
    if (t < t0 || t > t1) return false;
    float a = comp(r.o, ka) + t * comp(r.d, ka);
    float b = comp(r.o, kb) + t * comp(r.d, kb);
    // yz_rect tests z before y (src/plane.h:118); with plain comparisons the order cannot matter
    if (a < f[0] || a > f[1] || b < f[2] || b > f[3]) return false;
    rec.u = (a - f[0]) / (f[1] - f[0]);
    rec.v = (b - f[2]) / (f[3] - f[2]);
    rec.t = t;
    rec.p = at(r, t);
    rec.n = kk == 2 ? v3(0, 0, 1) : kk == 1 ? v3(0, 1, 0) : v3(1, 0, 0);
    return true;
}

// translate::hit src/transforms.h:14-28, rotate_y::hit src/transforms.h:102-130 : the way down
Ray chain_down(const rt_scene_desc *sc, int chain, Ray r)
{
    const rt_chain &c = sc->chains[chain];
    for (int i = 0; i < c.count; ++i) {
        const rt_xform &x = sc->xforms[c.first + i];
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
// ... and the way back up (innermost wrapper first)
void chain_up(const rt_scene_desc *sc, int chain, Rec &rec)
{
    const rt_chain &c = sc->chains[chain];
    for (int i = c.count - 1; i >= 0; --i) {
        const rt_xform &x = sc->xforms[c.first + i];
        if (x.kind == RT_XFORM_TRANSLATE) {
            rec.p = rec.p + v3(x.a, x.b, x.c);
        } else {
            float s = x.a, co = x.b;
            V3 p = rec.p, n = rec.n;
            p.x = co * rec.p.x + s * rec.p.z;
            p.z = -s * rec.p.x + co * rec.p.z;
            n.x = co * rec.n.x + s * rec.n.z;
            n.z = -s * rec.n.x + co * rec.n.z;
            rec.p = p, rec.n = n;
        }
    }
}

// one plain primitive in its own space
bool plain_hit(const rt_prim &pr, const Ray &lr, float t_min, float t_max, Rec &rec)
{
    bool h;
    switch (pr.kind) {
    case RT_PRIM_SPHERE:
        h = sphere_hit(v3(pr.f[0], pr.f[1], pr.f[2]), pr.f[3], true, lr, t_min, t_max, rec);
        break;
    case RT_PRIM_MOVING_SPHERE: { // center(t), src/sphere.h:142-145
        V3 c0 = v3(pr.f[0], pr.f[1], pr.f[2]), c1 = v3(pr.f[3], pr.f[4], pr.f[5]);
        V3 c = c0 + ((lr.time - pr.f[7]) / (pr.f[8] - pr.f[7])) * (c1 - c0);
        h = sphere_hit(c, pr.f[6], false, lr, t_min, t_max, rec);
        break;
    }
    default:
        h = rect_hit(pr.kind, pr.f, lr, t_min, t_max, rec);
    }
    if (h) {
        if (pr.flags & RT_PRIM_FLIP) rec.n = -rec.n; // flip_normals::hit, src/hitable.h:27-38
        rec.material = pr.material;
    }
    return h;
}

// any primitive, ray given in WORLD space
bool prim_hit(const rt_scene_desc *sc, int idx, const Ray &wr, float t_min, float t_max, Rec &rec, Rng &rng)
{
    const rt_prim &pr = sc->prims[idx];
    Ray lr = chain_down(sc, pr.chain, wr);
    bool h = false;
    if (pr.kind == RT_PRIM_GROUP) { // hitable_list::hit, src/hitable_list.h:18-34 (box::hit src/box.h:20-23)
        Rec tmp;
        float closest = t_max;
        for (int i = 0; i < pr.count; ++i) {
            if (plain_hit(sc->prims[pr.first + i], lr, t_min, closest, tmp)) {
                h = true;
                closest = tmp.t;
                rec = tmp;
                rec.prim = pr.first + i;
            }
        }
    } else if (pr.kind == RT_PRIM_MEDIUM) { // constant_medium::hit, src/volume.h:23-53
        Rec r1, r2;
        if (prim_hit(sc, pr.first, wr, -FLT_MAX, FLT_MAX, r1, rng)) {
            if (prim_hit(sc, pr.first, wr, r1.t + 0.0001f, FLT_MAX, r2, rng)) {
                if (r1.t < t_min) r1.t = t_min;
                if (r1.t > t_max) r1.t = t_max;
                if (r1.t >= r2.t) return false;
                if (r1.t < 0) r1.t = 0;
                // src/volume.h:41 assigns rec2.t = rec1.t and never uses the distance: no exit test
                float hit_distance = -(1.0f / pr.f[0]) * logf(rng.next());
                rec.t = r1.t + hit_distance / len(lr.d);
                rec.p = at(lr, rec.t);
                rec.n = v3(1, 0, 0);
                rec.u = rec.v = 0.0f;
                rec.material = pr.material;
                rec.prim = idx;
                h = true;
            }
        }
    } else {
        h = plain_hit(pr, lr, t_min, t_max, rec);
        if (h) rec.prim = idx;
    }
    if (h) chain_up(sc, pr.chain, rec);
    return h;
}

// world->hit(): bvh_node::hit, src/bvh.h:123-163, as one pass over the pre-order
// entries.  The recursion tests both children with the SAME t_max and keeps the
// nearer, the right one on a tie; scanning leaves left to right and replacing the
// best whenever !(best.t < t) is the same rule.
bool world_hit(const rt_scene_desc *sc, const Ray &wr, float t_min, float t_max, Rec &best, Rng &rng)
{
    bool any = false;
    int cur_chain = 0;
    Ray lr = wr;
    for (int i = 0; i < sc->n_entries;) {
        const rt_entry &e = sc->entries[i];
        if ((e.b & 255) == RT_ENTRY_INNER) {
            int chain = e.b >> 8;
            if (chain != cur_chain) lr = chain_down(sc, chain, wr), cur_chain = chain;
            i = aabb_hit(e.lo, e.hi, lr, t_min, t_max) ? i + 1 : e.a;
        } else {
            Rec rec;
            if (prim_hit(sc, e.a, wr, t_min, t_max, rec, rng)) {
                if (!any || !(best.t < rec.t)) best = rec;
                any = true;
            }
            ++i;
        }
    }
    return any;
}

// ---------------------------------------------------------------------------
// textures, src/texture.h and src/perlin.h
// ---------------------------------------------------------------------------
float perlin_noise(V3 p) // src/perlin.h:54-121
{
    float u = p.x - floorf(p.x), v = p.y - floorf(p.y), w = p.z - floorf(p.z);
    int i = int(floorf(p.x)), j = int(floorf(p.y)), k = int(floorf(p.z));
    float uu = (u * u * (3.0f - 2.0f * u)), vv = (v * v * (3.0f - 2.0f * v)), ww = (w * w * (3.0f - 2.0f * w));
    float accum = 0.0f;
    for (int di = 0; di < 2; ++di)
        for (int dj = 0; dj < 2; ++dj)
            for (int dk = 0; dk < 2; ++dk) {
                uint32_t h = 1619u * (uint32_t)(i + di) + 31337u * (uint32_t)(j + dj) + 6971u * (uint32_t)(k + dk) + 1013u;
                int index = (int)h;
                index ^= (index >> 8);
                index &= 0xff;
                V3 c = g_perlin_table ? v3(g_perlin_table[3 * index], g_perlin_table[3 * index + 1], g_perlin_table[3 * index + 2])
                                      : v3(0, 0, 0);
                V3 wv = v3(u - di, v - dj, w - dk);
                accum += (di * uu + (1.0f - di) * (1.0f - uu)) * (dj * vv + (1.0f - dj) * (1.0f - vv)) *
                         (dk * ww + (1.0f - dk) * (1.0f - ww)) * dot(c, wv);
            }
    return accum;
}
float perlin_turb(V3 p, int depth) // src/perlin.h:123-136
{
    float accum = 0.0f, weight = 1.0f;
    for (int i = 0; i < depth; ++i) {
        accum += weight * perlin_noise(p);
        weight *= 0.5f;
        p = p * 2.0f;
    }
    return fabsf(accum);
}

V3 tex_value(const rt_scene_desc *sc, int idx, float u, float v, V3 p)
{
    const rt_texture &t = sc->textures[idx];
    switch (t.kind) {
    case RT_TEX_CONSTANT:
        return v3(t.color[0], t.color[1], t.color[2]);
    case RT_TEX_CHECKER: { // src/texture.h:26-37
        float sines = sin(10.0f * p.x) * sin(10.0f * p.y) * sin(10.0f * p.z);
        return tex_value(sc, sines < 0.0f ? t.a : t.b, u, v, p);
    }
    case RT_TEX_NOISE: // src/texture.h:47-53
        return v3(1.0f, 1.0f, 1.0f) * 0.5f * (1.0f + sinf(t.scale * p.z + 10.0f * perlin_turb(t.scale * p, 7)));
    default: { // image, src/texture.h:63-77
        int i = int(u * (float)t.nx);
        int j = int((1.0f - v) * (float)t.ny - 0.001f);
        if (i < 0) i = 0;
        if (j < 0) j = 0;
        if (i > t.nx - 1) i = t.nx - 1;
        if (j > t.ny - 1) j = t.ny - 1;
        const uint8_t *d = sc->texels + t.offset;
        float r = float(d[3 * i + 3 * t.nx * j]) / 255.0f;
        float g = float(d[3 * i + 3 * t.nx * j + 1]) / 255.0f;
        float b = float(d[3 * i + 3 * t.nx * j + 2]) / 255.0f;
        return v3(r, g, b);
    }
    }
}

// ---------------------------------------------------------------------------
// sampling helpers, src/utils.h, src/vec3.h:173-182.  Constructor arguments are
// evaluated right to left by g++, so the LAST component takes the first draw.
// ---------------------------------------------------------------------------
V3 random_in_unit_sphere(Rng &rng) // src/utils.h:32-42
{
    V3 p;
    do {
        float c = rng.next(), b = rng.next(), a = rng.next();
        p = 2.0f * v3(a, b, c) - v3(1.0f, 1.0f, 1.0f);
    } while (sqlen(p) >= 1.0f);
    return p;
}
V3 random_in_unit_disk(Rng &rng) // src/utils.h:44-55
{
    V3 p;
    do {
        float b = rng.next(), a = rng.next();
        p = 2.0f * v3(a, b, 0.0f) - v3(1.0f, 1.0f, 0.0f);
    } while (dot(p, p) >= 1.0f);
    return p;
}
V3 random_cosine_direction(Rng &rng) // src/vec3.h:173-182, including the factor 2 on x and y
{
    float r1 = rng.next();
    float r2 = rng.next();
    float z = sqrtf(1.0f - r2);
    float phi = 2.0f * ORC_PI * r1;
    float x = cosf(phi) * 2.0f * sqrtf(r2);
    float y = sinf(phi) * 2.0f * sqrtf(r2);
    return v3(x, y, z);
}
float schlick(float cosine, float ref_idx) // src/utils.h:4-9
{
    float r0 = (1 - ref_idx) / (1 + ref_idx);
    r0 = r0 * r0;
    return r0 + (1 - r0) * powf((1 - cosine), 5);
}
V3 reflect(V3 v, V3 n) { return v - 2.0f * dot(v, n) * n; } // src/utils.h:11-14
bool refract(V3 v, V3 n, float ni_over_nt, V3 &refracted)  // src/utils.h:16-30
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

// ---------------------------------------------------------------------------
// important_hitables: hitable_list::pdf_value / random (src/hitable_list.h:67-88)
// over xz_rect (src/plane.h:76-98) and sphere (src/sphere.h:68-103) members
// ---------------------------------------------------------------------------
float light_pdf_value(const rt_scene_desc *sc, int light, V3 o, V3 v)
{
    if (light < 0) return 0.0f; // hitable::pdf_value, src/hitable.h:19
    const rt_prim &pr = sc->prims[light];
    Rec rec;
    Ray r{o, v, 0.0f};
    if (pr.kind == RT_PRIM_XZ_RECT) {
        if (rect_hit(pr.kind, pr.f, r, 0.0001f, FLT_MAX, rec)) {
            float area = (pr.f[1] - pr.f[0]) * (pr.f[3] - pr.f[2]);
            float distance_squared = rec.t * rec.t * sqlen(v);
            float cosine = fabsf(dot(v, rec.n) / len(v));
            return distance_squared / (cosine * area);
        }
        return 0.0f;
    }
    V3 center = v3(pr.f[0], pr.f[1], pr.f[2]);
    float radius = pr.f[3];
    if (sphere_hit(center, radius, true, r, 0.001f, FLT_MAX, rec)) {
        float cos_theta_max = sqrtf(1.0f - radius * radius / sqlen(center - o));
        float solid_angle = 2.0f * ORC_PI * (1.0f - cos_theta_max);
        return 1.0f / solid_angle;
    }
    return 0.0f;
}
V3 light_random(const rt_scene_desc *sc, int light, V3 o, Rng &rng)
{
    if (light < 0) return v3(1, 0, 0); // hitable::random, src/hitable.h:20
    const rt_prim &pr = sc->prims[light];
    if (pr.kind == RT_PRIM_XZ_RECT) {
        float rz = rng.next(), rx = rng.next(); // right-to-left argument evaluation
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
    float phi = 2.0f * ORC_PI * r1;
    float x = cosf(phi) * sqrtf(1.0f - z * z);
    float y = sinf(phi) * sqrtf(1.0f - z * z);
    return uvw.local(v3(x, y, z));
}
float lights_pdf_value(const rt_scene_desc *sc, V3 o, V3 v)
{
    float weight = 1.0f / sc->n_lights;
    float sum = 0.0f;
    for (int i = 0; i < sc->n_lights; ++i) sum += weight * light_pdf_value(sc, sc->lights[i], o, v);
    return sum;
}
V3 lights_random(const rt_scene_desc *sc, V3 o, Rng &rng)
{
    int index = int(rng.next() * sc->n_lights);
    if (index == sc->n_lights) index--;
    return light_random(sc, sc->lights[index], o, rng);
}

// ---------------------------------------------------------------------------
// color(), src/main.cpp:91-127, with the materials of src/material.h inlined
// ---------------------------------------------------------------------------
// orc_kat_bounce (test aid): when set, color() reports its first bounce here -- what was hit, the scattered ray and the
// factor the recursion's result would be multiplied by -- and returns instead of following the scattered ray.  The
// arithmetic of every branch is untouched; the probe only reads values the branch has computed.
struct BounceProbe {
    int hit = 0, material_kind = -1, scattered = 0;
    float t = 0.0f;
    V3 dir{0, 0, 0}, weight{0, 0, 0};
};
thread_local BounceProbe *tl_probe = nullptr;

V3 color(const rt_scene_desc *sc, const Ray &r, int max_depth, int depth, Rng &rng)
{
    ++tl_counters.rays;
    Rec h;
    if (!world_hit(sc, r, 0.001f, FLT_MAX, h, rng)) {
        ++tl_counters.end_miss;
        return v3(0, 0, 0);
    }
    const rt_material &m = sc->materials[h.material];
    V3 emitted = m.kind == RT_MAT_DIFFUSE_LIGHT ? tex_value(sc, m.texture, h.u, h.v, h.p) : v3(0, 0, 0);
    if (tl_probe) tl_probe->hit = 1, tl_probe->material_kind = m.kind, tl_probe->t = h.t, tl_probe->weight = emitted;
    if (!(depth < max_depth) || m.kind == RT_MAT_DIFFUSE_LIGHT) {
        if (m.kind == RT_MAT_DIFFUSE_LIGHT) ++tl_counters.end_light; else ++tl_counters.end_depth;
        return emitted;
    }
    ++tl_counters.scatters;
    switch (m.kind) {
    case RT_MAT_LAMBERTIAN: { // src/material.h:24-44 + src/main.cpp:105-116 + src/pdf.h
        V3 albedo = tex_value(sc, m.texture, h.u, h.v, h.p);
        Onb uvw;
        uvw.build_from_w(h.n);
        V3 dir;
        if (rng.next() < 0.5f) dir = unit(lights_random(sc, h.p, rng));
        else dir = unit(uvw.local(random_cosine_direction(rng)));
        Ray scattered{h.p, dir, r.time};
        float cosine = dot(unit(dir), uvw.w); // cosine_pdf::value, src/pdf.h:13-24
        float cos_val = cosine > 0.0f ? cosine / ORC_PI : 0.0f;
        float pdf_val = 0.5f * lights_pdf_value(sc, h.p, dir) + 0.5f * cos_val;
        float sc_cos = dot(h.n, dir); // lambertian::scattering_pdf, src/material.h:28-33
        if (sc_cos < 0.0f) sc_cos = 0.0f;
        float spdf = sc_cos / ORC_PI;
        if (tl_probe) {
            tl_probe->scattered = 1, tl_probe->dir = dir, tl_probe->weight = albedo * spdf / pdf_val;
            return v3(0, 0, 0);
        }
        return emitted + albedo * spdf * color(sc, scattered, max_depth, depth + 1, rng) / pdf_val;
    }
    case RT_MAT_METAL: { // src/material.h:65-78
        V3 reflected = reflect(unit(r.d), h.n);
        // (per-path stream: no draws at fuzz 0, include/rt_rng.h; the reference's stream draws them as the reference does)
        V3 fuzz = (rng.mode == ORC_RNG_COUNTER && m.p0 == 0.0f) ? v3(0, 0, 0) : m.p0 * random_in_unit_sphere(rng);
        Ray spec{h.p, reflected + fuzz, r.time};
        V3 albedo = tex_value(sc, m.texture, h.u, h.v, h.p);
        if (tl_probe) {
            tl_probe->scattered = 1, tl_probe->dir = spec.d, tl_probe->weight = albedo;
            return v3(0, 0, 0);
        }
        return albedo * color(sc, spec, max_depth, depth + 1, rng);
    }
    case RT_MAT_DIELECTRIC: { // src/material.h:92-145
        V3 outward_normal;
        V3 reflected = reflect(unit(r.d), h.n);
        float ni_over_nt, reflect_prob, cosine;
        float ref_idx = m.p0;
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
        Ray spec{h.p, rng.next() < reflect_prob ? reflected : refracted, r.time};
        if (tl_probe) {
            tl_probe->scattered = 1, tl_probe->dir = spec.d, tl_probe->weight = v3(1.0f, 1.0f, 1.0f);
            return v3(0, 0, 0);
        }
        return v3(1.0f, 1.0f, 1.0f) * color(sc, spec, max_depth, depth + 1, rng);
    }
    default: { // isotropic, src/material.h:187-196: time 0, flagged specular
        Ray spec{h.p, random_in_unit_sphere(rng), 0.0f};
        V3 albedo = tex_value(sc, m.texture, h.u, h.v, h.p);
        if (tl_probe) {
            tl_probe->scattered = 1, tl_probe->dir = spec.d, tl_probe->weight = albedo;
            return v3(0, 0, 0);
        }
        return albedo * color(sc, spec, max_depth, depth + 1, rng);
    }
    }
}

// camera::get_ray, src/camera.h:54-68
Ray get_ray(const rt_camera &c, float s, float t, Rng &rng)
{
    V3 cu = v3(c.u[0], c.u[1], c.u[2]), cv = v3(c.v[0], c.v[1], c.v[2]);
    V3 origin = v3(c.origin[0], c.origin[1], c.origin[2]);
    V3 llc = v3(c.lower_left_corner[0], c.lower_left_corner[1], c.lower_left_corner[2]);
    V3 hor = v3(c.horizontal[0], c.horizontal[1], c.horizontal[2]), ver = v3(c.vertical[0], c.vertical[1], c.vertical[2]);
    // per-path stream (include/rt_rng.h): no lens draws when they cannot matter; the reference's own
    // stream (MINSTD mode) always draws them, as src/camera.h:56 does
    V3 disk = (rng.mode == ORC_RNG_COUNTER && c.lens_radius == 0.0f) ? v3(0, 0, 0) : random_in_unit_disk(rng);
    V3 rd = c.lens_radius * disk;
    V3 offset = cu * rd.x + cv * rd.y;
    V3 oo = origin + offset;
    float time = c.time0 + rng.next() * (c.time1 - c.time0);
    return Ray{oo, (llc + s * hor + t * ver) - oo, time};
}

inline V3 de_nan(V3 c) // src/vec3.h:184-191
{
    if (!(c.x == c.x)) c.x = 0.0f;
    if (!(c.y == c.y)) c.y = 0.0f;
    if (!(c.z == c.z)) c.z = 0.0f;
    return c;
}

// one pixel of compute_one_sample_task::run, src/main.cpp:227-242
inline void shade_pixel(const rt_scene_desc *sc, int nx, int ny, int i, int j, int nss, int max_depth, Rng &rng,
                        int sample, uint32_t seed, float *px)
{
    V3 col = v3(0, 0, 0);
    for (int s = 0; s < nss; ++s) {
        if (rng.mode == ORC_RNG_COUNTER) {
            uint32_t pixel = rt_pixel_word((uint32_t)(ny - 1 - j), (uint32_t)i);
            rng.key = rt_path_key(seed, (uint32_t)(sample * nss + s));
            rng.pixel = pixel, rng.n = 0;
        }
        float u = float(i + rng.next()) / float(nx);
        float v = float(j + rng.next()) / float(ny);
        Ray r = get_ray(sc->camera, u, v, rng);
        ++tl_counters.paths;
        col = col + de_nan(color(sc, r, max_depth, 0, rng));
    }
    col = col / (float)nss;
    px[0] = 1.0f, px[1] = col.z, px[2] = col.y, px[3] = col.x; // A B G R
}

void merge_counters()
{
    static std::atomic_flag lock = ATOMIC_FLAG_INIT;
    while (lock.test_and_set()) {}
    Counters &g = g_counters, &t = tl_counters;
    g.rays += t.rays, g.slab_tests += t.slab_tests, g.rect_tests += t.rect_tests, g.sphere_tests += t.sphere_tests;
    g.scatters += t.scatters, g.paths += t.paths, g.end_miss += t.end_miss, g.end_light += t.end_light, g.end_depth += t.end_depth;
    t = Counters();
    lock.clear();
}

} // namespace

// ===========================================================================
// C API (ctypes)
// ===========================================================================
extern "C" {

void orc_set_perlin_table(const float *table_256x3) { g_perlin_table = table_256x3; }

// include/rt_rng.h as compiled: draws 0..n-1 of the path (seed, pixel, sample) -- tests/test_rng.py pins the header's
// arithmetic against a numpy restatement and checks the properties the kernels rely on
void orc_rng_draws(uint32_t seed, uint32_t pixel, uint32_t sample, int n, float *out)
{
    const uint32_t key = rt_path_key(seed, sample);
    for (int k = 0; k < n; ++k) out[k] = rt_draw(key, pixel, (uint32_t)k);
}
uint32_t orc_rng_key(uint32_t seed, uint32_t sample) { return rt_path_key(seed, sample); }
uint32_t orc_rng_mix_w(uint32_t x, uint32_t w) { return rt_mix32_w(x, w); }

// Renders samples [sample_begin, sample_end).  out_sum: ny*nx*4 floats, per pixel
// [A,B,G,R] summed over the samples in sample order (what src/main.cpp:720-726
// adds up).  out_per_sample: NULL or (sample_end-sample_begin)*ny*nx*4 floats, the
// reference's per-sample buffers.  rng_mode ORC_RNG_MINSTD: `state` is the minstd
// engine state to continue from (single thread, rows top to bottom, exactly the
// order of a `-t 1` run); ORC_RNG_COUNTER: `state` is the seed, any thread count.
// Returns the minstd state afterwards (MINSTD mode) or 0.
uint32_t orc_render(const rt_scene_desc *sc, int nx, int ny, int sample_begin, int sample_end, int nss, int max_depth,
                    int rng_mode, uint32_t state, int threads, float *out_sum, float *out_per_sample)
{
    size_t npx = (size_t)nx * ny;
    for (size_t k = 0; k < npx * 4; ++k) out_sum[k] = 0.0f;
    if (rng_mode == ORC_RNG_MINSTD) {
        std::default_random_engine engine(state);
        std::uniform_real_distribution<float> dist(0.0f, 1.0f);
        Rng rng;
        rng.mode = ORC_RNG_MINSTD, rng.engine = &engine, rng.dist = &dist;
        std::vector<float> one(npx * 4);
        for (int s = sample_begin; s < sample_end; ++s) {
            float *buf = out_per_sample ? out_per_sample + (size_t)(s - sample_begin) * npx * 4 : one.data();
            for (int j = ny - 1; j >= 0; --j)
                for (int i = 0; i < nx; ++i)
                    shade_pixel(sc, nx, ny, i, j, nss, max_depth, rng, s, 0, buf + ((size_t)(ny - 1 - j) * nx + i) * 4);
            for (size_t k = 0; k < npx * 4; ++k) out_sum[k] += buf[k];
        }
        merge_counters();
        std::ostringstream os; // the engine prints its state: one integer for minstd
        os << engine;
        uint32_t st = (uint32_t)std::stoul(os.str());
        return st;
    }
    if (threads < 1) threads = 1;
    std::atomic<int> next_row(0);
    auto work = [&]() {
        Rng rng;
        rng.mode = ORC_RNG_COUNTER;
        std::vector<float> px(4);
        for (;;) {
            int row = next_row.fetch_add(1);
            if (row >= ny) break;
            int j = ny - 1 - row;
            for (int i = 0; i < nx; ++i) {
                float *acc = out_sum + ((size_t)row * nx + i) * 4;
                for (int s = sample_begin; s < sample_end; ++s) {
                    shade_pixel(sc, nx, ny, i, j, nss, max_depth, rng, s, state, px.data());
                    for (int c = 0; c < 4; ++c) acc[c] += px[c];
                    if (out_per_sample) memcpy(out_per_sample + ((size_t)(s - sample_begin) * npx + (size_t)row * nx + i) * 4, px.data(), 16);
                }
            }
        }
        merge_counters();
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
    return 0;
}

// resolve, src/main.cpp:708-743; accum holds sums, A = number of samples
void orc_resolve(const float *accum, int nx, int ny, uint32_t *out)
{
    for (size_t p = 0; p < (size_t)nx * ny; ++p) {
        float inv = 1.0f / (float)(int)accum[4 * p];
        float fr = accum[4 * p + 3] * inv, fg = accum[4 * p + 2] * inv, fb = accum[4 * p + 1] * inv;
        float c[3] = {sqrtf(fr), sqrtf(fg), sqrtf(fb)};
        for (int k = 0; k < 3; ++k) {
            if (c[k] < 0.0f) c[k] = 0.0f;
            if (c[k] > 1.0f) c[k] = 1.0f;
        }
        uint32_t ir = (uint32_t)(c[0] * 255.99f), ig = (uint32_t)(c[1] * 255.99f), ib = (uint32_t)(c[2] * 255.99f);
        out[p] = 0xff000000u | (ib << 16) | (ig << 8) | ir;
    }
}

// camera::get_ray with explicit random inputs (mirrors rt_camera_rays)
void orc_camera_rays(const rt_scene_desc *sc, int n, const float *st, const float *lens, const float *time01, float *rays)
{
    const rt_camera &c = sc->camera;
    V3 cu = v3(c.u[0], c.u[1], c.u[2]), cv = v3(c.v[0], c.v[1], c.v[2]);
    V3 origin = v3(c.origin[0], c.origin[1], c.origin[2]);
    V3 llc = v3(c.lower_left_corner[0], c.lower_left_corner[1], c.lower_left_corner[2]);
    V3 hor = v3(c.horizontal[0], c.horizontal[1], c.horizontal[2]), ver = v3(c.vertical[0], c.vertical[1], c.vertical[2]);
    for (int k = 0; k < n; ++k) {
        V3 disk = lens ? v3(lens[2 * k], lens[2 * k + 1], 0.0f) : v3(0, 0, 0);
        V3 rd = c.lens_radius * disk;
        V3 offset = cu * rd.x + cv * rd.y;
        V3 oo = origin + offset;
        float time = c.time0 + (time01 ? time01[k] : 0.0f) * (c.time1 - c.time0);
        V3 d = (llc + st[2 * k] * hor + st[2 * k + 1] * ver) - oo;
        float *o = rays + 7 * k;
        o[0] = oo.x, o[1] = oo.y, o[2] = oo.z, o[3] = d.x, o[4] = d.y, o[5] = d.z, o[6] = time;
    }
}

// world->hit() for n rays (mirrors rt_trace_rays); medium draws come from stream (seed, ray index, 0)
void orc_trace_rays(const rt_scene_desc *sc, int n, const float *rays, float t_min, float t_max, uint32_t seed, rt_hit *out)
{
    for (int k = 0; k < n; ++k) {
        const float *f = rays + 7 * k;
        Ray r{v3(f[0], f[1], f[2]), v3(f[3], f[4], f[5]), f[6]};
        Rng rng;
        rng.mode = ORC_RNG_COUNTER, rng.key = rt_path_key(seed, 0), rng.pixel = (uint32_t)k, rng.n = 0;
        Rec h;
        rt_hit &o = out[k];
        memset(&o, 0, sizeof o);
        if (world_hit(sc, r, t_min, t_max, h, rng)) {
            o.prim_id = sc->prims[h.prim].user_id, o.material = h.material, o.t = h.t, o.u = h.u, o.v = h.v;
            o.p[0] = h.p.x, o.p[1] = h.p.y, o.p[2] = h.p.z, o.normal[0] = h.n.x, o.normal[1] = h.n.y, o.normal[2] = h.n.z;
        } else {
            o.prim_id = -1, o.material = -1;
        }
    }
    tl_counters = Counters();
}

void orc_counters_reset(void) { g_counters = Counters(); }
// rays, slab, rect, sphere, scatters, paths, end_miss, end_light, end_depth
void orc_counters_get(uint64_t out[9])
{
    const Counters &g = g_counters;
    uint64_t v[9] = {g.rays, g.slab_tests, g.rect_tests, g.sphere_tests, g.scatters, g.paths, g.end_miss, g.end_light, g.end_depth};
    memcpy(out, v, sizeof v);
}

// ---- function-level entry points for the known-answer tests -----------------
int orc_kat_aabb_hit(const float *o, const float *d, const float *lo, const float *hi, float t0, float t1)
{
    Ray r{v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2]), 0.0f};
    return aabb_hit(lo, hi, r, t0, t1);
}
// out: t u v p(3) n(3)
int orc_kat_sphere_hit(const float *o, const float *d, const float *c, float radius, float t0, float t1, float *out)
{
    Ray r{v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2]), 0.0f};
    Rec h;
    if (!sphere_hit(v3(c[0], c[1], c[2]), radius, true, r, t0, t1, h)) return 0;
    float v[9] = {h.t, h.u, h.v, h.p.x, h.p.y, h.p.z, h.n.x, h.n.y, h.n.z};
    memcpy(out, v, sizeof v);
    return 1;
}
int orc_kat_rect_hit(int axis, const float *o, const float *d, const float *f5, float t0, float t1, float *out)
{
    Ray r{v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2]), 0.0f};
    Rec h;
    int kind = axis == 0 ? RT_PRIM_XY_RECT : axis == 1 ? RT_PRIM_XZ_RECT : RT_PRIM_YZ_RECT;
    if (!rect_hit(kind, f5, r, t0, t1, h)) return 0;
    float v[9] = {h.t, h.u, h.v, h.p.x, h.p.y, h.p.z, h.n.x, h.n.y, h.n.z};
    memcpy(out, v, sizeof v);
    return 1;
}
// out: reflect(3) ok refracted(3) schlick
void orc_kat_optics(const float *v, const float *n, float ni_over_nt, float cosine, float ref_idx, float *out)
{
    V3 a = v3(v[0], v[1], v[2]), nn = v3(n[0], n[1], n[2]);
    V3 rf = reflect(a, nn), rr = v3(0, 0, 0);
    bool ok = refract(a, nn, ni_over_nt, rr);
    float o[8] = {rf.x, rf.y, rf.z, ok ? 1.f : 0.f, ok ? rr.x : 0.f, ok ? rr.y : 0.f, ok ? rr.z : 0.f, schlick(cosine, ref_idx)};
    memcpy(out, o, sizeof o);
}
// out: u(3) v(3) w(3) unit(3) su sv
void orc_kat_onb_uv(const float *n, float *out)
{
    Onb b;
    b.build_from_w(v3(n[0], n[1], n[2]));
    V3 p = unit(v3(n[0], n[1], n[2]));
    float su, sv;
    sphere_uv(p, su, sv);
    float o[14] = {b.u.x, b.u.y, b.u.z, b.v.x, b.v.y, b.v.z, b.w.x, b.w.y, b.w.z, p.x, p.y, p.z, su, sv};
    memcpy(out, o, sizeof o);
}
// kind 0: xz_rect f = x0 x1 z0 z1 y; kind 1: sphere f = cx cy cz r
float orc_kat_pdf_value(int kind, const float *o, const float *v, const float *f5)
{
    rt_scene_desc sc;
    memset(&sc, 0, sizeof sc);
    rt_prim pr;
    memset(&pr, 0, sizeof pr);
    pr.kind = kind == 0 ? RT_PRIM_XZ_RECT : RT_PRIM_SPHERE;
    memcpy(pr.f, f5, 5 * sizeof(float));
    sc.prims = &pr, sc.n_prims = 1;
    return light_pdf_value(&sc, 0, v3(o[0], o[1], o[2]), v3(v[0], v[1], v[2]));
}
float orc_kat_cosine_pdf_value(const float *normal, const float *dir)
{
    Onb uvw;
    uvw.build_from_w(v3(normal[0], normal[1], normal[2]));
    float cosine = dot(unit(v3(dir[0], dir[1], dir[2])), uvw.w);
    return cosine > 0.0f ? cosine / ORC_PI : 0.0f;
}
float orc_kat_perlin_noise(const float *p) { return perlin_noise(v3(p[0], p[1], p[2])); }
float orc_kat_perlin_turb(const float *p) { return perlin_turb(v3(p[0], p[1], p[2]), 7); }
// value of texture `idx` of a caller-made description
void orc_kat_texture(const rt_scene_desc *sc, int idx, float u, float v, const float *p, float *rgb)
{
    V3 c = tex_value(sc, idx, u, v, v3(p[0], p[1], p[2]));
    rgb[0] = c.x, rgb[1] = c.y, rgb[2] = c.z;
}

// The samplers on the per-path stream (key, pixel), each from draw 0 -- the layout of rt_debug_eval's RT_EVAL_SAMPLERS:
// out: lights_random(o)[3], random_cosine_direction[3], random_in_unit_sphere[3], random_in_unit_disk x y
void orc_kat_samplers(const rt_scene_desc *sc, uint32_t key, uint32_t pixel, const float *o, float *out11)
{
    auto fresh = [&]() {
        Rng r;
        r.mode = ORC_RNG_COUNTER, r.key = key, r.pixel = pixel, r.n = 0;
        return r;
    };
    Rng r0 = fresh(), r1 = fresh(), r2 = fresh(), r3 = fresh();
    V3 a = lights_random(sc, v3(o[0], o[1], o[2]), r0), b = random_cosine_direction(r1), c = random_in_unit_sphere(r2), d = random_in_unit_disk(r3);
    float v[11] = {a.x, a.y, a.z, b.x, b.y, b.z, c.x, c.y, c.z, d.x, d.y};
    memcpy(out11, v, sizeof v);
}
// world->hit() and ONE bounce of color() (src/main.cpp:91-127) at depth 0 of 50 on the stream (key, pixel) -- the layout of
// RT_EVAL_BOUNCE: hit?, material kind, scattered?, direction[3], weight[3] (the factor the rest of the path is multiplied by;
// the emitted radiance when the path ends here), t
void orc_kat_bounce(const rt_scene_desc *sc, const float *ray7, uint32_t key, uint32_t pixel, float *out10)
{
    Rng rng;
    rng.mode = ORC_RNG_COUNTER, rng.key = key, rng.pixel = pixel, rng.n = 0;
    BounceProbe probe;
    tl_probe = &probe;
    Ray r{v3(ray7[0], ray7[1], ray7[2]), v3(ray7[3], ray7[4], ray7[5]), ray7[6]};
    (void)color(sc, r, 50, 0, rng);
    tl_probe = nullptr;
    float v[10] = {(float)probe.hit, (float)probe.material_kind, (float)probe.scattered, probe.dir.x, probe.dir.y, probe.dir.z,
                   probe.weight.x, probe.weight.y, probe.weight.z, probe.t};
    memcpy(out10, v, sizeof v);
}

} // extern "C"
