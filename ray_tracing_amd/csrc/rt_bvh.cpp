// rt_bvh.cpp -- binned-SAH BVH2 over the world-space boxes of a scene's primitives (rt_bvh.h).
#include "rt_bvh.h"

#include "rt_box.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace {

struct Box {
    float lo[3], hi[3];
    void clear() { lo[0] = lo[1] = lo[2] = INFINITY, hi[0] = hi[1] = hi[2] = -INFINITY; }
    void add(const float p[3])
    {
        for (int k = 0; k < 3; ++k) lo[k] = std::min(lo[k], p[k]), hi[k] = std::max(hi[k], p[k]);
    }
    void add(const Box &b)
    {
        for (int k = 0; k < 3; ++k) lo[k] = std::min(lo[k], b.lo[k]), hi[k] = std::max(hi[k], b.hi[k]);
    }
    float area() const
    {
        float e[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
        if (e[0] < 0) return 0.0f;
        return e[0] * e[1] + e[1] * e[2] + e[2] * e[0];
    }
    bool finite() const
    {
        for (int k = 0; k < 3; ++k)
            if (!std::isfinite(lo[k]) || !std::isfinite(hi[k])) return false;
        return true;
    }
};

struct Item {
    Box b;
    float c[3];
    int32_t prim;
};

// local box of a primitive (src/sphere.h:61-66, 185-196; src/plane.h:31-35 and siblings; src/box.h:25-29)
bool local_box(const rt_scene_desc *d, const rt_prim &p, Box &b)
{
    b.clear();
    switch (p.kind) {
    case RT_PRIM_SPHERE: {
        float r = std::fabs(p.f[3]);
        for (int k = 0; k < 3; ++k) b.lo[k] = p.f[k] - r, b.hi[k] = p.f[k] + r;
        return true;
    }
    case RT_PRIM_MOVING_SPHERE: {
        float r = std::fabs(p.f[6]);
        for (int k = 0; k < 3; ++k)
            b.lo[k] = std::min(p.f[k], p.f[3 + k]) - r, b.hi[k] = std::max(p.f[k], p.f[3 + k]) + r;
        return true;
    }
    case RT_PRIM_XY_RECT:
    case RT_PRIM_XZ_RECT:
    case RT_PRIM_YZ_RECT: {
        const int ka = p.kind == RT_PRIM_YZ_RECT ? 1 : 0, kb = p.kind == RT_PRIM_XY_RECT ? 1 : 2, kk = 3 - ka - kb;
        b.lo[ka] = p.f[0], b.hi[ka] = p.f[1], b.lo[kb] = p.f[2], b.hi[kb] = p.f[3];
        b.lo[kk] = p.f[4] - 0.0001f, b.hi[kk] = p.f[4] + 0.0001f;
        return true;
    }
    case RT_PRIM_GROUP:
        for (int k = 0; k < p.count; ++k) {
            Box c;
            if (!local_box(d, d->prims[p.first + k], c)) return false;
            b.add(c);
        }
        return true;
    default: return false;
    }
}

// a point up an instance chain (innermost wrapper first), src/transforms.h:21, 114-122
void chain_up_point(const rt_scene_desc *d, int chain, float q[3])
{
    const rt_chain &c = d->chains[chain];
    for (int i = c.count - 1; i >= 0; --i) {
        const rt_xform &x = d->xforms[c.first + i];
        if (x.kind == RT_XFORM_TRANSLATE) {
            q[0] += x.a, q[1] += x.b, q[2] += x.c;
        } else {
            const float s = x.a, co = x.b, px = q[0], pz = q[2];
            q[0] = co * px + s * pz;
            q[2] = -s * px + co * pz;
        }
    }
}

// box of a primitive in WORLD space: the eight corners go up its instance chain (src/transforms.h:30-52, 69-100)
bool world_box(const rt_scene_desc *d, const rt_prim &p, Box &out)
{
    const rt_prim &shape = p.kind == RT_PRIM_MEDIUM ? d->prims[p.first] : p;
    Box l;
    if (!local_box(d, shape, l)) return false;
    out.clear();
    for (int corner = 0; corner < 8; ++corner) {
        float q[3] = {corner & 1 ? l.hi[0] : l.lo[0], corner & 2 ? l.hi[1] : l.lo[1], corner & 4 ? l.hi[2] : l.lo[2]};
        chain_up_point(d, shape.chain, q);
        out.add(q);
    }
    // float rounding of the rotation: keep the box conservative
    for (int k = 0; k < 3; ++k) {
        const float pad = 1e-5f * (std::fabs(out.lo[k]) + std::fabs(out.hi[k])) + 1e-6f;
        out.lo[k] -= pad, out.hi[k] += pad;
    }
    return out.finite();
}

// the tree is built with float boxes and references to primitives; rt_bvh_build packs it afterwards
struct Node32 {
    float lo0[3], hi0[3], lo1[3], hi1[3];
    int32_t ref0, ref1;
};

struct Builder {
    std::vector<Item> items;
    std::vector<Node32> *nodes;
    int max_depth = 0;
    int limit = RT_BVH_MAX_DEPTH; // inner levels the tree may have
    bool ok = true;

    static int log2_ceil(size_t n)
    {
        int l = 0;
        while (((size_t)1 << l) < n) ++l;
        return l;
    }

    // builds the subtree over items [a, b), b - a >= 2, into a new node; returns its index
    int32_t build(size_t a, size_t b, int depth)
    {
        max_depth = std::max(max_depth, depth);
        if (depth > limit) return ok = false, 0;
        const int32_t me = (int32_t)nodes->size();
        nodes->push_back(Node32{});
        const size_t n = b - a;
        Box cb;
        cb.clear();
        for (size_t i = a; i < b; ++i) cb.add(items[i].c);
        size_t mid = a + n / 2;
        int axis = 0;
        for (int k = 1; k < 3; ++k)
            if (cb.hi[k] - cb.lo[k] > cb.hi[axis] - cb.lo[axis]) axis = k;
        // the stack bound: once the remaining depth only just fits a balanced tree, split at the median
        const bool force_median = depth + log2_ceil(n) >= limit - 1;
        bool split_done = false;
        if (!force_median && n > 2) {
            constexpr int NB = 16;
            float best_cost = INFINITY;
            int best_axis = -1, best_bin = -1;
            for (int k = 0; k < 3; ++k) {
                const float ext = cb.hi[k] - cb.lo[k];
                if (!(ext > 0.0f)) continue;
                Box bb[NB];
                size_t cnt[NB] = {0};
                for (Box &x : bb) x.clear();
                const float scale = NB / ext;
                for (size_t i = a; i < b; ++i) {
                    int bin = std::min(NB - 1, std::max(0, (int)((items[i].c[k] - cb.lo[k]) * scale)));
                    bb[bin].add(items[i].b);
                    ++cnt[bin];
                }
                float right_area[NB];
                Box acc;
                acc.clear();
                size_t right_cnt[NB];
                size_t rc = 0;
                for (int i = NB - 1; i > 0; --i) {
                    acc.add(bb[i]);
                    rc += cnt[i];
                    right_area[i] = acc.area(), right_cnt[i] = rc;
                }
                acc.clear();
                size_t lc = 0;
                for (int i = 0; i < NB - 1; ++i) {
                    acc.add(bb[i]);
                    lc += cnt[i];
                    if (lc == 0 || right_cnt[i + 1] == 0) continue;
                    const float cost = acc.area() * (float)lc + right_area[i + 1] * (float)right_cnt[i + 1];
                    if (cost < best_cost) best_cost = cost, best_axis = k, best_bin = i;
                }
            }
            // One more candidate: the LARGEST item alone against all the others.  Bins of centroids cannot isolate an item whose
            // centre lies among the others' -- a scene-wide fog, a ground plane of one piece -- so it sank to the bottom of the
            // tree and every box above it was the whole scene (Next-Week final: every ray opened the eleven ancestors of the fog).
            if (depth + 1 + log2_ceil(n - 1) < limit - 1) {
                size_t big = a;
                for (size_t i = a + 1; i < b; ++i)
                    if (items[i].b.area() > items[big].b.area()) big = i;
                Box rest;
                rest.clear();
                for (size_t i = a; i < b; ++i)
                    if (i != big) rest.add(items[i].b);
                const float cost = items[big].b.area() + rest.area() * (float)(n - 1);
                if (cost < best_cost) {
                    std::swap(items[a], items[big]);
                    mid = a + 1;
                    split_done = true;
                    best_axis = -1;
                }
            }
            if (best_axis >= 0) {
                const float ext = cb.hi[best_axis] - cb.lo[best_axis], scale = 16 / ext, lo = cb.lo[best_axis];
                const int bax = best_axis, bbin = best_bin;
                auto it = std::partition(items.begin() + a, items.begin() + b, [=](const Item &x) {
                    int bin = std::min(15, std::max(0, (int)((x.c[bax] - lo) * scale)));
                    return bin <= bbin;
                });
                mid = (size_t)(it - items.begin());
                split_done = mid > a && mid < b;
            }
        }
        if (!split_done) {
            mid = a + n / 2;
            std::nth_element(items.begin() + a, items.begin() + mid, items.begin() + b,
                             [axis](const Item &x, const Item &y) { return x.c[axis] < y.c[axis]; });
        }
        Box b0, b1;
        b0.clear(), b1.clear();
        for (size_t i = a; i < mid; ++i) b0.add(items[i].b);
        for (size_t i = mid; i < b; ++i) b1.add(items[i].b);
        const int32_t r0 = mid - a == 1 ? ~items[a].prim : build(a, mid, depth + 1);
        const int32_t r1 = b - mid == 1 ? ~items[mid].prim : build(mid, b, depth + 1);
        Node32 &nd = (*nodes)[me];
        for (int k = 0; k < 3; ++k) nd.lo0[k] = b0.lo[k], nd.hi0[k] = b0.hi[k], nd.lo1[k] = b1.lo[k], nd.hi1[k] = b1.hi[k];
        nd.ref0 = r0, nd.ref1 = r1;
        return me;
    }
};

// float -> IEEE half, rounded towards -inf (up = false) or +inf (up = true)
uint16_t to_half(float x, bool up)
{
    if (std::isnan(x)) return 0x7e00;
    const bool neg = std::signbit(x);
    const double a = std::fabs((double)x);
    const bool away = neg ? !up : up; // round the magnitude away from zero?
    uint16_t h;
    if (a == 0.0) {
        h = 0;
    } else if (a > 65504.0) {
        h = away ? 0x7c00 : 0x7bff; // inf, or the largest finite half
    } else {
        int e;
        const double m = std::frexp(a, &e); // a = m * 2^e, m in [0.5, 1)
        int he = e - 1 + 15;                // biased exponent of the normal form 1.f * 2^(e-1)
        double frac;                        // mantissa in units of the last place
        if (he <= 0) {                      // subnormal half: units of 2^-24
            frac = a * 16777216.0;
            he = 0;
        } else {
            frac = (m * 2.0 - 1.0) * 1024.0;
        }
        double fl = std::floor(frac);
        if (away && fl != frac) fl += 1.0;
        uint32_t bits = ((uint32_t)he << 10) + (uint32_t)fl; // a carry out of the mantissa moves into the exponent
        if (bits >= 0x7c00u) bits = away ? 0x7c00u : 0x7bffu;
        h = (uint16_t)bits;
    }
    return (uint16_t)(h | (neg ? 0x8000u : 0u));
}

} // namespace

bool rt_bvh_build(const rt_scene_desc *d, std::vector<rt_bvh_node> &nodes, std::vector<rt_leaf> &leaves, std::vector<rt_prim> &prims,
                  int *depth)
{
    nodes.clear();
    leaves.clear();
    prims.assign(d->prims, d->prims + d->n_prims);
    if (d->n_chains > (1 << 24)) return false;
    std::vector<Node32> tree;
    Builder B;
    B.nodes = &tree;
    std::vector<char> seen((size_t)d->n_prims, 0);
    auto add_item = [&](int32_t pi) -> bool {
        // A one-primitive bvh_node holds its primitive twice (src/bvh.h:98-101) and the reference visits it twice.  For a
        // surface that changes nothing; a constant_medium draws a random number per visit (src/volume.h:43), so it gets
        // two chances to scatter.  The item is stored once, with the number of visits (rt_prim.reserved) for walk_leaf.
        if (seen[pi]) {
            ++prims[pi].reserved;
            return true;
        }
        seen[pi] = 1;
        prims[pi].reserved = 1;
        Item it;
        if (!world_box(d, d->prims[pi], it.b)) return false; // from the ORIGINAL primitive and its chain
        for (int k = 0; k < 3; ++k) it.c[k] = 0.5f * (it.b.lo[k] + it.b.hi[k]);
        it.prim = pi;
        B.items.push_back(it);
        rt_prim &p = prims[pi];
        if (p.chain != 0 && !(p.flags & RT_PRIM_NEEDS_UV)) { // a rigid chain moves a sphere's centre and nothing else
            if (p.kind == RT_PRIM_SPHERE) {
                chain_up_point(d, p.chain, p.f);
                p.chain = 0;
            } else if (p.kind == RT_PRIM_MOVING_SPHERE) {
                chain_up_point(d, p.chain, p.f);
                chain_up_point(d, p.chain, p.f + 3);
                p.chain = 0;
            }
        }
        return true;
    };
    // A list that is a box (rt_box.h) is ONE item and one slab test for the walk, also as the boundary of a
    // constant_medium: the library's copy of the list gets p0 / p1 and its members in the order [2 * axis + side].
    auto mark_box = [&](int32_t gi) -> bool {
        float lo[3], hi[3];
        int face[6];
        if (!rt_as_box(d, d->prims[gi], lo, hi, face)) return false;
        rt_prim &g = prims[gi];
        if (g.flags & RT_PRIM_IS_BOX) return true;
        g.flags |= RT_PRIM_IS_BOX;
        for (int k = 0; k < 3; ++k) g.f[k] = lo[k], g.f[3 + k] = hi[k];
        for (int k = 0; k < 6; ++k) prims[g.first + k] = d->prims[face[k]];
        return true;
    };
    for (int e = 0; e < d->n_entries; ++e) {
        if ((d->entries[e].b & 255) != RT_ENTRY_LEAF) continue;
        const int32_t pi = d->entries[e].a;
        const rt_prim &p = d->prims[pi];
        if (p.kind == RT_PRIM_MEDIUM && d->prims[p.first].kind == RT_PRIM_GROUP) mark_box(p.first);
        if (p.kind == RT_PRIM_GROUP && !mark_box(pi)) {
            for (int k = 0; k < p.count; ++k)
                if (!add_item(p.first + k)) return false;
        } else if (!add_item(pi)) {
            return false;
        }
    }
    if (B.items.empty()) return false;
    tree.reserve(B.items.size());
    if (B.items.size() == 1) { // a root with one child
        Node32 nd{};
        for (int k = 0; k < 3; ++k) nd.lo0[k] = B.items[0].b.lo[k], nd.hi0[k] = B.items[0].b.hi[k], nd.lo1[k] = 0.0f, nd.hi1[k] = 0.0f;
        nd.ref0 = ~B.items[0].prim, nd.ref1 = RT_BVH_NONE;
        tree.push_back(nd);
        *depth = 1;
    } else {
        const std::vector<Item> items = B.items; // build() permutes them
        B.build(0, B.items.size(), 1);
        if (!B.ok) { // too many primitives for the preferred bound: the deeper stack (five waves per SIMD)
            tree.clear();
            B.items = items, B.ok = true, B.max_depth = 0, B.limit = RT_BVH_DEPTH_LIMIT;
            B.build(0, B.items.size(), 1);
        }
        *depth = B.max_depth;
        if (!B.ok) return false;
    }
    // pack: leaves in the order the tree names them, boxes as halves rounded outwards
    auto leaf_of = [&](int32_t ref) -> int32_t {
        if (ref >= 0 || ref == RT_BVH_NONE) return ref;
        const int32_t pi = ~ref;
        const rt_prim &p = prims[pi];
        const rt_prim &shape = p.kind == RT_PRIM_MEDIUM ? prims[p.first] : p;
        rt_leaf lf{};
        for (int k = 0; k < 5; ++k) lf.f[k] = p.kind == RT_PRIM_MEDIUM ? 0.0f : p.f[k];
        lf.kind_chain = (uint32_t)p.kind | ((uint32_t)shape.chain << 8);
        lf.prim = pi;
        if ((p.kind <= RT_PRIM_MOVING_SPHERE || p.kind == RT_PRIM_XY_RECT || p.kind == RT_PRIM_XZ_RECT || p.kind == RT_PRIM_YZ_RECT) &&
            p.material >= 0 && p.material < d->n_materials) {
            const int32_t tx = d->materials[p.material].texture;
            if (tx >= 0 && tx < d->n_textures && d->textures[tx].kind == RT_TEX_NOISE) lf.prim |= RT_LEAF_PRIM_NOISE;
        }
        lf.visits = p.reserved;
        if (p.kind == RT_PRIM_GROUP) memcpy(&lf.visits, &p.f[5], 4); // a box: p1.z in the last word (visits do not matter to a surface)
        // self-contained forms (rt_bvh.h): a medium inside a sphere or a box, a moving sphere
        rt_leaf ext{};
        bool two = false;
        if (p.kind == RT_PRIM_MEDIUM && shape.kind == RT_PRIM_SPHERE) {
            for (int k = 0; k < 4; ++k) lf.f[k] = shape.f[k];
            lf.f[4] = p.f[0]; // density
            lf.kind_chain = (uint32_t)RT_LEAF_MEDIUM_SPHERE | ((uint32_t)shape.chain << 8);
        } else if (p.kind == RT_PRIM_MEDIUM && shape.kind == RT_PRIM_GROUP && (shape.flags & RT_PRIM_IS_BOX)) {
            for (int k = 0; k < 5; ++k) lf.f[k] = shape.f[k];
            memcpy(&lf.visits, &shape.f[5], 4);
            lf.kind_chain = (uint32_t)RT_LEAF_MEDIUM_BOX | ((uint32_t)shape.chain << 8);
            ext.f[0] = p.f[0], ext.visits = p.reserved, ext.prim = pi, ext.kind_chain = lf.kind_chain;
            two = true;
        } else if (p.kind == RT_PRIM_MOVING_SPHERE) { // f = c0 c1 r t0 t1 (include/rt_amd.h)
            lf.f[0] = p.f[0], lf.f[1] = p.f[1], lf.f[2] = p.f[2], lf.f[3] = p.f[6], lf.f[4] = p.f[7];
            const float inv_dt = 1.0f / (p.f[8] - p.f[7]);
            memcpy(&lf.visits, &inv_dt, 4);
            lf.kind_chain = (uint32_t)RT_LEAF_MOVING_SPHERE | ((uint32_t)shape.chain << 8);
            ext.f[0] = p.f[3] - p.f[0], ext.f[1] = p.f[4] - p.f[1], ext.f[2] = p.f[5] - p.f[2], ext.prim = pi, ext.kind_chain = lf.kind_chain;
            two = true;
        }
        leaves.push_back(lf);
        const int32_t leaf_ref = ~(int32_t)(leaves.size() - 1);
        if (two) leaves.push_back(ext);
        return leaf_ref;
    };
    nodes.resize(tree.size());
    for (size_t i = 0; i < tree.size(); ++i) {
        const Node32 &n = tree[i];
        rt_bvh_node &o = nodes[i];
        for (int k = 0; k < 3; ++k) {
            o.box0[k][0] = to_half(n.lo0[k], false), o.box0[k][1] = to_half(n.hi0[k], true);
            o.box1[k][0] = to_half(n.lo1[k], false), o.box1[k][1] = to_half(n.hi1[k], true);
        }
        o.ref0 = leaf_of(n.ref0), o.ref1 = leaf_of(n.ref1);
    }
    return true;
}
