// bvh_check.cpp -- host-side checks of ray_tracing_amd/csrc/rt_bvh.cpp and rt_order.cpp (built and run by tests/test_host_trees.py).
// Includes the builder's source to reach its internals (the directed float -> half rounding).
#include "../../ray_tracing_amd/csrc/rt_bvh.cpp"
#include "../../ray_tracing_amd/csrc/rt_order.h"
#include "../../ray_tracing_amd/csrc/rt_box.h"
#include "../../include/rt_host.h"

#include <cstdio>
#include <cstring>
#include <random>

static float half_value(uint16_t h)
{
    const int s = h >> 15, e = (h >> 10) & 31, m = h & 1023;
    double v = e == 0 ? m * std::ldexp(1.0, -24) : e == 31 ? (m ? NAN : INFINITY) : (1.0 + m / 1024.0) * std::ldexp(1.0, e - 15);
    return (float)(s ? -v : v);
}

static int check_halves()
{
    int bad = 0;
    auto check = [&](float x) {
        const uint16_t a = to_half(x, false), b = to_half(x, true);
        const float d = half_value(a), u = half_value(b);
        if (!(d <= x && x <= u)) ++bad; // brackets x
        const int ia = (a & 0x8000) ? -(a & 0x7fff) : (a & 0x7fff), ib = (b & 0x8000) ? -(b & 0x7fff) : (b & 0x7fff);
        if (std::isfinite(d) && std::isfinite(u) && d != u && ib - ia != 1) ++bad; // and is tight: neighbouring halves
        if (d == x && u != x) ++bad; // a representable value maps to itself both ways
    };
    const float specials[] = {0.f, -0.f, 1.f, -1.f, 65504.f, 65505.f, -65505.f, 1e6f, -1e6f, 1e-8f, -1e-8f, 6.1e-5f, 5.96e-8f, 3e-8f, 0.1f, 1000.02f, -5000.1f, 554.9999f, 2048.f, 2049.f};
    for (float x : specials) check(x);
    std::mt19937 g(1);
    std::uniform_real_distribution<float> u(-1, 1);
    for (int i = 0; i < 400000; ++i) check(u(g) * std::pow(10.f, (float)(int)(u(g) * 6)));
    return bad;
}

struct Stats {
    int nodes = 0, leaves = 0, depth = 0, bad_box = 0, dup = 0, missing = 0;
    // what the half-float rounding costs: the surface area (~ the chance that a ray enters) of the boxes as stored against the
    // exact boxes of the same subtrees, over the boxes of leaves and over those of inner nodes
    double leaf_area_half = 0, leaf_area_exact = 0, node_area_half = 0, node_area_exact = 0;
};
static double box_area(const float lo[3], const float hi[3])
{
    const double e[3] = {(double)hi[0] - lo[0], (double)hi[1] - lo[1], (double)hi[2] - lo[2]};
    return e[0] * e[1] + e[1] * e[2] + e[2] * e[0];
}

// every leaf's primitive box (world space, recomputed here) must lie inside the half-float box its parent holds for it
static Box walk(const rt_scene_desc *d, const std::vector<rt_bvh_node> &nodes, const std::vector<rt_leaf> &leaves, int32_t ref,
                 const float lo[3], const float hi[3], int depth, std::vector<int> &seen, Stats &st)
{
    Box exact;
    exact.clear();
    if (ref == RT_BVH_NONE) return exact;
    st.depth = std::max(st.depth, depth);
    if (ref < 0) {
        ++st.leaves;
        rt_leaf lf = leaves[~ref]; // (the flag for noise-textured primitives is not part of the index)
        lf.prim &= RT_LEAF_PRIM_MASK;
        if (seen[lf.prim]++) ++st.dup;
        Box b;
        if (!world_box(d, d->prims[lf.prim], b)) ++st.bad_box;
        for (int k = 0; k < 3; ++k)
            if (!(lo[k] <= b.lo[k] && b.hi[k] <= hi[k])) ++st.bad_box;
        // the leaf's kind is the primitive's, or one of the self-contained forms of a medium / a moving sphere (rt_bvh.h);
        // the forms of two slots are followed by a slot that no node refers to and that names the same primitive
        const int lk = (int)(lf.kind_chain & 255u), pk = d->prims[lf.prim].kind;
        const bool form_ok = lk == pk || ((lk == RT_LEAF_MEDIUM_SPHERE || lk == RT_LEAF_MEDIUM_BOX) && pk == RT_PRIM_MEDIUM) ||
                             (lk == RT_LEAF_MOVING_SPHERE && pk == RT_PRIM_MOVING_SPHERE);
        if (!form_ok) ++st.bad_box;
        if (lk == RT_LEAF_MEDIUM_BOX || lk == RT_LEAF_MOVING_SPHERE) {
            if ((size_t)(~ref) + 1 >= leaves.size() || (leaves[~ref + 1].prim & RT_LEAF_PRIM_MASK) != lf.prim) ++st.bad_box;
            ++st.leaves; // the second slot
        }
        if (std::isfinite(lo[0]) && std::isfinite(box_area(lo, hi))) st.leaf_area_half += box_area(lo, hi), st.leaf_area_exact += box_area(b.lo, b.hi);
        return b;
    }
    ++st.nodes;
    const rt_bvh_node &n = nodes[ref];
    float l0[3], h0[3], l1[3], h1[3];
    for (int k = 0; k < 3; ++k) {
        l0[k] = half_value(n.box0[k][0]), h0[k] = half_value(n.box0[k][1]), l1[k] = half_value(n.box1[k][0]), h1[k] = half_value(n.box1[k][1]);
        // children inside the parent (the root's parent is everything)
        if (!(lo[k] <= l0[k] && h0[k] <= hi[k])) ++st.bad_box;
        if (n.ref1 != RT_BVH_NONE && !(lo[k] <= l1[k] && h1[k] <= hi[k])) ++st.bad_box;
    }
    const Box e0 = walk(d, nodes, leaves, n.ref0, l0, h0, depth + 1, seen, st);
    const Box e1 = walk(d, nodes, leaves, n.ref1, l1, h1, depth + 1, seen, st);
    exact.add(e0);
    if (n.ref1 != RT_BVH_NONE) exact.add(e1);
    if (std::isfinite(lo[0]) && std::isfinite(box_area(lo, hi))) st.node_area_half += box_area(lo, hi), st.node_area_exact += box_area(exact.lo, exact.hi);
    return exact;
}

int main(int argc, char **argv)
{
    int failures = 0;
    const int bad_halves = check_halves();
    printf("half rounding: %d bad\n", bad_halves);
    failures += bad_halves;
    for (int a = 1; a < argc; ++a) {
        rt_host_scene *h = nullptr;
        if (rt_host_scene_build(argv[a], 1.5f, &h) != RT_OK) {
            printf("%s: build failed: %s\n", argv[a], rt_last_error());
            ++failures;
            continue;
        }
        const rt_scene_desc *d = rt_host_scene_desc(h);
        std::vector<rt_bvh_node> nodes;
        std::vector<rt_leaf> leaves;
        std::vector<rt_prim> prims;
        int depth = 0;
        const bool ok = rt_bvh_build(d, nodes, leaves, prims, &depth);
        Stats st;
        std::vector<int> seen((size_t)d->n_prims, 0);
        const float lo[3] = {-INFINITY, -INFINITY, -INFINITY}, hi[3] = {INFINITY, INFINITY, INFINITY};
        if (ok) walk(d, nodes, leaves, 0, lo, hi, 1, seen, st);
        // every primitive the entry array names (list members each) must be a leaf exactly once
        for (int e = 0; e < d->n_entries; ++e) {
            if ((d->entries[e].b & 255) != RT_ENTRY_LEAF) continue;
            const rt_prim &p = d->prims[d->entries[e].a];
            float blo[3], bhi[3];
            int bface[6];
            if (p.kind == RT_PRIM_GROUP && !rt_as_box(d, p, blo, bhi, bface)) {
                for (int k = 0; k < p.count; ++k) st.missing += seen[p.first + k] != 1;
            } else { // (a list that is a box is one leaf)
                st.missing += seen[d->entries[e].a] != 1;
            }
        }
        if (ok && !nodes.empty()) // the top of the tree: which references are leaves (a scene-wide primitive ends up beside the root)
            printf("%s: root children %s / %s", argv[a], nodes[0].ref0 < 0 ? "LEAF" : "node", nodes[0].ref1 < 0 ? (nodes[0].ref1 == RT_BVH_NONE ? "none" : "LEAF") : "node");
        if (ok && !nodes.empty() && nodes[0].ref0 >= 0)
            printf("; of child 0: %s / %s", nodes[nodes[0].ref0].ref0 < 0 ? "LEAF" : "node", nodes[nodes[0].ref0].ref1 < 0 ? "LEAF" : "node");
        if (ok && !nodes.empty() && nodes[0].ref1 >= 0)
            printf("; of child 1: %s / %s", nodes[nodes[0].ref1].ref0 < 0 ? "LEAF" : "node", nodes[nodes[0].ref1].ref1 < 0 ? "LEAF" : "node");
        if (ok) printf("\n");
        if (ok)
            printf("%s: area of the stored (half-float) boxes / exact boxes: leaves %.3f, inner nodes %.3f\n", argv[a],
                   st.leaf_area_half / std::max(st.leaf_area_exact, 1e-30), st.node_area_half / std::max(st.node_area_exact, 1e-30));
        std::vector<int32_t> order;
        int odepth = 0;
        const bool ordered = rt_order_build(d, order, &odepth);
        printf("%s: ok=%d nodes=%d leaves=%d depth=%d(%d) bad_box=%d dup=%d missing=%d | reference tree: entries=%d binary=%d depth=%d\n",
               argv[a], ok, st.nodes, st.leaves, st.depth, depth, st.bad_box, st.dup, st.missing, d->n_entries, ordered, odepth);
        if (!ok || st.bad_box || st.dup || st.missing || st.depth > depth + 1 || depth > RT_BVH_DEPTH_LIMIT || (int)leaves.size() != st.leaves || !ordered) ++failures;
        rt_host_scene_free(h);
    }
    printf("failures: %d\n", failures);
    return failures ? 1 : 0;
}
