// rt_order.cpp -- builds the side table of rt_order.h from a validated scene description.
#include "rt_order.h"

#include <cmath>

namespace {

struct Walk {
    const rt_scene_desc *d;
    std::vector<int32_t> *out;
    int max_depth;
    bool ok;
};

bool prim_centre(const rt_scene_desc *d, const rt_prim &p, float c[3])
{
    switch (p.kind) {
    case RT_PRIM_SPHERE: c[0] = p.f[0], c[1] = p.f[1], c[2] = p.f[2]; return true;
    case RT_PRIM_MOVING_SPHERE:
        for (int k = 0; k < 3; ++k) c[k] = 0.5f * (p.f[k] + p.f[3 + k]);
        return true;
    case RT_PRIM_XY_RECT: c[0] = 0.5f * (p.f[0] + p.f[1]), c[1] = 0.5f * (p.f[2] + p.f[3]), c[2] = p.f[4]; return true;
    case RT_PRIM_XZ_RECT: c[0] = 0.5f * (p.f[0] + p.f[1]), c[2] = 0.5f * (p.f[2] + p.f[3]), c[1] = p.f[4]; return true;
    case RT_PRIM_YZ_RECT: c[1] = 0.5f * (p.f[0] + p.f[1]), c[2] = 0.5f * (p.f[2] + p.f[3]), c[0] = p.f[4]; return true;
    case RT_PRIM_GROUP: {
        c[0] = c[1] = c[2] = 0.0f;
        for (int k = 0; k < p.count; ++k) {
            float m[3];
            if (!prim_centre(d, d->prims[p.first + k], m)) return false;
            for (int a = 0; a < 3; ++a) c[a] += m[a] / (float)p.count;
        }
        return true;
    }
    default: return false;
    }
}

// centre of entry e in the space of chain `chain`; false if it lives in another space
bool entry_centre(const rt_scene_desc *d, int e, int chain, float c[3])
{
    const rt_entry &en = d->entries[e];
    if ((en.b & 255) == RT_ENTRY_INNER) {
        if ((en.b >> 8) != chain) return false;
        for (int k = 0; k < 3; ++k) c[k] = 0.5f * (en.lo[k] + en.hi[k]);
        return true;
    }
    const rt_prim &p = d->prims[en.a];
    return p.chain == chain && prim_centre(d, p, c);
}

// returns the index after the subtree of entry i (or -1), sets has_medium
int subtree(Walk &w, int i, int depth, bool &has_medium)
{
    const rt_scene_desc *d = w.d;
    if (i < 0 || i >= d->n_entries || depth > RT_ORD_MAX_DEPTH) return w.ok = false, -1;
    if (depth > w.max_depth) w.max_depth = depth;
    const rt_entry &en = d->entries[i];
    if ((en.b & 255) != RT_ENTRY_INNER) {
        has_medium = d->prims[en.a].kind == RT_PRIM_MEDIUM;
        return i + 1;
    }
    bool ml = false, mr = false;
    const int left = i + 1;
    const int right = subtree(w, left, depth + 1, ml);
    if (!w.ok) return -1;
    const int end = subtree(w, right, depth + 1, mr);
    if (!w.ok || end != en.a) return w.ok = false, -1; // the skip link must close exactly two children
    has_medium = ml || mr;
    int flags = has_medium ? RT_ORD_HAS_MEDIUM : 0;
    float cl[3], cr[3];
    const int chain = en.b >> 8;
    if (entry_centre(d, left, chain, cl) && entry_centre(d, right, chain, cr)) {
        int axis = 0;
        for (int k = 1; k < 3; ++k)
            if (std::fabs(cl[k] - cr[k]) > std::fabs(cl[axis] - cr[axis])) axis = k;
        flags |= axis | (cl[axis] <= cr[axis] ? RT_ORD_LEFT_LOWER : 0);
    } else {
        flags |= RT_ORD_NO_PREFERENCE;
    }
    (*w.out)[2 * i] = right;
    (*w.out)[2 * i + 1] = flags;
    return end;
}

} // namespace

bool rt_order_build(const rt_scene_desc *d, std::vector<int32_t> &out, int *max_depth)
{
    out.assign(2 * (size_t)d->n_entries, 0);
    Walk w{d, &out, 0, true};
    bool m = false;
    const int end = subtree(w, 0, 1, m);
    *max_depth = w.max_depth;
    return w.ok && end == d->n_entries;
}
