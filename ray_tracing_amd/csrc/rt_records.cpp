// rt_records.cpp -- builds the register-resident record table (rt_records.h) from a scene description.
#include "rt_records.h"

#include "rt_box.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

namespace {

uint32_t fbits(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}

bool plain_static(int kind) { return kind == RT_PRIM_SPHERE || (kind >= RT_PRIM_XY_RECT && kind <= RT_PRIM_YZ_RECT); }

// the way back up an instance chain (innermost wrapper first), src/transforms.h:21, 114-122
void chain_up_point(const rt_scene_desc *d, int chain, float p[3])
{
    const rt_chain &c = d->chains[chain];
    for (int i = c.count - 1; i >= 0; --i) {
        const rt_xform &x = d->xforms[c.first + i];
        if (x.kind == RT_XFORM_TRANSLATE) {
            p[0] += x.a, p[1] += x.b, p[2] += x.c;
        } else {
            float s = x.a, co = x.b, px = p[0], pz = p[2];
            p[0] = co * px + s * pz;
            p[2] = -s * px + co * pz;
        }
    }
}
void chain_up_normal(const rt_scene_desc *d, int chain, float n[3])
{
    const rt_chain &c = d->chains[chain];
    for (int i = c.count - 1; i >= 0; --i) {
        const rt_xform &x = d->xforms[c.first + i];
        if (x.kind == RT_XFORM_ROTATE_Y) {
            float s = x.a, co = x.b, nx = n[0], nz = n[2];
            n[0] = co * nx + s * nz;
            n[2] = -s * nx + co * nz;
        }
    }
}

// the light records (important_hitables), from record `at` on; returns the index after them
int emit_lights(const rt_scene_desc *d, rt_record_table *out, int at)
{
    for (int l = 0; l < d->n_lights; ++l, ++at) {
        uint32_t *r = out->rec[at];
        int pi = d->lights[l];
        if (pi < 0) {
            r[7] = RT_REC_LIGHT_NONE;
            continue;
        }
        const rt_prim &p = d->prims[pi];
        for (int k = 0; k < 5; ++k) r[k] = fbits(p.f[k]);
        if (p.kind == RT_PRIM_XZ_RECT) {
            r[5] = fbits((p.f[1] - p.f[0]) * (p.f[3] - p.f[2])); // area, src/plane.h:82
            r[6] = fbits(1.0f / ((p.f[1] - p.f[0]) * (p.f[3] - p.f[2])));
            r[7] = RT_REC_LIGHT_RECT;
        } else {
            r[7] = RT_REC_LIGHT_SPHERE;
        }
    }
    return at;
}

} // namespace

extern "C" int rt_records_build(const rt_scene_desc *d, int keep_boxes, rt_record_table *out, const char **why)
{
    static const char *ok = "";
    *why = ok;
    memset(out, 0, sizeof *out);
    for (int i = 0; i < RT_REC_MAX; ++i) out->user_id[i] = out->prim[i] = -1;
    memset(out->face_user, 0xff, sizeof out->face_user);
    float box_lo[3], box_hi[3];
    int box_face[6];

    // ---- does the scene qualify, and how many records does it need? ----------------
    std::vector<int> map(d->n_entries + 1, 0); // entry index -> stream index
    std::map<int, int> chain_first;            // chain id -> first record of its operations
    int n_stream = 0;
    std::vector<int> chains_used;
    auto use_chain = [&](int c) {
        if (c != 0 && !chain_first.count(c)) {
            chain_first[c] = -1;
            chains_used.push_back(c);
        }
    };
    // A box test pays when it lets the WHOLE wave jump over a subtree.  With incoherent rays and
    // a handful of primitives that never happens, and the 64 lanes test every primitive anyway:
    // such scenes are stored without their inner nodes (a box only ever culls, it never adds a hit).
    int n_leaf_records = 0;
    for (int e = 0; e < d->n_entries; ++e)
        if ((d->entries[e].b & 255) == RT_ENTRY_LEAF) {
            const rt_prim &p = d->prims[d->entries[e].a];
            n_leaf_records += p.kind == RT_PRIM_GROUP && !rt_as_box(d, p, box_lo, box_hi, box_face) ? p.count : 1;
        }
    // (keep_boxes: test aid -- small scenes keep their box records too, so the tests reach the generic scan)
    const bool keep_inner = n_leaf_records > RT_REC_FLAT_LEAVES || keep_boxes != 0;
    for (int e = 0; e < d->n_entries; ++e) {
        map[e] = n_stream;
        const rt_entry &en = d->entries[e];
        if ((en.b & 255) == RT_ENTRY_INNER) {
            if (!keep_inner) continue;
            use_chain(en.b >> 8);
            ++n_stream;
            continue;
        }
        const rt_prim &p = d->prims[en.a];
        if (rt_as_box(d, p, box_lo, box_hi, box_face)) { // one slab test instead of six rect tests
            use_chain(p.chain);
            n_stream += 2;
        } else if (p.kind == RT_PRIM_GROUP) {
            for (int k = 0; k < p.count; ++k) {
                const rt_prim &c = d->prims[p.first + k];
                if (!plain_static(c.kind)) { *why = "a list member is not a static sphere or rect"; return 0; }
                if (c.flags & RT_PRIM_NEEDS_UV) { *why = "a texture reads (u,v)"; return 0; }
                // Members of a list compete in the scan like leaves of their own (see emit below).  That equals
                // hitable_list::hit for rects, whose test accepts t == closest_so_far (src/plane.h:13); a sphere's
                // test is strict (src/sphere.h:31), so a list that holds one is left to the generic path.
                if (c.kind == RT_PRIM_SPHERE && p.count > 1) { *why = "a list holds a sphere"; return 0; }
                if (c.kind != RT_PRIM_SPHERE) use_chain(p.chain);
            }
            n_stream += p.count;
        } else if (plain_static(p.kind)) {
            if (p.flags & RT_PRIM_NEEDS_UV) { *why = "a texture reads (u,v)"; return 0; }
            if (p.kind != RT_PRIM_SPHERE) use_chain(p.chain);
            ++n_stream;
        } else {
            *why = "scene has a constant_medium or a moving_sphere";
            return 0;
        }
    }
    map[d->n_entries] = n_stream;
    int n_chain_recs = 0;
    for (int c : chains_used) {
        if (d->chains[c].count > 7) { *why = "an instance chain is longer than 7"; return 0; }
        n_chain_recs += d->chains[c].count;
    }
    if (d->n_materials > 255) { *why = "more than 255 materials"; return 0; }
    int total = n_stream + n_chain_recs + d->n_materials + d->n_lights;
    if (total > RT_REC_MAX) { *why = "more than 64 records"; return 0; }

    // ---- lay out: stream | chain operations | materials | lights ---------------------
    int at = n_stream;
    for (int c : chains_used) {
        chain_first[c] = at;
        for (int k = 0; k < d->chains[c].count; ++k, ++at) {
            const rt_xform &x = d->xforms[d->chains[c].first + k];
            uint32_t *r = out->rec[at];
            r[0] = fbits(x.a), r[1] = fbits(x.b), r[2] = fbits(x.c);
            r[7] = x.kind == RT_XFORM_TRANSLATE ? RT_REC_XFORM_T : RT_REC_XFORM_R;
        }
    }
    auto chain_word = [&](int c) -> uint32_t {
        if (c == 0) return 0;
        return ((uint32_t)chain_first[c] | ((uint32_t)d->chains[c].count << 6)) << RT_KW_CHAIN_SHIFT;
    };
    out->materials_base = at;
    for (int m = 0; m < d->n_materials; ++m, ++at) {
        const rt_material &mt = d->materials[m];
        uint32_t *r = out->rec[at];
        r[0] = (uint32_t)mt.kind;
        r[2] = fbits(mt.p0);
        int tex = mt.texture;
        if (tex >= 0 && d->textures[tex].kind == RT_TEX_CONSTANT) {
            r[1] = (uint32_t)-1;
            for (int k = 0; k < 3; ++k) r[3 + k] = fbits(d->textures[tex].color[k]);
        } else {
            r[1] = (uint32_t)tex; // -1 for dielectric
            for (int k = 0; k < 3; ++k) r[3 + k] = fbits(1.0f);
        }
        r[7] = RT_REC_MATERIAL;
    }
    out->lights_base = at;
    out->n_lights = d->n_lights;
    at = emit_lights(d, out, at);
    out->n_records = at;
    out->n_stream = n_stream;
    out->flat = keep_inner ? 0 : 1;

    // ---- the traversal stream ------------------------------------------------------------
    auto emit_prim = [&](int s, int prim_index, const rt_prim &p, int chain, uint32_t list_bits) {
        uint32_t *r = out->rec[s];
        uint32_t kw = list_bits | ((uint32_t)p.material << RT_KW_MAT_SHIFT);
        if (p.flags & RT_PRIM_FLIP) kw |= RT_KW_FLIP;
        if (p.kind == RT_PRIM_SPHERE) {
            float c[3] = {p.f[0], p.f[1], p.f[2]};
            chain_up_point(d, chain, c);
            for (int k = 0; k < 3; ++k) r[k] = fbits(c[k]);
            r[3] = fbits(p.f[3]);
            r[4] = fbits(1.0f / p.f[3]);
            kw |= RT_REC_SPHERE;
        } else {
            for (int k = 0; k < 5; ++k) r[k] = fbits(p.f[k]);
            int axis = p.kind == RT_PRIM_XY_RECT ? 0 : p.kind == RT_PRIM_XZ_RECT ? 1 : 2;
            float n[3] = {axis == 2 ? 1.0f : 0.0f, axis == 1 ? 1.0f : 0.0f, axis == 0 ? 1.0f : 0.0f};
            if (p.flags & RT_PRIM_FLIP)
                for (float &x : n) x = -x;
            chain_up_normal(d, chain, n);
            r[5] = fbits(n[0]), r[6] = fbits(n[2]);
            kw |= RT_REC_RECT | ((uint32_t)axis << RT_KW_AXIS_SHIFT) | chain_word(chain);
        }
        r[7] = kw;
        out->user_id[s] = p.user_id;
        out->prim[s] = prim_index;
    };
    for (int e = 0; e < d->n_entries; ++e) {
        const rt_entry &en = d->entries[e];
        int s = map[e];
        if ((en.b & 255) == RT_ENTRY_INNER) {
            if (!keep_inner) continue;
            uint32_t *r = out->rec[s];
            for (int k = 0; k < 3; ++k) r[k] = fbits(en.lo[k]), r[4 + k] = fbits(en.hi[k]);
            r[3] = (uint32_t)map[en.a];
            r[7] = RT_REC_INNER | chain_word(en.b >> 8);
            continue;
        }
        const rt_prim &p = d->prims[en.a];
        if (rt_as_box(d, p, box_lo, box_hi, box_face)) {
            // The nearest of the six faces inside (t_min, best] is where the ray enters the slabs' intersection, or where
            // it leaves it when the entry lies before t_min: the scan does that one test (rt_fast.hpp) and the face is
            // named afterwards from the hit point, with what the second record holds.
            uint32_t *r = out->rec[s], *x = out->rec[s + 1];
            float c[3], ux[3] = {1.0f, 0.0f, 0.0f};
            for (int k = 0; k < 3; ++k) r[k] = fbits(box_lo[k]), r[4 + k] = fbits(box_hi[k]), c[k] = 0.5f * (box_lo[k] + box_hi[k]);
            chain_up_point(d, p.chain, c);
            chain_up_normal(d, p.chain, ux);
            for (int k = 0; k < 3; ++k) x[k] = fbits(c[k]);
            x[3] = fbits(ux[0]), x[4] = fbits(ux[2]);
            const uint32_t common = (3u << RT_KW_AXIS_SHIFT) | chain_word(p.chain) | RT_KW_FIRST | RT_KW_LAST;
            r[7] = RT_REC_BOX | common | ((uint32_t)d->prims[p.first].material << RT_KW_MAT_SHIFT);
            x[7] = RT_REC_BOX_AUX | common;
            out->prim[s] = en.a;
            for (int k = 0; k < 6; ++k) out->face_user[s][k] = d->prims[box_face[k]].user_id;
        } else if (p.kind == RT_PRIM_GROUP) {
            // hitable_list::hit (src/hitable_list.h:18-34) keeps the nearest member, the later one on a tie, and
            // bvh_node::hit lets that result replace the best so far unless the best is strictly nearer
            // (src/bvh.h:132-139).  Letting every member compete with the best so far under the same rule
            // (t <= best) ends with the same (t, member): the minimum, latest on ties.  So members are emitted
            // as leaves of their own and the scan needs no list state.
            for (int k = 0; k < p.count; ++k)
                emit_prim(s + k, p.first + k, d->prims[p.first + k], p.chain, RT_KW_FIRST | RT_KW_LAST);
        } else {
            emit_prim(s, en.a, p, p.chain, RT_KW_FIRST | RT_KW_LAST);
        }
    }
    // Ranks: stream records are in the order of the reference's depth-first walk here (entries are stored in pre-order).
    // A primitive the walk meets twice (the only object of a one-object node) takes the rank of its last visit.
    {
        std::map<int, int> last; // prim -> stream index of its last visit
        for (int i = 0; i < n_stream; ++i)
            if (out->prim[i] >= 0) last[out->prim[i]] = i;
        for (int i = 0; i < n_stream; ++i)
            if (out->prim[i] >= 0) out->rec[i][7] |= ((uint32_t)last[out->prim[i]] & RT_KW_RANK_MASK) << RT_KW_RANK_SHIFT;
    }
    // A flat stream is stored in runs: chain by chain (world space first), within a chain the rects by axis and then
    // the boxes (each with its second record: the stable sort keeps the pair together), and all spheres (world space) at
    // the end.  The scan (rt_fast.hpp) then runs one tight loop per run, with nothing to decode
    // per record; the run lengths travel in the kernel arguments.  (More than RT_REC_MAX_RUNS chains: n_runs = 0 and the
    // generic scan, which decodes every record, is used.)
    // (The order of a flat stream decides nothing: of two hits at the same distance the rank decides, rt_records.h.)
    if (out->flat) {
        std::vector<int> order(n_stream);
        for (int i = 0; i < n_stream; ++i) order[i] = i;
        auto chain_of = [&](const uint32_t *r) { return (r[7] >> RT_KW_CHAIN_SHIFT) & RT_KW_CHAIN_MASK; };
        auto key = [&](const uint32_t *r) {
            if ((r[7] & 15u) == RT_REC_SPHERE) return 0xffffffffu;
            return (chain_of(r) << 2) | ((r[7] >> RT_KW_AXIS_SHIFT) & 3u);
        };
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return key(out->rec[a]) < key(out->rec[b]); });
        // The reference's tree holds the only object of a one-object node twice (left == right, src/bvh.h:92-94), and so does
        // the entry array.  Without media a second test of the same primitive changes nothing: it is dropped here.
        {
            std::vector<int> kept;
            std::vector<char> seen(d->n_prims, 0);
            for (int i : order) {
                const bool aux = (out->rec[i][7] & 15u) == RT_REC_BOX_AUX;
                const int prim = aux ? out->prim[i - 1] : out->prim[i];
                if (!aux && seen[prim]) continue;
                if (aux && seen[prim] == 2) continue;
                if (!aux) seen[prim] = 1;
                else seen[prim] = 2; // the second half of a box goes with its first occurrence only
                kept.push_back(i);
            }
            order.swap(kept);
        }
        rt_record_table tmp = *out;
        for (int i = (int)order.size(); i < n_stream; ++i) {
            memset(out->rec[i], 0, sizeof out->rec[i]);
            out->user_id[i] = out->prim[i] = -1;
            memset(out->face_user[i], 0xff, sizeof out->face_user[i]);
        }
        n_stream = (int)order.size();
        out->n_stream = n_stream;
        for (int i = 0; i < n_stream; ++i) {
            memcpy(out->rec[i], tmp.rec[order[i]], sizeof out->rec[i]);
            out->user_id[i] = tmp.user_id[order[i]];
            out->prim[i] = tmp.prim[order[i]];
            memcpy(out->face_user[i], tmp.face_user[order[i]], sizeof out->face_user[i]);
        }
        int n_runs = 0;
        bool fits = true;
        uint32_t counts[RT_REC_MAX_RUNS][4] = {}; // xy, xz, yz rects; box records (two per box)
        uint32_t chains[RT_REC_MAX_RUNS] = {};
        for (int i = 0; i < n_stream; ++i) {
            uint32_t *r = out->rec[i];
            if ((r[7] & 15u) == RT_REC_SPHERE) {
                ++out->n_spheres;
                continue;
            }
            const bool first = i == 0 || chain_of(out->rec[i - 1]) != chain_of(r);
            if (first) {
                if (n_runs == RT_REC_MAX_RUNS) { fits = false; continue; }
                chains[n_runs++] = chain_of(r);
            }
            if (fits && ++counts[n_runs - 1][(r[7] >> RT_KW_AXIS_SHIFT) & 3u] > 30u) fits = false;
        }
        out->n_runs = fits ? n_runs : 0;
        if (fits && n_runs == 0 && out->n_spheres > 0) out->n_runs = 1; // spheres only: one empty run in world space
        for (int k = 0; k < RT_REC_MAX_RUNS; ++k)
            out->run[k] = fits ? ((counts[k][3] / 2) << 24) | (chains[k] << 15) | (counts[k][0] << 10) | (counts[k][1] << 5) | counts[k][2] : 0u;
    }
    return 1;
}

// A table that holds only the light records (for scenes that do not fit: the per-lane fast path samples the
// important list from it, the scene itself stays in the generic tables).
extern "C" int rt_records_build_lights(const rt_scene_desc *d, rt_record_table *out)
{
    memset(out, 0, sizeof *out);
    for (int i = 0; i < RT_REC_MAX; ++i) out->user_id[i] = out->prim[i] = -1;
    if (d->n_lights > RT_REC_MAX) return 0;
    // lights must be usable from WORLD space: a light under an instance chain has no pdf_value/random of its own
    // in the reference either (translate / rotate_y do not forward them, src/hitable.h:19-20); the flattener
    // writes -1 for those.
    out->lights_base = 0;
    out->n_lights = d->n_lights;
    out->n_records = emit_lights(d, out, 0);
    out->flat = 1;
    return 1;
}
