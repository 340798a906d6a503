// rt_flatten.cpp -- see rt_flatten.h
#include "rt_flatten.h"

#include <cstring>
#include <tuple>

namespace {

struct unsupported {
    std::string what;
};

struct flattener {
    rt_flat_scene &out;
    std::map<const texture *, int> tex_index;
    std::map<const material *, int> mat_index;
    std::map<std::tuple<const hitable *, int, unsigned>, int> prim_index;
    std::map<std::pair<int, const hitable *>, int> chain_index;
    int next_user_id = 0;

    explicit flattener(rt_flat_scene &o) : out(o)
    {
        out.chains.push_back(rt_chain{0, 0}); // chain 0: identity
    }

    // -- textures / materials ------------------------------------------------
    int add_texture(const texture *t)
    {
        if (!t) throw unsupported{"null texture"};
        auto it = tex_index.find(t);
        if (it != tex_index.end()) return it->second;
        rt_texture r;
        memset(&r, 0, sizeof r);
        r.a = r.b = -1;
        switch (t->tex_kind()) {
        case RT_HT_CONSTANT: {
            const constant_texture *c = static_cast<const constant_texture *>(t);
            r.kind = RT_TEX_CONSTANT;
            for (int i = 0; i < 3; ++i) r.color[i] = c->color[i];
            break;
        }
        case RT_HT_CHECKER: {
            const checker_texture *c = static_cast<const checker_texture *>(t);
            r.kind = RT_TEX_CHECKER;
            r.a = add_texture(c->odd);
            r.b = add_texture(c->even);
            break;
        }
        case RT_HT_NOISE:
            r.kind = RT_TEX_NOISE;
            r.scale = static_cast<const noise_texture *>(t)->scale;
            break;
        case RT_HT_IMAGE: {
            const image_texture *im = static_cast<const image_texture *>(t);
            if (!im->data || im->nx <= 0 || im->ny <= 0) throw unsupported{"image_texture without pixels"};
            r.kind = RT_TEX_IMAGE;
            r.nx = im->nx;
            r.ny = im->ny;
            r.offset = (int64_t)out.texels.size();
            out.texels.insert(out.texels.end(), im->data, im->data + (size_t)3 * im->nx * im->ny);
            break;
        }
        }
        int idx = (int)out.textures.size();
        out.textures.push_back(r);
        tex_index[t] = idx;
        return idx;
    }

    bool texture_reads_uv(int idx) const
    {
        const rt_texture &t = out.textures[idx];
        if (t.kind == RT_TEX_IMAGE) return true;
        if (t.kind == RT_TEX_CHECKER) return texture_reads_uv(t.a) || texture_reads_uv(t.b);
        return false;
    }

    int add_material(const material *m)
    {
        if (!m) throw unsupported{"primitive without material"};
        auto it = mat_index.find(m);
        if (it != mat_index.end()) return it->second;
        rt_material r;
        memset(&r, 0, sizeof r);
        r.texture = -1;
        switch (m->mat_kind()) {
        case RT_HM_LAMBERTIAN:
            r.kind = RT_MAT_LAMBERTIAN;
            r.texture = add_texture(static_cast<const lambertian *>(m)->albedo);
            break;
        case RT_HM_METAL:
            r.kind = RT_MAT_METAL;
            r.texture = add_texture(static_cast<const metal *>(m)->albedo);
            r.p0 = static_cast<const metal *>(m)->fuzz;
            break;
        case RT_HM_DIELECTRIC:
            r.kind = RT_MAT_DIELECTRIC;
            r.p0 = static_cast<const dielectric *>(m)->ref_idx;
            break;
        case RT_HM_LIGHT:
            r.kind = RT_MAT_DIFFUSE_LIGHT;
            r.texture = add_texture(static_cast<const diffuse_light *>(m)->emit);
            break;
        case RT_HM_ISOTROPIC:
            r.kind = RT_MAT_ISOTROPIC;
            r.texture = add_texture(static_cast<const isotropic *>(m)->albedo);
            break;
        }
        int idx = (int)out.materials.size();
        out.materials.push_back(r);
        mat_index[m] = idx;
        return idx;
    }

    // -- instance chains -----------------------------------------------------
    int extend_chain(int parent, const hitable *wrapper, const rt_xform &op)
    {
        auto key = std::make_pair(parent, wrapper);
        auto it = chain_index.find(key);
        if (it != chain_index.end()) return it->second;
        rt_chain p = out.chains[parent];
        rt_chain c;
        c.first = (int)out.xforms.size();
        c.count = p.count + 1;
        for (int i = 0; i < p.count; ++i) out.xforms.push_back(out.xforms[p.first + i]);
        out.xforms.push_back(op);
        int idx = (int)out.chains.size();
        out.chains.push_back(c);
        chain_index[key] = idx;
        return idx;
    }

    // Strips flip_normals / translate / rotate_y wrappers, folding them into
    // (chain, flip).  Returns the wrapped object.
    const hitable *unwrap(const hitable *h, int &chain, unsigned &flip)
    {
        for (;;) {
            if (!h) throw unsupported{"null hitable in scene tree"};
            switch (h->kind()) {
            case RT_H_FLIP:
                flip ^= RT_PRIM_FLIP;
                h = static_cast<const flip_normals *>(h)->ptr;
                break;
            case RT_H_TRANSLATE: {
                const translate *t = static_cast<const translate *>(h);
                rt_xform op{RT_XFORM_TRANSLATE, t->offset[0], t->offset[1], t->offset[2]};
                chain = extend_chain(chain, h, op);
                h = t->ptr;
                break;
            }
            case RT_H_ROTATE_Y: {
                const rotate_y *r = static_cast<const rotate_y *>(h);
                rt_xform op{RT_XFORM_ROTATE_Y, r->sin_theta, r->cos_theta, 0.0f};
                chain = extend_chain(chain, h, op);
                h = r->ptr;
                break;
            }
            default:
                return h;
            }
        }
    }

    // -- primitives ------------------------------------------------------------
    rt_prim blank(int kind, int chain, unsigned flip)
    {
        rt_prim p;
        memset(&p, 0, sizeof p);
        p.kind = kind;
        p.flags = flip;
        p.material = -1;
        p.chain = chain;
        p.first = -1;
        p.user_id = -1;
        return p;
    }

    void set_material(rt_prim &p, const material *m)
    {
        p.material = add_material(m);
        int tex = out.materials[p.material].texture;
        if (tex >= 0 && texture_reads_uv(tex)) p.flags |= RT_PRIM_NEEDS_UV;
    }

    // sphere, moving_sphere, rects: fills one record (not yet stored)
    bool fill_plain(const hitable *h, rt_prim &p)
    {
        switch (h->kind()) {
        case RT_H_SPHERE: {
            const sphere *s = static_cast<const sphere *>(h);
            p.kind = RT_PRIM_SPHERE;
            p.f[0] = s->center[0], p.f[1] = s->center[1], p.f[2] = s->center[2], p.f[3] = s->radius;
            set_material(p, s->mat);
            return true;
        }
        case RT_H_MOVING_SPHERE: {
            const moving_sphere *s = static_cast<const moving_sphere *>(h);
            p.kind = RT_PRIM_MOVING_SPHERE;
            for (int i = 0; i < 3; ++i) p.f[i] = s->center0[i], p.f[3 + i] = s->center1[i];
            p.f[6] = s->radius, p.f[7] = s->time0, p.f[8] = s->time1;
            set_material(p, s->mat_ptr);
            return true;
        }
        case RT_H_XY_RECT: {
            const xy_rect *r = static_cast<const xy_rect *>(h);
            p.kind = RT_PRIM_XY_RECT;
            p.f[0] = r->x0, p.f[1] = r->x1, p.f[2] = r->y0, p.f[3] = r->y1, p.f[4] = r->z;
            set_material(p, r->mat_ptr);
            return true;
        }
        case RT_H_XZ_RECT: {
            const xz_rect *r = static_cast<const xz_rect *>(h);
            p.kind = RT_PRIM_XZ_RECT;
            p.f[0] = r->x0, p.f[1] = r->x1, p.f[2] = r->z0, p.f[3] = r->z1, p.f[4] = r->y;
            set_material(p, r->mat_ptr);
            return true;
        }
        case RT_H_YZ_RECT: {
            const yz_rect *r = static_cast<const yz_rect *>(h);
            p.kind = RT_PRIM_YZ_RECT;
            p.f[0] = r->y0, p.f[1] = r->y1, p.f[2] = r->z0, p.f[3] = r->z1, p.f[4] = r->x;
            set_material(p, r->mat_ptr);
            return true;
        }
        default:
            return false;
        }
    }

    // Primitive for `h` (already unwrapped) living in `chain`; plain primitives,
    // boxes and flat lists only.  Returns its index.  with_ids: assign user ids
    // (false for medium boundaries and lights, which are not hit targets).
    int add_simple(const hitable *h, int chain, unsigned flip, bool with_ids)
    {
        auto key = std::make_tuple(h, chain, flip | (with_ids ? 0x100u : 0u));
        auto it = prim_index.find(key);
        if (it != prim_index.end()) return it->second;
        int idx;
        rt_prim p = blank(0, chain, flip);
        if (fill_plain(h, p)) {
            if (with_ids) p.user_id = next_user_id++;
            idx = (int)out.prims.size();
            out.prims.push_back(p);
        } else if (h->kind() == RT_H_BOX || h->kind() == RT_H_LIST) {
            const hitable_list *l = h->kind() == RT_H_BOX
                                        ? static_cast<const hitable_list *>(static_cast<const box *>(h)->list_ptr)
                                        : static_cast<const hitable_list *>(h);
            if (!l || l->list_size < 1) throw unsupported{"empty hitable_list"};
            idx = (int)out.prims.size();
            out.prims.push_back(blank(RT_PRIM_GROUP, chain, 0));
            int first = (int)out.prims.size();
            // reserve the children contiguously
            for (int i = 0; i < l->list_size; ++i) out.prims.push_back(blank(0, chain, 0));
            for (int i = 0; i < l->list_size; ++i) {
                int cchain = chain;
                unsigned cflip = flip;
                const hitable *c = unwrap(l->list[i], cchain, cflip);
                if (cchain != chain)
                    throw unsupported{"translate/rotate_y inside a hitable_list member (only flip_normals is "
                                      "supported there)"};
                rt_prim cp = blank(0, chain, cflip);
                if (!fill_plain(c, cp))
                    throw unsupported{"hitable_list member must be a sphere, moving_sphere or rect"};
                if (with_ids) cp.user_id = next_user_id++;
                out.prims[first + i] = cp;
            }
            out.prims[idx].first = first;
            out.prims[idx].count = l->list_size;
        } else {
            throw unsupported{"this hitable cannot be used here (expected a primitive, box or flat hitable_list)"};
        }
        prim_index[key] = idx;
        return idx;
    }

    void push_leaf(int prim)
    {
        rt_entry e;
        memset(&e, 0, sizeof e);
        e.a = prim;
        e.b = RT_ENTRY_LEAF | (out.prims[prim].chain << 8);
        out.entries.push_back(e);
    }

    // depth-first emission of one subtree
    void emit(const hitable *h, int chain, unsigned flip)
    {
        h = unwrap(h, chain, flip);
        switch (h->kind()) {
        case RT_H_BVH: {
            if (flip) throw unsupported{"flip_normals around a bvh_node"};
            const bvh_node *n = static_cast<const bvh_node *>(h);
            rt_entry e;
            for (int i = 0; i < 3; ++i) e.lo[i] = n->box.min()[i], e.hi[i] = n->box.max()[i];
            e.a = -1;
            e.b = RT_ENTRY_INNER | (chain << 8);
            size_t at = out.entries.size();
            out.entries.push_back(e);
            emit(n->left, chain, 0);
            emit(n->right, chain, 0); // left == right (n == 1) is visited twice, as the reference does
            out.entries[at].a = (int32_t)out.entries.size();
            break;
        }
        case RT_H_MEDIUM: {
            const constant_medium *m = static_cast<const constant_medium *>(h);
            auto key = std::make_tuple(h, chain, 0x200u);
            auto it = prim_index.find(key);
            int idx;
            if (it != prim_index.end()) {
                idx = it->second;
            } else {
                int bchain = chain;
                unsigned bflip = 0;
                const hitable *b = unwrap(m->boundary, bchain, bflip);
                int bidx = add_simple(b, bchain, bflip, false);
                rt_prim p = blank(RT_PRIM_MEDIUM, chain, 0);
                p.first = bidx;
                p.f[0] = m->density;
                p.material = add_material(m->phase_function);
                p.user_id = next_user_id++;
                idx = (int)out.prims.size();
                out.prims.push_back(p);
                prim_index[key] = idx;
            }
            push_leaf(idx);
            break;
        }
        default:
            push_leaf(add_simple(h, chain, flip, true));
        }
    }

    void add_lights(const hitable *important)
    {
        if (!important || important->kind() != RT_H_LIST)
            throw unsupported{"important_hitables must be a hitable_list (src/main.cpp:107 dereferences it)"};
        const hitable_list *l = static_cast<const hitable_list *>(important);
        if (l->list_size < 1) throw unsupported{"important_hitables is empty"};
        for (int i = 0; i < l->list_size; ++i) {
            const hitable *h = l->list[i];
            // only xz_rect and sphere override pdf_value/random (src/plane.h:76-98, src/sphere.h:68-103)
            if (h && (h->kind() == RT_H_XZ_RECT || h->kind() == RT_H_SPHERE)) {
                rt_prim p = blank(0, 0, 0);
                fill_plain(h, p);
                out.lights.push_back((int)out.prims.size());
                out.prims.push_back(p);
            } else {
                out.lights.push_back(-1);
            }
        }
    }
};

} // namespace

const rt_scene_desc *rt_flat_scene::finish()
{
    memset(&desc, 0, sizeof desc);
    desc.struct_size = sizeof(rt_scene_desc);
    desc.abi_version = RT_ABI_VERSION;
    desc.entries = entries.data(), desc.n_entries = (int32_t)entries.size();
    desc.prims = prims.data(), desc.n_prims = (int32_t)prims.size();
    desc.xforms = xforms.data(), desc.n_xforms = (int32_t)xforms.size();
    desc.chains = chains.data(), desc.n_chains = (int32_t)chains.size();
    desc.materials = materials.data(), desc.n_materials = (int32_t)materials.size();
    desc.textures = textures.data(), desc.n_textures = (int32_t)textures.size();
    desc.texels = texels.data(), desc.n_texel_bytes = (int64_t)texels.size();
    desc.lights = lights.data(), desc.n_lights = (int32_t)lights.size();
    desc.camera = camera;
    return &desc;
}

int rt_flatten(const hitable *world, const hitable *important, const camera *cam, rt_flat_scene &out)
{
    out = rt_flat_scene();
    if (!world || !cam) {
        out.status = RT_ERR_INVALID;
        out.error = "rt_flatten: world and camera are required";
        return out.status;
    }
    try {
        flattener f(out);
        f.emit(world, 0, 0);
        f.add_lights(important);
    } catch (const unsupported &u) {
        out.status = RT_ERR_UNSUPPORTED;
        out.error = "rt_flatten: " + u.what;
        return out.status;
    }
    rt_camera &c = out.camera;
    for (int i = 0; i < 3; ++i) {
        c.origin[i] = cam->origin[i], c.u[i] = cam->u[i], c.v[i] = cam->v[i], c.w[i] = cam->w[i];
        c.lower_left_corner[i] = cam->lower_left_corner[i];
        c.horizontal[i] = cam->horizontal[i], c.vertical[i] = cam->vertical[i];
    }
    c.lens_radius = cam->lens_radius, c.time0 = cam->time0, c.time1 = cam->time1;
    out.finish();
    return RT_OK;
}
