// rt_host.cpp -- C entry points of the host scene layer (include/rt_host.h)
#include "../../include/rt_host.h"

#include <cstring>
#include <sstream>

#include "../csrc/rt_error.h"
#include "rt_flatten.h"
#include "rt_scenes.h"

struct rt_host_scene {
    rt_flat_scene flat;
};

namespace {
typedef void (*scene_fn)(hitable **, hitable **, camera **, float);
struct named_scene {
    const char *name;
    scene_fn fn;
};
const named_scene k_scenes[] = {
    {"cornell_box", cornell_box},
};
uint32_t g_state_after_build = 0;
} // namespace

extern "C" const char *rt_host_scene_names(void)
{
    static std::string names;
    if (names.empty())
        for (const named_scene &s : k_scenes) names += (names.empty() ? "" : ",") + std::string(s.name);
    return names.c_str();
}

extern "C" uint32_t rt_host_ran01_state(void) { return g_state_after_build; }

extern "C" int rt_host_scene_build(const char *name, float aspect, rt_host_scene **out)
{
    if (!name || !out) return rt_fail(RT_ERR_INVALID, "rt_host_scene_build: null argument");
    *out = nullptr;
    scene_fn fn = nullptr;
    for (const named_scene &s : k_scenes)
        if (!strcmp(s.name, name)) fn = s.fn;
    if (!fn) return rt_fail(RT_ERR_INVALID, "rt_host_scene_build: unknown scene '%s' (have: %s)", name, rt_host_scene_names());
    if (!(aspect > 0.0f)) return rt_fail(RT_ERR_INVALID, "rt_host_scene_build: aspect must be positive");

    rt_host_rng_instance().reset(); // a fresh process, as far as RAN01() is concerned
    hitable *world = nullptr, *important = nullptr;
    camera *cam = nullptr;
    fn(&world, &important, &cam, aspect);
    // src/main.cpp:508-511: the BVH is always built over the list the scene function returns
    hitable_list *wl = static_cast<hitable_list *>(world);
    bvh_node *root = new bvh_node(wl->list, wl->list_size, 0.0f, 1.0f);
    {
        std::ostringstream os;
        os << rt_host_rng_instance().engine;
        g_state_after_build = (uint32_t)std::stoul(os.str());
    }
    rt_host_scene *s = new rt_host_scene;
    int st = rt_flatten(root, important, cam, s->flat);
    // the object tree is leaked on purpose, like the reference's (nothing there is ever freed);
    // scenes are built a handful of times per process
    if (st != RT_OK) {
        rt_fail(st, "%s", s->flat.error.c_str());
        delete s;
        return st;
    }
    *out = s;
    return RT_OK;
}

extern "C" const rt_scene_desc *rt_host_scene_desc(const rt_host_scene *scene) { return scene ? &scene->flat.desc : nullptr; }
extern "C" void rt_host_scene_free(rt_host_scene *scene) { delete scene; }
