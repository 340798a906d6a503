// rt_host.cpp -- C entry points of the host scene layer (include/rt_host.h)
#include "../../include/rt_host.h"

#include <cstdlib>
#include <cstring>
#include <sstream>

#include "../csrc/rt_error.h"
#include "rt_flatten.h"
#include "rt_scenes.h"
#include "rt_scenes_ext.h"

#include <dlfcn.h>

#include <vector>

// Texels of the earth texture (reference: stbi_load("../data/earth.jpg"), src/scenes.h:296).  JPEG decoding is host
// I/O outside the hot path; the product ships the decoded 1024x512 RGB pixels (ray_tracing_amd/data/, written by
// `oracle/_ref/ref_harness earth`, i.e. decoded by the reference's own stb_image) and reads them from beside the library.
static std::string g_earth_problem; // why rt_ext_earth_pixels returned nullptr (rt_host_scene_build reports it)
unsigned char *rt_ext_earth_pixels(int *nx, int *ny)
{
    static std::vector<unsigned char> px;
    if (px.empty()) {
        // RT_AMD_DATA_DIR names the directory that holds earth_1024x512.rgb; default: data/ beside the library
        std::string dir;
        if (const char *e = getenv("RT_AMD_DATA_DIR")) {
            dir = e;
        } else {
            Dl_info info;
            dir = ".";
            if (dladdr((void *)&rt_ext_earth_pixels, &info) && info.dli_fname) {
                dir = info.dli_fname;
                size_t slash = dir.find_last_of('/');
                dir = slash == std::string::npos ? "." : dir.substr(0, slash);
            }
            dir += "/data";
        }
        const std::string path = dir + "/earth_1024x512.rgb";
        if (FILE *f = fopen(path.c_str(), "rb")) {
            px.resize((size_t)3 * 1024 * 512);
            const size_t got = fread(px.data(), 1, px.size(), f);
            const bool more = got == px.size() && fgetc(f) != EOF;
            fclose(f);
            if (got != px.size() || more) {
                px.clear();
                g_earth_problem = path + " is not 1024 x 512 x 3 bytes (a text-mode checkout can do that: it is binary data)";
            }
        } else {
            g_earth_problem = "cannot open " + path + " (the decoded earth texture; set RT_AMD_DATA_DIR to the directory that holds it)";
        }
    }
    if (px.empty()) return nullptr; // rt_host_scene_build fails with g_earth_problem: no scene is built without its texture
    *nx = 1024, *ny = 512;
    return px.data();
}

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
    {"cornell_box_volumes", cornell_box_volumes},
    {"random_spheres", random_spheres},
    {"next_week_final", next_week_final},
    {"simple_scene", simple_scene},
    {"another_simple", another_simple},
    {"two_perlin_spheres", two_perlin_spheres},
    {"sphere_field_4k", sphere_field_4k},
    {"sphere_field_1m", sphere_field_1m},
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
    g_earth_problem.clear();
    fn(&world, &important, &cam, aspect);
    // a scene that asked for the earth texture and did not get it is not the scene of that name: fail, loudly
    if (!g_earth_problem.empty()) return rt_fail(RT_ERR_INVALID, "rt_host_scene_build('%s'): %s", name, g_earth_problem.c_str());
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
