// rt_flatten.h -- walks a finished host scene tree (rt_scene_api.h) once and
// emits the flat arrays of include/rt_amd.h.
//
// This is the "thin C-ABI flattener" between the kept scene API and the GPU.
// The walk is the depth-first, left-then-right order in which the reference's
// bvh_node::hit recursion (src/bvh.h:123-163) meets the tree, so one linear
// pass over `entries` on the device visits boxes and primitives in exactly
// the reference's order (that order matters: ties go to the later primitive,
// and constant_medium::hit draws random numbers as it is visited).
//
// Supported compositions (anything else fails with RT_ERR_UNSUPPORTED and a
// message, never silently):
//   world     := bvh_node tree | hitable_list of items | item
//   item      := wrappers* ( sphere | moving_sphere | *_rect | box
//                          | hitable_list of (flip_normals? rect|sphere|moving_sphere)
//                          | bvh_node tree | constant_medium(boundary) )
//   wrappers  := flip_normals | translate | rotate_y          (any order, any depth)
//   boundary  := wrappers* ( sphere | moving_sphere | *_rect | box | flat hitable_list )
// A hitable_list at the root is flattened as one GROUP primitive when all its
// members are plain primitives; main.cpp always wraps the world in a bvh_node
// (src/main.cpp:508-511), which is the normal case.
#ifndef RT_FLATTEN_H_
#define RT_FLATTEN_H_

#include <map>
#include <string>
#include <vector>

#include "../../include/rt_amd.h"
#include "rt_scene_api.h"

struct rt_flat_scene {
    std::vector<rt_entry> entries;
    std::vector<rt_prim> prims;
    std::vector<rt_xform> xforms;
    std::vector<rt_chain> chains;
    std::vector<rt_material> materials;
    std::vector<rt_texture> textures;
    std::vector<uint8_t> texels;
    std::vector<int32_t> lights;
    rt_camera camera;
    rt_scene_desc desc; // points into the vectors above; valid until they change
    std::string error;
    int status = RT_OK;

    const rt_scene_desc *finish();
};

// world: root of the scene (normally the bvh_node built over the world list);
// important: the hitable_list handed to color() as important_hitables
// (src/main.cpp:107), may be null; cam: the camera.
int rt_flatten(const hitable *world, const hitable *important, const camera *cam, rt_flat_scene &out);

#endif // RT_FLATTEN_H_
