/* rt_host.h -- C entry points of the host scene layer (ray_tracing_amd/host).
 *
 * C++ callers use the scene API directly (rt_scene_api.h, rt_scenes.h,
 * rt_flatten.h).  These wrappers exist so that non-C++ hosts (the Python test
 * and benchmark harness, via ctypes) can ask for one of the built-in scenes by
 * name and receive the flattened description that rt_scene_create() takes.
 * They stand where `cornell_box(&world,&important,&cam,aspect); new
 * bvh_node(list,n,time0,time1)` stands in the reference (src/main.cpp:501-511).
 */
#ifndef RT_HOST_H_
#define RT_HOST_H_

#include "rt_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_host_scene rt_host_scene;

/* Builds scene `name` ("cornell_box", ...; rt_host_scene_names() lists them)
 * for an image of the given aspect ratio: restarts the RAN01() stream at the
 * position the reference has when main() begins, runs the scene function,
 * builds the bvh_node over the world list exactly as src/main.cpp:508-511 does,
 * and flattens.  Returns RT_OK or a negative rt_status (see rt_last_error()). */
int rt_host_scene_build(const char* name, float aspect, rt_host_scene** out);
const rt_scene_desc* rt_host_scene_desc(const rt_host_scene* scene);
void rt_host_scene_free(rt_host_scene* scene);

/* Comma-separated list of built-in scene names. */
const char* rt_host_scene_names(void);

/* State of the minstd engine behind RAN01() after the last scene build: the
 * stream position at which the reference's first render task starts drawing
 * (src/main.cpp:62).  Test infrastructure uses it to continue that stream. */
uint32_t rt_host_ran01_state(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_HOST_H_ */
