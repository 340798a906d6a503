// records_check.cpp -- host-side checks of ray_tracing_amd/csrc/rt_records.cpp (built and run by tests/test_host_trees.py):
// the record table of a small scene, its box records and its runs.
#include "../../ray_tracing_amd/csrc/rt_records.h"
#include "../../include/rt_host.h"

#include <cstdio>
#include <cstring>
#include <set>

static float f(const uint32_t *r, int k)
{
    float v;
    memcpy(&v, &r[k], 4);
    return v;
}

int main(int argc, char **argv)
{
    int failures = 0;
    for (int a = 1; a < argc; ++a) {
        rt_host_scene *h = nullptr;
        if (rt_host_scene_build(argv[a], 1.0f, &h) != RT_OK) {
            printf("%s: build failed: %s\n", argv[a], rt_last_error());
            ++failures;
            continue;
        }
        const rt_scene_desc *d = rt_host_scene_desc(h);
        rt_record_table t;
        const char *why = "";
        const int fits = rt_records_build(d, 0, &t, &why);
        int bad = 0, boxes = 0, rects = 0, spheres = 0;
        if (fits) {
            std::set<int> ids;
            for (int i = 0; i < t.n_stream; ++i) {
                const uint32_t kind = t.rec[i][7] & 15u;
                if (kind == RT_REC_BOX) {
                    ++boxes;
                    // its second half follows, p0 < p1, the six faces are six different primitives of the scene
                    if (i + 1 >= t.n_stream || (t.rec[i + 1][7] & 15u) != RT_REC_BOX_AUX) ++bad;
                    for (int k = 0; k < 3; ++k)
                        if (!(f(t.rec[i], k) < f(t.rec[i], 4 + k))) ++bad;
                    for (int k = 0; k < 6; ++k) {
                        if (t.face_user[i][k] < 0 || (!ids.insert(t.face_user[i][k]).second && t.flat)) ++bad;
                    }
                    // the image of the x axis has unit length
                    const float ax = f(t.rec[i + 1], 3), az = f(t.rec[i + 1], 4);
                    if (!(ax * ax + az * az > 0.9999f && ax * ax + az * az < 1.0001f)) ++bad;
                    if (t.user_id[i] != -1) ++bad;
                } else if (kind == RT_REC_BOX_AUX) {
                    if (i == 0 || (t.rec[i - 1][7] & 15u) != RT_REC_BOX) ++bad;
                } else if (kind == RT_REC_RECT) {
                    ++rects;
                    if (t.user_id[i] < 0 || (!ids.insert(t.user_id[i]).second && t.flat)) ++bad;
                } else if (kind == RT_REC_SPHERE) {
                    ++spheres; // (a flat stream holds every primitive once; a stream with inner records follows the reference's tree)
                    if (t.user_id[i] < 0 || (!ids.insert(t.user_id[i]).second && t.flat)) ++bad;
                } else if (kind != RT_REC_INNER) {
                    ++bad; // nothing else belongs in the stream
                }
            }
            if (t.flat && t.n_runs > 0) { // the runs cover the stream: rects by axis, then boxes, chain by chain; spheres last
                int covered = 0, at = 0;
                for (int r = 0; r < t.n_runs; ++r) {
                    const uint32_t w = t.run[r];
                    const int n[4] = {(int)((w >> 10) & 31u), (int)((w >> 5) & 31u), (int)(w & 31u), (int)(w >> 24)};
                    for (int axis = 0; axis < 3; ++axis)
                        for (int k = 0; k < n[axis]; ++k, ++at)
                            if ((t.rec[at][7] & 15u) != RT_REC_RECT || ((t.rec[at][7] >> RT_KW_AXIS_SHIFT) & 3u) != (uint32_t)axis ||
                                ((t.rec[at][7] >> RT_KW_CHAIN_SHIFT) & RT_KW_CHAIN_MASK) != ((w >> 15) & RT_KW_CHAIN_MASK))
                                ++bad;
                    for (int k = 0; k < n[3]; ++k, at += 2)
                        if ((t.rec[at][7] & 15u) != RT_REC_BOX || ((t.rec[at][7] >> RT_KW_CHAIN_SHIFT) & RT_KW_CHAIN_MASK) != ((w >> 15) & RT_KW_CHAIN_MASK))
                            ++bad;
                    covered += n[0] + n[1] + n[2] + 2 * n[3];
                }
                for (int k = 0; k < t.n_spheres; ++k, ++at)
                    if ((t.rec[at][7] & 15u) != RT_REC_SPHERE) ++bad;
                if (covered + t.n_spheres != t.n_stream || at != t.n_stream) ++bad;
            }
            if (t.n_records > RT_REC_MAX || t.lights_base + t.n_lights != t.n_records) ++bad;
        }
        printf("%s: fits=%d (%s) stream=%d records=%d flat=%d runs=%d rects=%d boxes=%d spheres=%d bad=%d\n", argv[a], fits, why,
               fits ? t.n_stream : 0, fits ? t.n_records : 0, fits ? t.flat : 0, fits ? t.n_runs : 0, rects, boxes, spheres, bad);
        failures += bad;
        rt_host_scene_free(h);
    }
    printf("failures: %d\n", failures);
    return failures ? 1 : 0;
}
