// rt_scenes_ext.h -- the benchmark scenes that are NOT live in the reference.
//
// The reference ships exactly one scene that compiles, cornell_box (src/scenes.h:163-198).  Its other
// scene functions (src/scenes.h:35-85, 203-229, 234-314) are commented out and use an older API (they
// return one hitable*, have no important list and no camera).  BASELINE.json's configs 1 and 3 name
// those scenes, so they are re-stated here against the CURRENT signature
//     void f( hitable **scene, hitable **important_hitables, camera **cam, float aspect )
// using nothing but the API both header sets share.  THIS FILE IS COMPILED TWICE: against the reference's
// own headers inside oracle/_ref/ref_harness (to write golden vectors with the reference's code) and
// against rt_scene_api.h inside the product (to render on the GPU), so the two sides build the same
// objects from the same RAN01() stream.
//
// What had to be decided because the old functions do not say: every scene needs an important list made of
// xz_rect / sphere objects (only those implement pdf_value/random, src/hitable.h:19-20), a camera, and,
// for the book-1 scene, a light (the miss colour is black, src/main.cpp:125).  Each choice is marked DECISION.
#ifndef RT_SCENES_EXT_H_
#define RT_SCENES_EXT_H_

#include <cmath>
#include <cstring>

// provided by whoever includes this file: RGB pixels of data/earth.jpg (1024x512), or nullptr
unsigned char *rt_ext_earth_pixels(int *nx, int *ny);

// src/scenes.h:203-229 -- the Cornell box with two smoke boxes
inline void cornell_box_volumes(hitable **scene, hitable **important_hitables, camera **cam, float aspect)
{
    hitable **list = new hitable *[8];
    int i = 0;
    material *red = new lambertian(new constant_texture(vec3(0.65f, 0.05f, 0.05f)));
    material *white = new lambertian(new constant_texture(vec3(0.73f, 0.73f, 0.73f)));
    material *green = new lambertian(new constant_texture(vec3(0.12f, 0.45f, 0.15f)));
    material *light = new diffuse_light(new constant_texture(vec3(15, 15, 15)));
    list[i++] = new flip_normals(new yz_rect(0, 555, 0, 555, 555, green));
    list[i++] = new yz_rect(0, 555, 0, 555, 0, red);
    list[i++] = new flip_normals(new xz_rect(0, 555, 0, 555, 555, white));
    hitable *lamp = new xz_rect(213, 343, 227, 332, 554, light);
    list[i++] = lamp;
    list[i++] = new xz_rect(0, 555, 0, 555, 0, white);
    list[i++] = new flip_normals(new xy_rect(0, 555, 0, 555, 555, white));
    hitable *b1 = new translate(new rotate_y(new box(vec3(0, 0, 0), vec3(165, 165, 165), white), -18), vec3(130, 0, 65));
    hitable *b2 = new translate(new rotate_y(new box(vec3(0, 0, 0), vec3(165, 330, 165), white), 15), vec3(265, 0, 295));
    list[i++] = new constant_medium(b1, 0.02f, new constant_texture(vec3(1.0f, 1.0f, 1.0f)));
    list[i++] = new constant_medium(b2, 0.01f, new constant_texture(vec3(0.0f, 0.0f, 0.0f)));
    hitable **imp = new hitable *[1];
    imp[0] = lamp; // DECISION: the lamp is the important list
    *important_hitables = new hitable_list(imp, 1);
    *scene = new hitable_list(list, i);
    // DECISION: the camera of the live cornell_box (src/scenes.h:194-197)
    *cam = new camera(vec3(278.0f, 278.0f, -800.0f), vec3(278.0f, 278.0f, 278.0f), vec3(0.0f, 1.0f, 0.0f), 40.0f, aspect, 0.0f,
                      800.0f, 0.0f, 1.0f);
}

// src/scenes.h:35-85 -- "random spheres" (book 1 cover), n = 500: BASELINE.json config 1
inline void random_spheres(hitable **scene, hitable **important_hitables, camera **cam, float aspect)
{
    int n = 500;
    int surf_radius = ((int)sqrtf((float)n)) / 2 - 1;
    hitable **list = new hitable *[n + 3];
    list[0] = new sphere(vec3(0, -1000, 0), 1000, new lambertian(new constant_texture(vec3(0.5f, 0.5f, 0.5f))));
    int i = 1;
    for (int a = -surf_radius; a < surf_radius; ++a) {
        for (int b = -surf_radius; b < surf_radius; ++b) {
            // one draw per statement: the reference's nested constructor calls leave the order to the compiler
            float choose_mat = RAN01();
            float cx = a + 0.9f * RAN01();
            float cz = b + 0.9f * RAN01();
            vec3 center(cx, 0.2f, cz);
            if ((center - vec3(4.0f, 0.2f, 0.0f)).length() > 0.9f) {
                if (choose_mat < 0.8f) {
                    float lift = 0.5f * RAN01();
                    float r0 = RAN01() * RAN01();
                    float g0 = RAN01() * RAN01();
                    float b0 = RAN01() * RAN01();
                    list[i++] = new moving_sphere(center, center + vec3(0.0f, lift, 0.0f), 0.0f, 1.0f, 0.2f,
                                                  new lambertian(new constant_texture(vec3(r0, g0, b0))));
                } else if (choose_mat < 0.95f) {
                    float r0 = 0.5f * (1.0f + RAN01());
                    float g0 = 0.5f * (1.0f + RAN01());
                    float b0 = 0.5f * (1.0f + RAN01());
                    float fz = 0.5f * RAN01();
                    list[i++] = new sphere(center, 0.2f, new metal(new constant_texture(vec3(r0, g0, b0)), fz));
                } else {
                    list[i++] = new sphere(center, 0.2f, new dielectric(1.5f));
                }
            }
        }
    }
    list[i++] = new sphere(vec3(0.0f, 1.0f, 0.0f), 1.0f, new dielectric(1.5f));
    list[i++] = new sphere(vec3(-4.0f, 1.0f, 0.0f), 1.0f, new lambertian(new constant_texture(vec3(0.4f, 0.2f, 0.1f))));
    list[i++] = new sphere(vec3(4.0f, 1.0f, 0.0f), 1.0f, new metal(new constant_texture(vec3(0.7f, 0.6f, 0.5f)), 0.0f));
    // DECISION: the old function has no emitter and a miss is black (src/main.cpp:125).  Light = one panel
    // above the scene, in the style of the reference's rect lamps; it is the important list.
    hitable *panel = new xz_rect(-6, 6, -6, 6, 12, new diffuse_light(new constant_texture(vec3(6, 6, 6))));
    list[i++] = panel;
    hitable **imp = new hitable *[1];
    imp[0] = panel;
    *important_hitables = new hitable_list(imp, 1);
    *scene = new hitable_list(list, i);
    // DECISION: the book's camera for this scene (the reference keeps none), aperture 0.1 so the lens is exercised
    *cam = new camera(vec3(13.0f, 2.0f, 3.0f), vec3(0.0f, 0.0f, 0.0f), vec3(0.0f, 1.0f, 0.0f), 20.0f, aspect, 0.1f, 10.0f, 0.0f,
                      1.0f);
}

// src/scenes.h:234-314 -- "The Next Week" final scene: BASELINE.json config 3
inline void next_week_final(hitable **scene, hitable **important_hitables, camera **cam, float aspect)
{
    hitable **list = new hitable *[30];
    hitable **boxlist = new hitable *[400];
    hitable **boxlist2 = new hitable *[1000];
    material *white = new lambertian(new constant_texture(vec3(0.73f, 0.73f, 0.73f)));
    material *ground = new lambertian(new constant_texture(vec3(0.48f, 0.83f, 0.53f)));
    int nb = 20, b = 0;
    for (int i = 0; i < nb; ++i) {
        for (int j = 0; j < nb; ++j) {
            float w = 100.0f;
            float x0 = -1000.0f + i * w, z0 = -1000.0f + j * w, y0 = 0.0f;
            float x1 = x0 + w, y1 = 100.0f * (RAN01() + 0.01f), z1 = z0 + w;
            boxlist[b++] = new box(vec3(x0, y0, z0), vec3(x1, y1, z1), ground);
        }
    }
    int l = 0;
    list[l++] = new bvh_node(boxlist, b, 0.0f, 1.0f);
    material *light = new diffuse_light(new constant_texture(vec3(7, 7, 7)));
    hitable *lamp = new xz_rect(123, 423, 147, 412, 554, light);
    list[l++] = lamp;
    vec3 center(400, 400, 200);
    list[l++] = new moving_sphere(center, center + vec3(30, 0, 0), 0, 1, 50,
                                  new lambertian(new constant_texture(vec3(0.7f, 0.3f, 0.1f))));
    hitable *glass_ball = new sphere(vec3(260, 150, 45), 50, new dielectric(1.5f));
    list[l++] = glass_ball;
    list[l++] = new sphere(vec3(0, 150, 145), 50, new metal(new constant_texture(vec3(0.8f, 0.8f, 0.9f)), 10.0f));
    hitable *boundary = new sphere(vec3(360, 150, 145), 70, new dielectric(1.5f));
    list[l++] = boundary;
    list[l++] = new constant_medium(boundary, 0.2f, new constant_texture(vec3(0.2f, 0.4f, 0.9f)));
    boundary = new sphere(vec3(0, 0, 0), 5000, new dielectric(1.5f));
    list[l++] = new constant_medium(boundary, 0.0001f, new constant_texture(vec3(1.0f, 1.0f, 1.0f)));
    int nx = 0, ny = 0;
    unsigned char *tex_data = rt_ext_earth_pixels(&nx, &ny);
    if (tex_data) list[l++] = new sphere(vec3(400, 200, 400), 100, new lambertian(new image_texture(tex_data, nx, ny)));
    list[l++] = new sphere(vec3(220, 280, 300), 80, new lambertian(new noise_texture(0.1f)));
    int ns = 1000;
    for (int j = 0; j < ns; ++j) {
        float z = 165.0f * RAN01(), y = 165.0f * RAN01(), x = 165.0f * RAN01(); // last argument first, as g++ evaluates the original
        boxlist2[j] = new sphere(vec3(x, y, z), 10, white);
    }
    list[l++] = new translate(new rotate_y(new bvh_node(boxlist2, ns, 0.0f, 1.0f), 15.0f), vec3(-100.0f, 270.0f, 395.0f));
    hitable **imp = new hitable *[2];
    imp[0] = lamp, imp[1] = glass_ball; // DECISION: lamp + glass ball, like the live cornell_box samples its lamp and its glass sphere
    *important_hitables = new hitable_list(imp, 2);
    *scene = new hitable_list(list, l);
    // the camera the reference keeps in a comment for this scene (src/scenes.h:14)
    *cam = new camera(vec3(350.0f, 278.0f, -450.0f), vec3(180.0f, 278.0f, 278.0f), vec3(0.0f, 1.0f, 0.0f), 45.0f, aspect, 0.0f,
                      800.0f, 0.0f, 1.0f);
}

// src/scenes.h:90-107 -- "simple scene": glass, earth-textured and metal spheres under a sphere light and a small rect light
inline void simple_scene(hitable **scene, hitable **important_hitables, camera **cam, float aspect)
{
    int nx = 0, ny = 0;
    unsigned char *tex_data = rt_ext_earth_pixels(&nx, &ny);
    hitable **list = new hitable *[10];
    int i = 0;
    // DECISION: the ground carries the checker texture the reference keeps in a comment two lines below (src/scenes.h:101);
    // no other scene of the reference uses checker_texture
    list[i++] = new sphere(vec3(0, -1000, 0), 1000,
                           new lambertian(new checker_texture(new constant_texture(vec3(0.4f, 0.2f, 0.1f)),
                                                              new constant_texture(vec3(0.9f, 0.9f, 0.9f)))));
    list[i++] = new sphere(vec3(4.0f, 1.0f, 0.0f), 1.0f, new dielectric(1.5f));
    if (tex_data) list[i++] = new sphere(vec3(0.0f, 1.0f, 0.0f), 1.0f, new lambertian(new image_texture(tex_data, nx, ny)));
    list[i++] = new sphere(vec3(-4.0f, 1.0f, 0.0f), 1.0f, new metal(new constant_texture(vec3(0.7f, 0.6f, 0.5f)), 0.0f));
    hitable *lamp = new sphere(vec3(0, 6, 0), 3, new diffuse_light(new constant_texture(vec3(4, 4, 3))));
    list[i++] = lamp;
    list[i++] = new xy_rect(3, 5, 1, 3, -2, new diffuse_light(new constant_texture(vec3(4, 4, 4))));
    hitable **imp = new hitable *[1];
    imp[0] = lamp; // DECISION: the sphere lamp (an xy_rect has no pdf_value/random, src/plane.h:9-29)
    *important_hitables = new hitable_list(imp, 1);
    *scene = new hitable_list(list, i);
    // DECISION: the book's camera for its sphere scenes
    *cam = new camera(vec3(13.0f, 2.0f, 3.0f), vec3(0.0f, 1.0f, 0.0f), vec3(0.0f, 1.0f, 0.0f), 30.0f, aspect, 0.0f, 10.0f, 0.0f, 1.0f);
}

// src/scenes.h:112-146 -- "another simple": 25 glass spheres filled with media of rising density and index, under a white sky
inline void another_simple(hitable **scene, hitable **important_hitables, camera **cam, float aspect)
{
    int l = 0;
    hitable **list = new hitable *[100];
    list[l++] = new flip_normals(new sphere(vec3(0, 0, 0), 50, new diffuse_light(new constant_texture(vec3(1, 1, 1)))));
    list[l++] = new sphere(vec3(0, -1000, 0), 1000, new lambertian(new constant_texture(vec3(0.5f, 0.5f, 0.5f))));
    const float sphere_height = 17.0f;
    const float sphere_radius = 1.0f;
    const float diameter = 2.0f * sphere_radius;
    const int nb = 5;
    const float half_scene_size = (nb * diameter) / 2.0f;
    hitable *middle = nullptr;
    for (int i = 0; i < nb; ++i) {
        for (int j = 0; j < nb; ++j) {
            vec3 position(i * diameter - half_scene_size + sphere_radius, sphere_height, j * diameter - half_scene_size + sphere_radius);
            float density = (nb - j) * 1.0f;
            float ior = 1.0f + (i + 1) * 0.1f;
            hitable *boundary = new sphere(position, 1.0f, new dielectric(ior));
            if (i == nb / 2 && j == nb / 2) middle = boundary;
            list[l++] = boundary;
            list[l++] = new constant_medium(boundary, density, new constant_texture(vec3(0.2f, 0.4f, 0.9f)));
        }
    }
    hitable **imp = new hitable *[1];
    imp[0] = middle; // DECISION: the sky is a flip_normals wrapper (no pdf_value/random, src/hitable.h:27-38); the list
                     // samples the middle glass sphere, as the live cornell_box samples its glass sphere
    *important_hitables = new hitable_list(imp, 1);
    *scene = new hitable_list(list, l);
    // DECISION: a camera below the grid, looking up at it
    *cam = new camera(vec3(0.0f, 3.0f, -18.0f), vec3(0.0f, 15.0f, 0.0f), vec3(0.0f, 1.0f, 0.0f), 40.0f, aspect, 0.0f, 10.0f, 0.0f, 1.0f);
}

// src/scenes.h:151-160 -- two spheres with the perlin texture
inline void two_perlin_spheres(hitable **scene, hitable **important_hitables, camera **cam, float aspect)
{
    texture *pertex = new noise_texture(2.0f);
    hitable **list = new hitable *[3];
    list[0] = new sphere(vec3(0, -1000, 0), 1000, new lambertian(pertex));
    list[1] = new sphere(vec3(0.0f, 2.0f, 0.0f), 2.0f, new lambertian(pertex));
    // DECISION: the old function has no emitter and a miss is black (src/main.cpp:125): a sphere lamp, the important list
    hitable *lamp = new sphere(vec3(0, 7, 0), 2, new diffuse_light(new constant_texture(vec3(4, 4, 4))));
    list[2] = lamp;
    hitable **imp = new hitable *[1];
    imp[0] = lamp;
    *important_hitables = new hitable_list(imp, 1);
    *scene = new hitable_list(list, 3);
    *cam = new camera(vec3(13.0f, 2.0f, 3.0f), vec3(0.0f, 2.0f, 0.0f), vec3(0.0f, 1.0f, 0.0f), 30.0f, aspect, 0.0f, 10.0f, 0.0f, 1.0f);
}

// BASELINE.json config 5: "procedural 1M-sphere scene (deep BVH, incoherent secondary rays)".  The reference holds no
// such scene; this one is written in its style so that the reference's own code can render it too (the harness does, at
// 4096 spheres, for the golden vectors).  DECISION, all of it: n spheres of radius 0.25 with centres uniform in a cube of
// half-side 0.5*cbrt(n) (6.5 % of the volume is sphere at any n), 70 % lambertian / 20 % metal / 10 % glass, one light
// panel above the cube (the important list), camera outside a corner.  One bvh_node over everything, as main() builds it.
inline void sphere_field(int n, hitable **scene, hitable **important_hitables, camera **cam, float aspect)
{
    const float half = 0.5f * cbrtf((float)n);
    hitable **list = new hitable *[n + 1];
    for (int i = 0; i < n; ++i) {
        float x = half * (2.0f * RAN01() - 1.0f);
        float y = half * (2.0f * RAN01() - 1.0f);
        float z = half * (2.0f * RAN01() - 1.0f);
        float choose_mat = RAN01();
        material *m;
        if (choose_mat < 0.7f) {
            float r0 = RAN01() * RAN01();
            float g0 = RAN01() * RAN01();
            float b0 = RAN01() * RAN01();
            m = new lambertian(new constant_texture(vec3(0.1f + 0.9f * r0, 0.1f + 0.9f * g0, 0.1f + 0.9f * b0)));
        } else if (choose_mat < 0.9f) {
            float r0 = 0.5f * (1.0f + RAN01());
            float g0 = 0.5f * (1.0f + RAN01());
            float b0 = 0.5f * (1.0f + RAN01());
            float fz = 0.3f * RAN01();
            m = new metal(new constant_texture(vec3(r0, g0, b0)), fz);
        } else {
            m = new dielectric(1.5f);
        }
        list[i] = new sphere(vec3(x, y, z), 0.25f, m);
    }
    hitable *panel = new xz_rect(-half, half, -half, half, 1.5f * half, new diffuse_light(new constant_texture(vec3(5, 5, 5))));
    list[n] = panel;
    hitable **imp = new hitable *[1];
    imp[0] = panel;
    *important_hitables = new hitable_list(imp, 1);
    *scene = new hitable_list(list, n + 1);
    *cam = new camera(vec3(2.6f * half, 1.2f * half, 2.6f * half), vec3(0.0f, 0.0f, 0.0f), vec3(0.0f, 1.0f, 0.0f), 35.0f, aspect, 0.0f,
                      10.0f, 0.0f, 1.0f);
}
inline void sphere_field_4k(hitable **scene, hitable **important_hitables, camera **cam, float aspect)
{
    sphere_field(4096, scene, important_hitables, cam, aspect);
}
inline void sphere_field_1m(hitable **scene, hitable **important_hitables, camera **cam, float aspect)
{
    sphere_field(1000000, scene, important_hitables, cam, aspect);
}

typedef void (*rt_ext_scene_fn)(hitable **, hitable **, camera **, float);
inline rt_ext_scene_fn rt_extra_scene(const char *name)
{
    if (!strcmp(name, "cornell_box_volumes")) return cornell_box_volumes;
    if (!strcmp(name, "random_spheres")) return random_spheres;
    if (!strcmp(name, "next_week_final")) return next_week_final;
    if (!strcmp(name, "simple_scene")) return simple_scene;
    if (!strcmp(name, "another_simple")) return another_simple;
    if (!strcmp(name, "two_perlin_spheres")) return two_perlin_spheres;
    if (!strcmp(name, "sphere_field_4k")) return sphere_field_4k;
    if (!strcmp(name, "sphere_field_1m")) return sphere_field_1m;
    return nullptr;
}

#endif // RT_SCENES_EXT_H_
