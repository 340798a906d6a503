// rt_scenes.h -- scene builders with the reference's signature
//     void cornell_box( hitable **scene, hitable **important_hitables, camera **cam, float aspect )
// (src/scenes.h:163).  The numbers below ARE the scene: they are the Cornell-box
// data of src/scenes.h:165-197 and have to be identical for images to agree.
#ifndef RT_SCENES_H_
#define RT_SCENES_H_

#include "rt_scene_api.h"

inline void cornell_box(hitable **scene, hitable **important_hitables, camera **cam, float aspect)
{
    auto flat = [](float r, float g, float b) { return new constant_texture(vec3(r, g, b)); };
    material *red = new lambertian(flat(0.65f, 0.05f, 0.05f));
    material *white = new lambertian(flat(0.73f, 0.73f, 0.73f));
    material *green = new lambertian(flat(0.12f, 0.45f, 0.15f));
    material *lamp = new diffuse_light(flat(15, 15, 15));
    material *aluminium = new metal(flat(0.8f, 0.85f, 0.88f), 0.0f);
    material *glass = new dielectric(1.5f);

    hitable *ceiling_light = new xz_rect(213, 343, 227, 332, 554, lamp);
    hitable *glass_ball = new sphere(vec3(190.0f, 90.0f, 190.0f), 90.0f, glass);
    hitable *tall_block =
        new translate(new rotate_y(new box(vec3(0, 0, 0), vec3(165, 330, 165), aluminium), 15), vec3(265, 0, 295));

    // order matters: the BVH build sorts this array in place and its comparator never
    // reports equality, so the tree depends on the starting order (src/scenes.h:176-190)
    hitable **world = new hitable *[8]{
        new flip_normals(new yz_rect(0, 555, 0, 555, 555, green)), // left wall
        new yz_rect(0, 555, 0, 555, 0, red),                       // right wall
        new flip_normals(new xz_rect(0, 555, 0, 555, 555, white)), // ceiling
        ceiling_light,
        new xz_rect(0, 555, 0, 555, 0, white),                     // floor
        new flip_normals(new xy_rect(0, 555, 0, 555, 555, white)), // back wall
        glass_ball,
        tall_block,
    };
    hitable **sampled = new hitable *[2]{ceiling_light, glass_ball};

    *important_hitables = new hitable_list(sampled, 2);
    *scene = new hitable_list(world, 8);
    *cam = new camera(vec3(278.0f, 278.0f, -800.0f), vec3(278.0f, 278.0f, 278.0f), vec3(0.0f, 1.0f, 0.0f), 40.0f, aspect,
                      0.0f, 800.0f, 0.0f, 1.0f);
}

#endif // RT_SCENES_H_
