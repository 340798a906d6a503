// rt_scene_api.h -- host-side scene construction API of the MI355X path tracer.
//
// Scene-building code written against droune2001/ray_tracing's headers
// (scenes.h, hitable.h, hitable_list.h, bvh.h, sphere.h, plane.h, box.h,
// transforms.h, volume.h, material.h, texture.h, camera.h) compiles unchanged
// against this one header: same type names, constructor signatures, public
// members, PI and RAN01().  What differs is what the objects are FOR.  In the
// reference every object intersects and shades on the CPU through virtual
// hit()/scatter()/value() calls; here the objects only describe the scene.
// They know their bounding boxes (the BVH builder needs them) and their
// parameters; rt_flatten() (rt_flatten.h) walks the finished tree once and
// emits the flat arrays of include/rt_amd.h, and the GPU does the rest.
// There is deliberately no hit()/scatter()/value() here: no CPU render path
// exists in the product.
//
// Everything that decides the SHAPE of the flattened scene follows the
// reference bit for bit, so that the GPU walks the very tree the reference
// would: bounding boxes (each class cites its source), the random-axis /
// qsort / median-split BVH build (src/bvh.h:79-121), the camera frame
// (src/camera.h:29-52), and the global minstd stream behind RAN01()
// (src/main.cpp:62-64), including the 1536 draws the reference's static
// initialisers consume before main() (src/perlin.h:145-185).
#ifndef RT_SCENE_API_H_
#define RT_SCENE_API_H_

#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>

#ifndef PI
#define PI 3.14159f // src/main.cpp:54-56 -- NOT M_PI; the value is part of the images
#endif

// ---------------------------------------------------------------------------
// RAN01(): one process-wide engine, as in the reference.  std::default_random_engine
// is libstdc++'s minstd_rand0; the discard() reproduces the stream position the
// reference is at when main() starts.
// ---------------------------------------------------------------------------
struct rt_host_rng {
    std::default_random_engine engine;
    std::uniform_real_distribution<float> dist{0.0f, 1.0f};
    rt_host_rng() { reset(); }
    void reset()
    {
        engine = std::default_random_engine();
        dist.reset();
        engine.discard(1536);
    }
    float next() { return dist(engine); }
};
inline rt_host_rng &rt_host_rng_instance()
{
    static rt_host_rng g;
    return g;
}
#define RAN01() (rt_host_rng_instance().next())

// ---------------------------------------------------------------------------
// vec3 / ray / aabb
// ---------------------------------------------------------------------------
struct vec3 {
    float e[3];
    vec3() : e{0.0f, 0.0f, 0.0f} {}
    vec3(float a, float b, float c) : e{a, b, c} {}
    float x() const { return e[0]; }
    float y() const { return e[1]; }
    float z() const { return e[2]; }
    float r() const { return e[0]; }
    float g() const { return e[1]; }
    float b() const { return e[2]; }
    float operator[](int i) const { return e[i]; }
    float &operator[](int i) { return e[i]; }
    vec3 operator-() const { return vec3(-e[0], -e[1], -e[2]); }
    const vec3 &operator+() const { return *this; }
    float squared_length() const { return e[0] * e[0] + e[1] * e[1] + e[2] * e[2]; }
    float length() const { return sqrtf(squared_length()); }
    vec3 &operator+=(const vec3 &o) { for (int i = 0; i < 3; ++i) e[i] += o.e[i]; return *this; }
    vec3 &operator-=(const vec3 &o) { for (int i = 0; i < 3; ++i) e[i] -= o.e[i]; return *this; }
    vec3 &operator*=(const vec3 &o) { for (int i = 0; i < 3; ++i) e[i] *= o.e[i]; return *this; }
    vec3 &operator/=(const vec3 &o) { for (int i = 0; i < 3; ++i) e[i] /= o.e[i]; return *this; }
    vec3 &operator*=(float t) { for (int i = 0; i < 3; ++i) e[i] *= t; return *this; }
    vec3 &operator/=(float t) { for (int i = 0; i < 3; ++i) e[i] /= t; return *this; }
};
inline vec3 operator+(const vec3 &a, const vec3 &b) { return vec3(a.e[0] + b.e[0], a.e[1] + b.e[1], a.e[2] + b.e[2]); }
inline vec3 operator-(const vec3 &a, const vec3 &b) { return vec3(a.e[0] - b.e[0], a.e[1] - b.e[1], a.e[2] - b.e[2]); }
inline vec3 operator*(const vec3 &a, const vec3 &b) { return vec3(a.e[0] * b.e[0], a.e[1] * b.e[1], a.e[2] * b.e[2]); }
inline vec3 operator/(const vec3 &a, const vec3 &b) { return vec3(a.e[0] / b.e[0], a.e[1] / b.e[1], a.e[2] / b.e[2]); }
inline vec3 operator*(const vec3 &a, float t) { return vec3(t * a.e[0], t * a.e[1], t * a.e[2]); }
inline vec3 operator*(float t, const vec3 &a) { return vec3(t * a.e[0], t * a.e[1], t * a.e[2]); }
inline vec3 operator/(const vec3 &a, float t) { return vec3(a.e[0] / t, a.e[1] / t, a.e[2] / t); }
inline float dot(const vec3 &a, const vec3 &b) { return a.e[0] * b.e[0] + a.e[1] * b.e[1] + a.e[2] * b.e[2]; }
inline vec3 cross(const vec3 &a, const vec3 &b)
{
    return vec3(a.e[1] * b.e[2] - a.e[2] * b.e[1], -(a.e[0] * b.e[2] - a.e[2] * b.e[0]),
                a.e[0] * b.e[1] - a.e[1] * b.e[0]);
}
inline vec3 unit_vector(vec3 v) { return v / v.length(); }

struct ray {
    vec3 _origin, _direction;
    float _time = 0.0f;
    ray() {}
    ray(const vec3 &o, const vec3 &d, float t = 0.0f) : _origin(o), _direction(d), _time(t) {}
    vec3 origin() const { return _origin; }
    vec3 direction() const { return _direction; }
    float time() const { return _time; }
};

struct aabb { // src/aabb.h:7-19 (defaults included)
    vec3 _min = vec3(FLT_MIN, FLT_MIN, FLT_MIN);
    vec3 _max = vec3(FLT_MAX, FLT_MAX, FLT_MAX);
    aabb() {}
    aabb(const vec3 &a, const vec3 &b) : _min(a), _max(b) {}
    vec3 min() const { return _min; }
    vec3 max() const { return _max; }
};
inline aabb surrounding_box(aabb p, aabb q) // src/aabb.h:21-32
{
    vec3 lo(fminf(p._min.e[0], q._min.e[0]), fminf(p._min.e[1], q._min.e[1]), fminf(p._min.e[2], q._min.e[2]));
    vec3 hi(fmaxf(p._max.e[0], q._max.e[0]), fmaxf(p._max.e[1], q._max.e[1]), fmaxf(p._max.e[2], q._max.e[2]));
    return aabb(lo, hi);
}

// ---------------------------------------------------------------------------
// Tags: the reference is built without RTTI (build.bat:7) and its classes carry
// no type field, so the flattener could not walk its tree.  Each class here
// reports what it is.
// ---------------------------------------------------------------------------
enum rt_host_kind {
    RT_H_SPHERE, RT_H_MOVING_SPHERE, RT_H_XY_RECT, RT_H_XZ_RECT, RT_H_YZ_RECT, RT_H_FLIP,
    RT_H_TRANSLATE, RT_H_ROTATE_Y, RT_H_BOX, RT_H_LIST, RT_H_BVH, RT_H_MEDIUM
};
enum rt_host_tex_kind { RT_HT_CONSTANT, RT_HT_CHECKER, RT_HT_NOISE, RT_HT_IMAGE };
enum rt_host_mat_kind { RT_HM_LAMBERTIAN, RT_HM_METAL, RT_HM_DIELECTRIC, RT_HM_LIGHT, RT_HM_ISOTROPIC };

// ---------------------------------------------------------------------------
// textures (src/texture.h) and materials (src/material.h): parameters only
// ---------------------------------------------------------------------------
struct texture {
    virtual ~texture() {}
    virtual rt_host_tex_kind tex_kind() const = 0;
};
struct constant_texture : texture {
    vec3 color;
    constant_texture() {}
    constant_texture(const vec3 &c) : color(c) {}
    rt_host_tex_kind tex_kind() const override { return RT_HT_CONSTANT; }
};
struct checker_texture : texture {
    texture *odd = nullptr, *even = nullptr;
    checker_texture() {}
    checker_texture(texture *t0, texture *t1) : odd(t0), even(t1) {}
    rt_host_tex_kind tex_kind() const override { return RT_HT_CHECKER; }
};
struct noise_texture : texture {
    float scale = 1.0f;
    noise_texture() {}
    noise_texture(float sc) : scale(sc) {}
    rt_host_tex_kind tex_kind() const override { return RT_HT_NOISE; }
};
struct image_texture : texture {
    unsigned char *data = nullptr;
    int nx = 0, ny = 0;
    image_texture() {}
    image_texture(unsigned char *pixels, int A, int B) : data(pixels), nx(A), ny(B) {}
    rt_host_tex_kind tex_kind() const override { return RT_HT_IMAGE; }
};

struct material {
    virtual ~material() {}
    virtual rt_host_mat_kind mat_kind() const = 0;
};
struct lambertian : material {
    texture *albedo = nullptr;
    lambertian(texture *a) : albedo(a) {}
    rt_host_mat_kind mat_kind() const override { return RT_HM_LAMBERTIAN; }
};
struct metal : material {
    texture *albedo = nullptr;
    float fuzz = 1.0f;
    metal(texture *a, float f) : albedo(a), fuzz(f < 1.0f ? f : 1.0f) {} // clamp: src/material.h:48-58
    rt_host_mat_kind mat_kind() const override { return RT_HM_METAL; }
};
struct dielectric : material {
    float ref_idx = 1.0f;
    dielectric(float ri) : ref_idx(ri) {}
    rt_host_mat_kind mat_kind() const override { return RT_HM_DIELECTRIC; }
};
struct diffuse_light : material {
    texture *emit = nullptr;
    diffuse_light(texture *E) : emit(E) {}
    rt_host_mat_kind mat_kind() const override { return RT_HM_LIGHT; }
};
struct isotropic : material {
    texture *albedo = nullptr;
    isotropic(texture *a) : albedo(a) {}
    rt_host_mat_kind mat_kind() const override { return RT_HM_ISOTROPIC; }
};

// ---------------------------------------------------------------------------
// geometry
// ---------------------------------------------------------------------------
struct hitable {
    virtual ~hitable() {}
    virtual bool bounding_box(float t0, float t1, aabb &box) const = 0;
    virtual rt_host_kind kind() const = 0;
};

struct flip_normals : hitable { // src/hitable.h:23-46
    hitable *ptr;
    flip_normals(hitable *h) : ptr(h) {}
    bool bounding_box(float t0, float t1, aabb &box) const override { return ptr->bounding_box(t0, t1, box); }
    rt_host_kind kind() const override { return RT_H_FLIP; }
};

class sphere : public hitable { // src/sphere.h:4-19, bbox :61-66
  public:
    material *mat = nullptr;
    vec3 center = vec3(0, 0, 0);
    float radius = 1.0f;
    sphere() {}
    sphere(vec3 cen, float r, material *m) : mat(m), center(cen), radius(r) {}
    bool bounding_box(float, float, aabb &box) const override
    {
        vec3 rr(radius, radius, radius);
        box = aabb(center - rr, center + rr);
        return true;
    }
    rt_host_kind kind() const override { return RT_H_SPHERE; }
};

class moving_sphere : public hitable { // src/sphere.h:115-139, bbox :185-193
  public:
    material *mat_ptr = nullptr;
    vec3 center0 = vec3(0, 0, 0), center1 = vec3(1, 0, 0);
    float time0 = 0.0f, time1 = 1.0f, radius = 1.0f;
    moving_sphere() {}
    moving_sphere(vec3 c0, vec3 c1, float t0, float t1, float r, material *m)
        : mat_ptr(m), center0(c0), center1(c1), time0(t0), time1(t1), radius(r) {}
    bool bounding_box(float, float, aabb &box) const override
    {
        vec3 rr(radius, radius, radius);
        box = surrounding_box(aabb(center0 - rr, center0 + rr), aabb(center1 - rr, center1 + rr));
        return true;
    }
    rt_host_kind kind() const override { return RT_H_MOVING_SPHERE; }
};

// axis-aligned rectangles, src/plane.h; the box is 2e-4 thick along the normal
struct xy_rect : hitable {
    material *mat_ptr = nullptr;
    float x0 = 0, x1 = 0, y0 = 0, y1 = 0, z = 0;
    xy_rect() {}
    xy_rect(float ax0, float ax1, float ay0, float ay1, float az, material *m)
        : mat_ptr(m), x0(ax0), x1(ax1), y0(ay0), y1(ay1), z(az) {}
    bool bounding_box(float, float, aabb &box) const override
    {
        box = aabb(vec3(x0, y0, z - 0.0001f), vec3(x1, y1, z + 0.0001f));
        return true;
    }
    rt_host_kind kind() const override { return RT_H_XY_RECT; }
};
struct xz_rect : hitable {
    material *mat_ptr = nullptr;
    float x0 = 0, x1 = 0, z0 = 0, z1 = 0, y = 0;
    xz_rect() {}
    xz_rect(float ax0, float ax1, float az0, float az1, float ay, material *m)
        : mat_ptr(m), x0(ax0), x1(ax1), z0(az0), z1(az1), y(ay) {}
    bool bounding_box(float, float, aabb &box) const override
    {
        box = aabb(vec3(x0, y - 0.0001f, z0), vec3(x1, y + 0.0001f, z1));
        return true;
    }
    rt_host_kind kind() const override { return RT_H_XZ_RECT; }
};
struct yz_rect : hitable {
    material *mat_ptr = nullptr;
    float z0 = 0, z1 = 0, y0 = 0, y1 = 0, x = 0;
    yz_rect() {}
    yz_rect(float ay0, float ay1, float az0, float az1, float ax, material *m)
        : mat_ptr(m), z0(az0), z1(az1), y0(ay0), y1(ay1), x(ax) {}
    bool bounding_box(float, float, aabb &box) const override
    {
        box = aabb(vec3(x - 0.0001f, y0, z0), vec3(x + 0.0001f, y1, z1));
        return true;
    }
    rt_host_kind kind() const override { return RT_H_YZ_RECT; }
};

class hitable_list : public hitable { // src/hitable_list.h:4-65
  public:
    hitable **list = nullptr;
    int list_size = 0;
    hitable_list() {}
    hitable_list(hitable **l, int n) : list(l), list_size(n) {}
    bool bounding_box(float t0, float t1, aabb &box) const override
    {
        if (list_size < 1) return false;
        aabb tmp;
        if (!list[0]->bounding_box(t0, t1, tmp)) return false;
        box = tmp;
        for (int i = 1; i < list_size; ++i) {
            if (!list[i]->bounding_box(t0, t1, tmp)) return false;
            box = surrounding_box(box, tmp);
        }
        return true;
    }
    rt_host_kind kind() const override { return RT_H_LIST; }
};

struct box : hitable { // src/box.h:4-33: +z,-z,+y,-y,+x,-x faces, the minus ones flipped
    vec3 pmin, pmax;
    hitable *list_ptr = nullptr;
    box() {}
    box(const vec3 &p0, const vec3 &p1, material *m) : pmin(p0), pmax(p1)
    {
        hitable **f = new hitable *[6];
        f[0] = new xy_rect(p0.x(), p1.x(), p0.y(), p1.y(), p1.z(), m);
        f[1] = new flip_normals(new xy_rect(p0.x(), p1.x(), p0.y(), p1.y(), p0.z(), m));
        f[2] = new xz_rect(p0.x(), p1.x(), p0.z(), p1.z(), p1.y(), m);
        f[3] = new flip_normals(new xz_rect(p0.x(), p1.x(), p0.z(), p1.z(), p0.y(), m));
        f[4] = new yz_rect(p0.y(), p1.y(), p0.z(), p1.z(), p1.x(), m);
        f[5] = new flip_normals(new yz_rect(p0.y(), p1.y(), p0.z(), p1.z(), p0.x(), m));
        list_ptr = new hitable_list(f, 6);
    }
    bool bounding_box(float, float, aabb &b) const override
    {
        b = aabb(pmin, pmax);
        return true;
    }
    rt_host_kind kind() const override { return RT_H_BOX; }
};

struct translate : hitable { // src/transforms.h:4-41
    hitable *ptr;
    vec3 offset;
    translate(hitable *p, const vec3 &t) : ptr(p), offset(t) {}
    bool bounding_box(float t0, float t1, aabb &b) const override
    {
        if (!ptr->bounding_box(t0, t1, b)) return false;
        b = aabb(b.min() + offset, b.max() + offset);
        return true;
    }
    rt_host_kind kind() const override { return RT_H_TRANSLATE; }
};

struct rotate_y : hitable { // src/transforms.h:45-100, 132-136
    hitable *ptr;
    float sin_theta, cos_theta;
    bool hasbox;
    aabb bbox;
    rotate_y(hitable *p, float angle) : ptr(p)
    {
        float radians = (PI / 180.0f) * angle;
        sin_theta = sinf(radians);
        cos_theta = cosf(radians);
        hasbox = ptr->bounding_box(0, 1, bbox);
        float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        // the eight corners of the child's box, rotated; the int*float products keep
        // the reference's rounding (and its NaN for an infinite box)
        for (int corner = 0; corner < 8; ++corner) {
            int i = (corner >> 2) & 1, j = (corner >> 1) & 1, k = corner & 1;
            float x = i * bbox.max().x() + (1 - i) * bbox.min().x();
            float y = j * bbox.max().y() + (1 - j) * bbox.min().y();
            float z = k * bbox.max().z() + (1 - k) * bbox.min().z();
            float c[3] = {cos_theta * x + sin_theta * z, y, -sin_theta * x + cos_theta * z};
            for (int a = 0; a < 3; ++a) {
                if (c[a] > hi[a]) hi[a] = c[a];
                if (c[a] < lo[a]) lo[a] = c[a];
            }
        }
        bbox = aabb(vec3(lo[0], lo[1], lo[2]), vec3(hi[0], hi[1], hi[2]));
    }
    bool bounding_box(float, float, aabb &b) const override
    {
        b = bbox;
        return hasbox;
    }
    rt_host_kind kind() const override { return RT_H_ROTATE_Y; }
};

struct constant_medium : hitable { // src/volume.h:4-21
    hitable *boundary;
    float density;
    material *phase_function;
    constant_medium(hitable *b, float d, texture *a) : boundary(b), density(d), phase_function(new isotropic(a)) {}
    bool bounding_box(float t0, float t1, aabb &b) const override { return boundary->bounding_box(t0, t1, b); }
    rt_host_kind kind() const override { return RT_H_MEDIUM; }
};

// ---------------------------------------------------------------------------
// bvh_node, src/bvh.h:4-121.  One RAN01() per node picks the axis, the caller's
// array is qsort()ed IN PLACE by box minimum with a comparator that never
// returns 0, the split is at n/2, n==1 duplicates the child, n==2 hangs both
// children directly under the node.  glibc's qsort is used on purpose: with
// such a comparator the resulting order belongs to the sort implementation.
// ---------------------------------------------------------------------------
template <int AXIS> inline int rt_box_min_compare(const void *pa, const void *pb)
{
    aabb ba, bb;
    const hitable *ha = *(hitable *const *)pa, *hb = *(hitable *const *)pb;
    if (!ha->bounding_box(0, 0, ba) || !hb->bounding_box(0, 0, bb))
        fprintf(stderr, "no bounding box in bvh_node constructor\n");
    return (ba.min()[AXIS] - bb.min()[AXIS] < 0.0) ? -1 : 1;
}

struct bvh_node : hitable {
    hitable *left = nullptr, *right = nullptr;
    aabb box;
    bvh_node() {}
    bvh_node(hitable **l, int n, float time0, float time1)
    {
        int axis = int(3.0f * RAN01());
        qsort(l, n, sizeof(hitable *),
              axis == 0 ? rt_box_min_compare<0> : axis == 1 ? rt_box_min_compare<1> : rt_box_min_compare<2>);
        if (n == 1) {
            left = right = l[0];
        } else if (n == 2) {
            left = l[0];
            right = l[1];
        } else {
            left = new bvh_node(l, n / 2, time0, time1);
            right = new bvh_node(l + n / 2, n - n / 2, time0, time1);
        }
        aabb bl, br;
        if (!left->bounding_box(time0, time1, bl) || !right->bounding_box(time0, time1, br))
            fprintf(stderr, "no bounding box in bvh_node constructor\n");
        box = surrounding_box(bl, br);
    }
    bool bounding_box(float, float, aabb &b) const override
    {
        b = box;
        return true;
    }
    rt_host_kind kind() const override { return RT_H_BVH; }
};

// ---------------------------------------------------------------------------
// camera, src/camera.h:4-52.  get_ray() runs on the device (rt_camera_rays / the
// render kernel); the constructor's arithmetic is kept operation for operation
// because primary rays must match the reference bit for bit.
// ---------------------------------------------------------------------------
struct camera {
    vec3 origin, u, v, w;
    vec3 lower_left_corner, horizontal, vertical;
    float lens_radius = 1.0f, time0 = 0.0f, time1 = 1.0f;
    camera(vec3 eye, vec3 lookat, vec3 up, float vfov, float aspect, float aperture, float focus_dist, float t0,
           float t1)
        : origin(eye), time0(t0), time1(t1)
    {
        lens_radius = aperture / 2.0f;
        float theta = vfov * PI / 180.0f;
        float half_height = 1.0f * tanf(theta / 2.0f);
        float half_width = aspect * half_height;
        w = unit_vector(eye - lookat);
        u = unit_vector(cross(up, w));
        v = cross(w, u);
        lower_left_corner = origin - u * half_width * focus_dist - v * half_height * focus_dist - w * focus_dist;
        horizontal = 2.0f * half_width * focus_dist * u;
        vertical = 2.0f * half_height * focus_dist * v;
    }
};

#endif // RT_SCENE_API_H_
