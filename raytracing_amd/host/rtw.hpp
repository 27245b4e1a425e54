// rtw.hpp -- the reference's scene-description surface, as a recording layer.
//
// A scene builder written against huris/RayTracing ("make_shared<sphere>(point3(0,1,0), 1.0,
// material1)", "world.add(...)", "make_shared<bvh_node>(world, 0.0, 1.0)", "camera cam(lookfrom,
// lookat, vup, 20, aspect, aperture, dist_to_focus)") compiles against this header unchanged.
// Nothing here intersects or shades anything: objects remember their constructor arguments (plus
// what the reference's constructors compute from them: bounding boxes, BVH topology, Perlin
// tables, rotation sines), and flattener::flatten() turns the object graph into the plain arrays
// of rt_scene_desc (include/raytracing_amd.h) that the GPU library consumes.
//
// Class and member names follow the reference (B1 = "Ray Tracing In a Weekend/code",
// B2 = "Ray Tracing The Next Week/code"):
//   vec3/point3/color  B1/vec3.h:9-68      hittable_list  B2/hittable_list.h:13-26
//   sphere  B2/sphere.h:9-39               moving_sphere  B2/moving_sphere.h:8-25
//   xy_rect/xz_rect/yz_rect  B2/aarect.h   box  B2/box.h:9-39      bvh_node  B2/bvh.h:10-89
//   translate/rotate_y  B2/hittable.h:31-111    constant_medium  B2/constant_medium.h:10-28
//   lambertian/metal/dielectric/diffuse_light/isotropic  B2/material.h
//   solid_color/checker_texture/noise_texture/image_texture  B2/texture.h    perlin  B2/perlin.h
//   camera  B2/camera.h:6-55
// Random numbers come from the C library's rand() exactly like B1/rtweekend.h:24-27, and every
// constructor that draws (perlin, bvh_node) draws in the reference's order, so a builder that
// follows the reference's statement order reproduces the reference's scene byte for byte.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "raytracing_amd.h"

namespace rtw {

using std::make_shared;
using std::shared_ptr;

const double infinity = std::numeric_limits<double>::infinity();
const double pi = std::acos(-1.0);

inline double degrees_to_radians(double degrees) { return degrees * pi / 180.0; }
inline double random_double() { return std::rand() / (RAND_MAX + 1.0); }
inline double random_double(double lo, double hi) { return lo + (hi - lo) * random_double(); }
inline int random_int(int lo, int hi) { return static_cast<int>(random_double(lo, hi + 1)); }

struct vec3 {
    double e[3];
    vec3() : e{0, 0, 0} {}
    vec3(double a, double b, double c) : e{a, b, c} {}
    double x() const { return e[0]; }
    double y() const { return e[1]; }
    double z() const { return e[2]; }
    double operator[](int i) const { return e[i]; }
    double& operator[](int i) { return e[i]; }
    vec3 operator-() const { return vec3(-e[0], -e[1], -e[2]); }
    vec3& operator+=(const vec3& o) {
        for (int i = 0; i < 3; ++i) e[i] += o.e[i];
        return *this;
    }
    vec3& operator*=(double t) {
        for (int i = 0; i < 3; ++i) e[i] *= t;
        return *this;
    }
    vec3& operator/=(double t) { return *this *= 1 / t; }
    double lenth_squared() const { return e[0] * e[0] + e[1] * e[1] + e[2] * e[2]; }  // [sic], B1/vec3.h:44
    double length_squared() const { return lenth_squared(); }
    double length() const { return std::sqrt(lenth_squared()); }
    // The reference writes vec3(random_double(), random_double(), random_double()); its canonical
    // build (g++) evaluates the arguments last to first, so the FIRST draw becomes z.  Spelled out
    // here so that every compiler rolls the same scene.
    static vec3 random() {
        const double c = random_double(), b = random_double(), a = random_double();
        return vec3(a, b, c);
    }
    static vec3 random(double lo, double hi) {
        const double c = random_double(lo, hi), b = random_double(lo, hi), a = random_double(lo, hi);
        return vec3(a, b, c);
    }
};
using point3 = vec3;
using color = vec3;

inline vec3 operator+(const vec3& u, const vec3& v) { return vec3(u.e[0] + v.e[0], u.e[1] + v.e[1], u.e[2] + v.e[2]); }
inline vec3 operator-(const vec3& u, const vec3& v) { return vec3(u.e[0] - v.e[0], u.e[1] - v.e[1], u.e[2] - v.e[2]); }
inline vec3 operator*(const vec3& u, const vec3& v) { return vec3(u.e[0] * v.e[0], u.e[1] * v.e[1], u.e[2] * v.e[2]); }
inline vec3 operator*(double t, const vec3& v) { return vec3(t * v.e[0], t * v.e[1], t * v.e[2]); }
inline vec3 operator*(const vec3& v, double t) { return t * v; }
inline vec3 operator/(const vec3& v, double t) { return (1 / t) * v; }
inline double dot(const vec3& u, const vec3& v) { return u.e[0] * v.e[0] + u.e[1] * v.e[1] + u.e[2] * v.e[2]; }
inline vec3 cross(const vec3& u, const vec3& v) {
    return vec3(u.e[1] * v.e[2] - u.e[2] * v.e[1], u.e[2] * v.e[0] - u.e[0] * v.e[2], u.e[0] * v.e[1] - u.e[1] * v.e[0]);
}
inline vec3 unit_vector(const vec3& v) { return v / v.length(); }

// ------------------------------------------------------------------ aabb (B2/aabb.h)
struct aabb {
    point3 minimum, maximum;
    aabb() {}
    aabb(const point3& a, const point3& b) : minimum(a), maximum(b) {}
    point3 min() const { return minimum; }
    point3 max() const { return maximum; }
};
inline aabb surrounding_box(const aabb& box0, const aabb& box1) {
    const point3 small(std::fmin(box0.min().x(), box1.min().x()), std::fmin(box0.min().y(), box1.min().y()),
                       std::fmin(box0.min().z(), box1.min().z()));
    const point3 big(std::fmax(box0.max().x(), box1.max().x()), std::fmax(box0.max().y(), box1.max().y()),
                     std::fmax(box0.max().z(), box1.max().z()));
    return aabb(small, big);
}

class flattener;

// ------------------------------------------------------------------ textures (B2/texture.h, B2/perlin.h)
struct texture {
    virtual ~texture() = default;
    virtual int emit(flattener& f) const = 0;  // index in flattener::textures
};

struct solid_color : texture {
    color color_value;
    solid_color() {}
    explicit solid_color(color c) : color_value(c) {}
    solid_color(double red, double green, double blue) : color_value(red, green, blue) {}
    int emit(flattener& f) const override;
};

struct checker_texture : texture {
    shared_ptr<texture> odd, even;
    checker_texture() {}
    checker_texture(shared_ptr<texture> _even, shared_ptr<texture> _odd) : odd(std::move(_odd)), even(std::move(_even)) {}
    checker_texture(color c1, color c2) : odd(make_shared<solid_color>(c2)), even(make_shared<solid_color>(c1)) {}
    int emit(flattener& f) const override;
};

// The tables a perlin object rolls in its constructor (B2/perlin.h:8-15, 70-84): 256 unit vectors,
// then three Fisher-Yates permutations, all from rand() in that order.
struct perlin {
    rt_perlin tables;
    perlin() {
        for (int i = 0; i < 256; ++i) {
            const vec3 v = unit_vector(vec3::random(-1, 1));
            for (int c = 0; c < 3; ++c) tables.ranvec[i][c] = v[c];
        }
        generate_perm(tables.perm_x);
        generate_perm(tables.perm_y);
        generate_perm(tables.perm_z);
    }

private:
    static void generate_perm(int32_t* p) {
        for (int i = 0; i < 256; ++i) p[i] = i;
        for (int i = 255; i > 0; --i) {
            const int target = random_int(0, i);
            std::swap(p[i], p[target]);
        }
    }
};

struct noise_texture : texture {
    perlin noise;
    double scale = 0;
    noise_texture() {}
    explicit noise_texture(double sc) : scale(sc) {}
    int emit(flattener& f) const override;
};

// image_texture (B2/texture.h:62-110).  The reference decodes with its vendored stb_image; this layer
// reads binary PPM (P6, maxval 255) and otherwise records "no data", for which the reference -- and
// the kernels -- return cyan (B2/texture.h:73-76, 87).  A caller with its own decoder uses the
// (pixels, width, height) constructor.
struct image_texture : texture {
    std::vector<uint8_t> data;
    int width = 0, height = 0;
    image_texture() {}
    explicit image_texture(const char* filename) {
        FILE* fp = std::fopen(filename, "rb");
        if (fp) {
            int w = 0, h = 0, maxv = 0;
            if (std::fscanf(fp, "P6 %d %d %d", &w, &h, &maxv) == 3 && maxv == 255 && w > 0 && h > 0 && std::fgetc(fp) != EOF) {
                data.resize((size_t)3 * w * h);
                if (std::fread(data.data(), 1, data.size(), fp) == data.size()) {
                    width = w;
                    height = h;
                } else {
                    data.clear();
                }
            }
            std::fclose(fp);
        }
        if (data.empty()) std::fprintf(stderr, "ERROR: Could not load texture image file '%s'.\n", filename);
    }
    image_texture(const uint8_t* rgb, int w, int h) : data(rgb, rgb + (size_t)3 * w * h), width(w), height(h) {}
    int emit(flattener& f) const override;
};

// ------------------------------------------------------------------ materials (B2/material.h)
struct material {
    virtual ~material() = default;
    virtual rt_material describe(flattener& f) const = 0;
};

struct lambertian : material {
    shared_ptr<texture> albedo;
    explicit lambertian(const color& a) : albedo(make_shared<solid_color>(a)) {}
    explicit lambertian(shared_ptr<texture> a) : albedo(std::move(a)) {}
    rt_material describe(flattener& f) const override;
};

struct metal : material {
    color albedo;
    double fuzz;
    metal(const color& a, double f) : albedo(a), fuzz(f < 1 ? f : 1) {}  // B1/material.h:38
    rt_material describe(flattener&) const override {
        rt_material m;
        std::memset(&m, 0, sizeof(m));
        m.kind = RT_MAT_METAL;
        m.texture = -1;
        for (int i = 0; i < 3; ++i) m.albedo[i] = albedo[i];
        m.param = fuzz;
        return m;
    }
};

struct dielectric : material {
    double ir;
    explicit dielectric(double index_of_refraction) : ir(index_of_refraction) {}
    rt_material describe(flattener&) const override {
        rt_material m;
        std::memset(&m, 0, sizeof(m));
        m.kind = RT_MAT_DIELECTRIC;
        m.texture = -1;
        m.param = ir;
        return m;
    }
};

struct diffuse_light : material {
    shared_ptr<texture> emit;
    explicit diffuse_light(shared_ptr<texture> a) : emit(std::move(a)) {}
    explicit diffuse_light(color c) : emit(make_shared<solid_color>(c)) {}
    rt_material describe(flattener& f) const override;
};

struct isotropic : material {
    shared_ptr<texture> albedo;
    explicit isotropic(color c) : albedo(make_shared<solid_color>(c)) {}
    explicit isotropic(shared_ptr<texture> a) : albedo(std::move(a)) {}
    rt_material describe(flattener& f) const override;
};

// ------------------------------------------------------------------ hittables
struct hittable {
    virtual ~hittable() = default;
    virtual int emit(flattener& f) const = 0;  // appends this object's node(s); returns its node index
    virtual bool bounding_box(double time0, double time1, aabb& output_box) const = 0;
};

struct sphere : hittable {
    point3 center;
    double radius;
    shared_ptr<material> mat_ptr;
    sphere(point3 cen, double r, shared_ptr<material> m) : center(cen), radius(r), mat_ptr(std::move(m)) {}
    int emit(flattener& f) const override;
    bool bounding_box(double, double, aabb& output_box) const override {  // B2/sphere.h:71-75
        output_box = aabb(center - vec3(radius, radius, radius), center + vec3(radius, radius, radius));
        return true;
    }
};

struct moving_sphere : hittable {
    point3 center0, center1;
    double time0, time1, radius;
    shared_ptr<material> mat_ptr;
    moving_sphere(point3 cen0, point3 cen1, double _time0, double _time1, double r, shared_ptr<material> m)
        : center0(cen0), center1(cen1), time0(_time0), time1(_time1), radius(r), mat_ptr(std::move(m)) {}
    point3 center(double time) const { return center0 + ((time - time0) / (time1 - time0)) * (center1 - center0); }
    int emit(flattener& f) const override;
    bool bounding_box(double _time0, double _time1, aabb& output_box) const override {  // B2/moving_sphere.h:59-68
        const aabb box0(center(_time0) - vec3(radius, radius, radius), center(_time0) + vec3(radius, radius, radius));
        const aabb box1(center(_time1) - vec3(radius, radius, radius), center(_time1) + vec3(radius, radius, radius));
        output_box = surrounding_box(box0, box1);
        return true;
    }
};

// One class for the three axis-aligned rectangles; xy_rect / xz_rect / yz_rect below fix the plane.
struct aarect : hittable {
    int kind;  // RT_NODE_XY_RECT / XZ / YZ
    double a0, a1, b0, b1, k;
    shared_ptr<material> mp;
    aarect(int kind_, double _a0, double _a1, double _b0, double _b1, double _k, shared_ptr<material> mat)
        : kind(kind_), a0(_a0), a1(_a1), b0(_b0), b1(_b1), k(_k), mp(std::move(mat)) {}
    int emit(flattener& f) const override;
    bool bounding_box(double, double, aabb& output_box) const override {  // B2/aarect.h:16-21, 58-63, 100-105
        if (kind == RT_NODE_XY_RECT) output_box = aabb(point3(a0, b0, k - 0.0001), point3(a1, b1, k + 0.0001));
        else if (kind == RT_NODE_XZ_RECT) output_box = aabb(point3(a0, k - 0.0001, b0), point3(a1, k + 0.0001, b1));
        else output_box = aabb(point3(k - 0.0001, a0, b0), point3(k + 0.0001, a1, b1));
        return true;
    }
};
struct xy_rect : aarect {
    xy_rect(double x0, double x1, double y0, double y1, double k, shared_ptr<material> m) : aarect(RT_NODE_XY_RECT, x0, x1, y0, y1, k, std::move(m)) {}
};
struct xz_rect : aarect {
    xz_rect(double x0, double x1, double z0, double z1, double k, shared_ptr<material> m) : aarect(RT_NODE_XZ_RECT, x0, x1, z0, z1, k, std::move(m)) {}
};
struct yz_rect : aarect {
    yz_rect(double y0, double y1, double z0, double z1, double k, shared_ptr<material> m) : aarect(RT_NODE_YZ_RECT, y0, y1, z0, z1, k, std::move(m)) {}
};

struct hittable_list : hittable {
    std::vector<shared_ptr<hittable>> objects;
    hittable_list() {}
    explicit hittable_list(shared_ptr<hittable> object) { add(std::move(object)); }
    void clear() { objects.clear(); }
    void add(shared_ptr<hittable> object) { objects.push_back(std::move(object)); }
    int emit(flattener& f) const override;
    bool bounding_box(double time0, double time1, aabb& output_box) const override {  // B2/hittable_list.h:43-55
        if (objects.empty()) return false;
        aabb temp_box;
        bool first_box = true;
        for (const auto& object : objects) {
            if (!object->bounding_box(time0, time1, temp_box)) return false;
            output_box = first_box ? temp_box : surrounding_box(output_box, temp_box);
            first_box = false;
        }
        return true;
    }
};

struct box : hittable {  // B2/box.h:27-39: six rectangles in this order
    point3 box_min, box_max;
    hittable_list sides;
    box(const point3& p0, const point3& p1, shared_ptr<material> ptr) : box_min(p0), box_max(p1) {
        sides.add(make_shared<xy_rect>(p0.x(), p1.x(), p0.y(), p1.y(), p1.z(), ptr));
        sides.add(make_shared<xy_rect>(p0.x(), p1.x(), p0.y(), p1.y(), p0.z(), ptr));
        sides.add(make_shared<xz_rect>(p0.x(), p1.x(), p0.z(), p1.z(), p1.y(), ptr));
        sides.add(make_shared<xz_rect>(p0.x(), p1.x(), p0.z(), p1.z(), p0.y(), ptr));
        sides.add(make_shared<yz_rect>(p0.y(), p1.y(), p0.z(), p1.z(), p1.x(), ptr));
        sides.add(make_shared<yz_rect>(p0.y(), p1.y(), p0.z(), p1.z(), p0.x(), ptr));
    }
    int emit(flattener& f) const override;
    bool bounding_box(double, double, aabb& output_box) const override {
        output_box = aabb(box_min, box_max);
        return true;
    }
};

struct translate : hittable {
    shared_ptr<hittable> ptr;
    vec3 offset;
    translate(shared_ptr<hittable> p, const vec3& displacement) : ptr(std::move(p)), offset(displacement) {}
    int emit(flattener& f) const override;
    bool bounding_box(double time0, double time1, aabb& output_box) const override {  // B2/hittable.h:54-60
        if (!ptr->bounding_box(time0, time1, output_box)) return false;
        output_box = aabb(output_box.min() + offset, output_box.max() + offset);
        return true;
    }
};

struct rotate_y : hittable {
    shared_ptr<hittable> ptr;
    double sin_theta, cos_theta;
    bool hasbox;
    aabb bbox;
    rotate_y(shared_ptr<hittable> p, double angle) : ptr(std::move(p)) {  // B2/hittable.h:81-111
        const double radians = degrees_to_radians(angle);
        sin_theta = std::sin(radians);
        cos_theta = std::cos(radians);
        hasbox = ptr->bounding_box(0, 1, bbox);
        point3 mn(infinity, infinity, infinity), mx(-infinity, -infinity, -infinity);
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 2; j++)
                for (int k = 0; k < 2; k++) {
                    const double x = i * bbox.max().x() + (1 - i) * bbox.min().x();
                    const double y = j * bbox.max().y() + (1 - j) * bbox.min().y();
                    const double z = k * bbox.max().z() + (1 - k) * bbox.min().z();
                    const double newx = cos_theta * x + sin_theta * z;
                    const double newz = -sin_theta * x + cos_theta * z;
                    const vec3 tester(newx, y, newz);
                    for (int c = 0; c < 3; c++) {
                        mn[c] = std::fmin(mn[c], tester[c]);
                        mx[c] = std::fmax(mx[c], tester[c]);
                    }
                }
        bbox = aabb(mn, mx);
    }
    int emit(flattener& f) const override;
    bool bounding_box(double, double, aabb& output_box) const override {
        output_box = bbox;
        return hasbox;
    }
};

// flip_face of "The Rest Of Your Life" (public book text; the reference repository has no book-3 code)
struct flip_face : hittable {
    shared_ptr<hittable> ptr;
    explicit flip_face(shared_ptr<hittable> p) : ptr(std::move(p)) {}
    int emit(flattener& f) const override;
    bool bounding_box(double time0, double time1, aabb& output_box) const override {
        return ptr->bounding_box(time0, time1, output_box);
    }
};

struct constant_medium : hittable {
    shared_ptr<hittable> boundary;
    shared_ptr<material> phase_function;
    double neg_inv_density;
    constant_medium(shared_ptr<hittable> b, double d, shared_ptr<texture> a)
        : boundary(std::move(b)), phase_function(make_shared<isotropic>(std::move(a))), neg_inv_density(-1 / d) {}
    constant_medium(shared_ptr<hittable> b, double d, color c)
        : boundary(std::move(b)), phase_function(make_shared<isotropic>(c)), neg_inv_density(-1 / d) {}
    int emit(flattener& f) const override;
    bool bounding_box(double time0, double time1, aabb& output_box) const override {
        return boundary->bounding_box(time0, time1, output_box);
    }
};

// bvh_node (B2/bvh.h:51-89): the constructor's algorithm step for step -- it copies the object vector
// at every node, draws the split axis with random_int(0,2), compares boxes taken at time (0,0), sorts
// the span with std::sort and splits at the middle; a span of one object becomes BOTH children.
struct bvh_node : hittable {
    shared_ptr<hittable> left, right;
    aabb bounds;
    bvh_node(const hittable_list& list, double time0, double time1) : bvh_node(list.objects, 0, list.objects.size(), time0, time1) {}
    bvh_node(const std::vector<shared_ptr<hittable>>& src_objects, size_t start, size_t end, double time0, double time1) {
        auto objects = src_objects;
        const int axis = random_int(0, 2);
        auto comparator = [axis](const shared_ptr<hittable> a, const shared_ptr<hittable> b) {
            aabb box_a, box_b;
            if (!a->bounding_box(0, 0, box_a) || !b->bounding_box(0, 0, box_b)) std::fprintf(stderr, "No bounding box in bvh_node constructor.\n");
            return box_a.min().e[axis] < box_b.min().e[axis];
        };
        const size_t object_span = end - start;
        if (object_span == 1) {
            left = right = objects[start];
        } else if (object_span == 2) {
            if (comparator(objects[start], objects[start + 1])) {
                left = objects[start];
                right = objects[start + 1];
            } else {
                left = objects[start + 1];
                right = objects[start];
            }
        } else {
            std::sort(objects.begin() + start, objects.begin() + end, comparator);
            const size_t mid = start + object_span / 2;
            left = make_shared<bvh_node>(objects, start, mid, time0, time1);
            right = make_shared<bvh_node>(objects, mid, end, time0, time1);
        }
        aabb box_left, box_right;
        if (!left->bounding_box(time0, time1, box_left) || !right->bounding_box(time0, time1, box_right))
            std::fprintf(stderr, "No bounding box in bvh_node constructor.\n");
        bounds = surrounding_box(box_left, box_right);
    }
    int emit(flattener& f) const override;
    bool bounding_box(double, double, aabb& output_box) const override {
        output_box = bounds;
        return true;
    }
};

// ------------------------------------------------------------------ flattener
// Owns the arrays an rt_scene_desc points to.  Nodes are written parent first; an object reachable
// twice (a one-object bvh_node's two children, a medium's boundary that is also in the world) is
// written twice, so the graph becomes a tree.  Materials and textures are numbered in order of first
// use; a checker's two sub-textures come before it, a noise texture's table before it.  The golden
// scene files dumped from the reference program use the same conventions.
class flattener {
public:
    std::vector<rt_node> nodes;
    std::vector<int32_t> children;
    std::vector<rt_material> materials;
    std::vector<rt_texture> textures;
    std::vector<rt_perlin> perlins;
    std::vector<rt_image> images;
    std::vector<std::vector<uint8_t>> image_bytes;

    int material_index(const shared_ptr<material>& m) {
        if (!m) return -1;  // the shapes of a light list carry no material
        auto it = mats_seen_.find(m.get());
        if (it != mats_seen_.end()) return it->second;
        const rt_material d = m->describe(*this);
        materials.push_back(d);
        return mats_seen_[m.get()] = (int)materials.size() - 1;
    }
    int texture_index(const shared_ptr<texture>& t) {
        auto it = tex_seen_.find(t.get());
        if (it != tex_seen_.end()) return it->second;
        const int i = t->emit(*this);
        return tex_seen_[t.get()] = i;
    }
    int push_texture(int kind, int a = -1, int b = -1) {
        rt_texture t;
        std::memset(&t, 0, sizeof(t));
        t.kind = kind;
        t.a = a;
        t.b = b;
        textures.push_back(t);
        return (int)textures.size() - 1;
    }
    int push_node(int kind, int material) {
        rt_node n;
        std::memset(&n, 0, sizeof(n));
        n.kind = kind;
        n.material = material;
        n.child0 = n.child1 = -1;
        nodes.push_back(n);
        return (int)nodes.size() - 1;
    }
    int push_list(int kind, const std::vector<shared_ptr<hittable>>& objs) {
        const int self = push_node(kind, -1);
        std::vector<int32_t> kids;
        kids.reserve(objs.size());
        for (const auto& o : objs) kids.push_back(o->emit(*this));
        nodes[self].child0 = (int32_t)children.size();
        nodes[self].child1 = (int32_t)kids.size();
        children.insert(children.end(), kids.begin(), kids.end());
        return self;
    }
    // world must be the root list.  sky_mode / background as in rt_scene_desc.  `lights` = the shapes light is
    // sampled towards under RT_RENDER_IMPORTANCE (book 3's lights argument), written after the world as a list
    // node of their own.
    rt_scene_desc flatten(const hittable_list& world, int sky_mode = RT_SKY_GRADIENT, color background = color(0, 0, 0),
                          const hittable_list* lights = nullptr) {
        nodes.clear();
        children.clear();
        materials.clear();
        textures.clear();
        perlins.clear();
        images.clear();
        image_bytes.clear();
        mats_seen_.clear();
        tex_seen_.clear();
        const int root = world.emit(*this);
        const int lights_node = lights ? lights->emit(*this) : -1;
        for (size_t i = 0; i < images.size(); ++i) images[i].rgb = image_bytes[i].data();
        rt_scene_desc d;
        std::memset(&d, 0, sizeof(d));
        d.abi_version = RT_ABI_VERSION;
        d.root = root;
        d.n_nodes = (int32_t)nodes.size();
        d.n_children = (int32_t)children.size();
        d.n_materials = (int32_t)materials.size();
        d.n_textures = (int32_t)textures.size();
        d.n_perlins = (int32_t)perlins.size();
        d.n_images = (int32_t)images.size();
        d.nodes = nodes.data();
        d.children = children.data();
        d.materials = materials.data();
        d.textures = textures.data();
        d.perlins = perlins.data();
        d.images = images.data();
        d.sky_mode = sky_mode;
        d.lights_plus1 = lights_node + 1;
        for (int i = 0; i < 3; ++i) d.background[i] = background[i];
        return d;
    }

private:
    std::unordered_map<const material*, int> mats_seen_;
    std::unordered_map<const texture*, int> tex_seen_;
};

inline int solid_color::emit(flattener& f) const {
    const int i = f.push_texture(RT_TEX_SOLID);
    for (int c = 0; c < 3; ++c) f.textures[i].p[c] = color_value[c];
    return i;
}
inline int checker_texture::emit(flattener& f) const {
    const int e = f.texture_index(even);
    const int o = f.texture_index(odd);
    return f.push_texture(RT_TEX_CHECKER, e, o);
}
inline int noise_texture::emit(flattener& f) const {
    f.perlins.push_back(noise.tables);
    const int i = f.push_texture(RT_TEX_NOISE, (int)f.perlins.size() - 1);
    f.textures[i].p[0] = scale;
    return i;
}
inline int image_texture::emit(flattener& f) const {
    int image = -1;
    if (!data.empty()) {
        f.image_bytes.push_back(data);
        rt_image im;
        im.width = width;
        im.height = height;
        im.rgb = nullptr;  // fixed up in flatten()
        f.images.push_back(im);
        image = (int)f.images.size() - 1;
    }
    return f.push_texture(RT_TEX_IMAGE, image);
}

inline rt_material textured(flattener& f, int kind, const shared_ptr<texture>& t) {
    rt_material m;
    std::memset(&m, 0, sizeof(m));
    m.kind = kind;
    m.texture = f.texture_index(t);
    return m;
}
inline rt_material lambertian::describe(flattener& f) const { return textured(f, RT_MAT_LAMBERTIAN, albedo); }
inline rt_material diffuse_light::describe(flattener& f) const { return textured(f, RT_MAT_DIFFUSE_LIGHT, emit); }
inline rt_material isotropic::describe(flattener& f) const { return textured(f, RT_MAT_ISOTROPIC, albedo); }

inline int sphere::emit(flattener& f) const {
    const int i = f.push_node(RT_NODE_SPHERE, f.material_index(mat_ptr));
    for (int c = 0; c < 3; ++c) f.nodes[i].p[c] = center[c];
    f.nodes[i].p[3] = radius;
    return i;
}
inline int moving_sphere::emit(flattener& f) const {
    const int i = f.push_node(RT_NODE_MOVING_SPHERE, f.material_index(mat_ptr));
    for (int c = 0; c < 3; ++c) {
        f.nodes[i].p[c] = center0[c];
        f.nodes[i].p[3 + c] = center1[c];
    }
    f.nodes[i].p[6] = time0;
    f.nodes[i].p[7] = time1;
    f.nodes[i].p[8] = radius;
    return i;
}
inline int aarect::emit(flattener& f) const {
    const int i = f.push_node(kind, f.material_index(mp));
    const double v[5] = {a0, a1, b0, b1, k};
    std::memcpy(f.nodes[i].p, v, sizeof(v));
    return i;
}
inline int hittable_list::emit(flattener& f) const { return f.push_list(RT_NODE_LIST, objects); }
inline int box::emit(flattener& f) const {
    const int i = f.push_list(RT_NODE_BOX, sides.objects);
    for (int c = 0; c < 3; ++c) {
        f.nodes[i].p[c] = box_min[c];
        f.nodes[i].p[3 + c] = box_max[c];
    }
    return i;
}
inline int bvh_node::emit(flattener& f) const {
    const int i = f.push_node(RT_NODE_BVH, -1);
    for (int c = 0; c < 3; ++c) {
        f.nodes[i].p[c] = bounds.min()[c];
        f.nodes[i].p[3 + c] = bounds.max()[c];
    }
    const int l = left->emit(f);
    const int r = right->emit(f);
    f.nodes[i].child0 = l;
    f.nodes[i].child1 = r;
    return i;
}
inline int translate::emit(flattener& f) const {
    const int i = f.push_node(RT_NODE_TRANSLATE, -1);
    for (int c = 0; c < 3; ++c) f.nodes[i].p[c] = offset[c];
    const int k = ptr->emit(f);
    f.nodes[i].child0 = k;
    return i;
}
inline int flip_face::emit(flattener& f) const {
    const int i = f.push_node(RT_NODE_FLIP_FACE, -1);
    const int k = ptr->emit(f);
    f.nodes[i].child0 = k;
    return i;
}
inline int rotate_y::emit(flattener& f) const {
    const int i = f.push_node(RT_NODE_ROTATE_Y, -1);
    f.nodes[i].p[0] = sin_theta;
    f.nodes[i].p[1] = cos_theta;
    f.nodes[i].p[2] = hasbox ? 1.0 : 0.0;
    for (int c = 0; c < 3; ++c) {
        f.nodes[i].p[3 + c] = bbox.min()[c];
        f.nodes[i].p[6 + c] = bbox.max()[c];
    }
    const int k = ptr->emit(f);
    f.nodes[i].child0 = k;
    return i;
}
inline int constant_medium::emit(flattener& f) const {
    const int i = f.push_node(RT_NODE_MEDIUM, f.material_index(phase_function));
    f.nodes[i].p[0] = neg_inv_density;
    const int k = boundary->emit(f);
    f.nodes[i].child0 = k;
    return i;
}

// ------------------------------------------------------------------ camera (B2/camera.h:6-55)
class camera {
public:
    camera(point3 lookfrom, point3 lookat, vec3 vup, double vfov, double aspect_ratio, double aperture,
           double focus_dist, double time0 = 0, double time1 = 0) {
        rt_camera_init(&desc_, lookfrom.e, lookat.e, vup.e, vfov, aspect_ratio, aperture, focus_dist, time0, time1);
    }
    const rt_camera_desc& desc() const { return desc_; }

private:
    rt_camera_desc desc_;
};

}  // namespace rtw
