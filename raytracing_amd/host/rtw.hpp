// rtw.hpp -- the reference's scene-description surface, as a recording layer.
//
// A scene builder written against huris/RayTracing ("make_shared<sphere>(point3(0,1,0), 1.0,
// material1)", "world.add(...)", "camera cam(lookfrom, lookat, vup, 20, aspect, aperture,
// dist_to_focus)") compiles against this header unchanged.  Nothing here intersects or
// shades anything: objects only remember their constructor arguments, and flatten() turns
// the object graph into the plain arrays of rt_scene_desc (include/raytracing_amd.h) that
// the GPU library consumes.
//
// Class and member names follow the reference (B1 = "Ray Tracing In a Weekend/code",
// B2 = "Ray Tracing The Next Week/code"):
//   vec3/point3/color        B1/vec3.h:9-68        sphere          B1/sphere.h:9-21
//   hittable, hittable_list  B1/hittable.h:22-25, B1/hittable_list.h:12-24
//   lambertian/metal/dielectric  B1/material.h:15-89      camera   B1/camera.h:6-48
// Random numbers come from the C library's rand() exactly like B1/rtweekend.h:24-27, so a
// builder that draws in the reference's order reproduces the reference's scene.
#pragma once

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <stdexcept>
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

// ------------------------------------------------------------------ materials
struct material {
    virtual ~material() = default;
    virtual rt_material describe(std::vector<rt_texture>& textures) const = 0;
};

inline int push_solid(std::vector<rt_texture>& textures, const color& c) {
    rt_texture t;
    std::memset(&t, 0, sizeof(t));
    t.kind = RT_TEX_SOLID;
    t.a = t.b = -1;
    for (int i = 0; i < 3; ++i) t.p[i] = c[i];
    textures.push_back(t);
    return (int)textures.size() - 1;
}

struct lambertian : material {
    color albedo;
    explicit lambertian(const color& a) : albedo(a) {}
    rt_material describe(std::vector<rt_texture>& textures) const override {
        rt_material m;
        std::memset(&m, 0, sizeof(m));
        m.kind = RT_MAT_LAMBERTIAN;
        m.texture = push_solid(textures, albedo);
        for (int i = 0; i < 3; ++i) m.albedo[i] = albedo[i];
        return m;
    }
};

struct metal : material {
    color albedo;
    double fuzz;
    metal(const color& a, double f) : albedo(a), fuzz(f < 1 ? f : 1) {}  // B1/material.h:38
    rt_material describe(std::vector<rt_texture>&) const override {
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
    rt_material describe(std::vector<rt_texture>&) const override {
        rt_material m;
        std::memset(&m, 0, sizeof(m));
        m.kind = RT_MAT_DIELECTRIC;
        m.texture = -1;
        m.param = ir;
        return m;
    }
};

// ------------------------------------------------------------------ hittables
class flattener;

struct hittable {
    virtual ~hittable() = default;
    virtual int emit(flattener& f) const = 0;  // appends this object's node(s); returns its node index
};

struct sphere : hittable {
    point3 center;
    double radius;
    shared_ptr<material> mat_ptr;
    sphere(point3 cen, double r, shared_ptr<material> m) : center(cen), radius(r), mat_ptr(std::move(m)) {}
    int emit(flattener& f) const override;
};

struct hittable_list : hittable {
    std::vector<shared_ptr<hittable>> objects;
    hittable_list() {}
    explicit hittable_list(shared_ptr<hittable> object) { add(std::move(object)); }
    void clear() { objects.clear(); }
    void add(shared_ptr<hittable> object) { objects.push_back(std::move(object)); }
    int emit(flattener& f) const override;
};

// Owns the arrays an rt_scene_desc points to.  Materials are numbered in order of first use
// and a lambertian's solid texture is created just before it, which is also how the golden
// scene files are laid out.
class flattener {
public:
    std::vector<rt_node> nodes;
    std::vector<int32_t> children;
    std::vector<rt_material> materials;
    std::vector<rt_texture> textures;
    std::vector<rt_perlin> perlins;
    std::vector<rt_image> images;

    int material_index(const shared_ptr<material>& m) {
        auto it = seen_.find(m.get());
        if (it != seen_.end()) return it->second;
        materials.push_back(m->describe(textures));
        return seen_[m.get()] = (int)materials.size() - 1;
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
    // world must be the root list.  sky_mode / background as in rt_scene_desc.
    rt_scene_desc flatten(const hittable_list& world, int sky_mode = RT_SKY_GRADIENT, color background = color(0, 0, 0)) {
        nodes.clear();
        children.clear();
        materials.clear();
        textures.clear();
        seen_.clear();
        const int root = world.emit(*this);
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
        for (int i = 0; i < 3; ++i) d.background[i] = background[i];
        return d;
    }

private:
    std::unordered_map<const material*, int> seen_;
};

inline int sphere::emit(flattener& f) const {
    const int m = f.material_index(mat_ptr);
    const int i = f.push_node(RT_NODE_SPHERE, m);
    for (int c = 0; c < 3; ++c) f.nodes[i].p[c] = center[c];
    f.nodes[i].p[3] = radius;
    return i;
}

inline int hittable_list::emit(flattener& f) const {
    const int self = f.push_node(RT_NODE_LIST, -1);
    std::vector<int32_t> kids;
    kids.reserve(objects.size());
    for (const auto& o : objects) kids.push_back(o->emit(f));
    f.nodes[self].child0 = (int32_t)f.children.size();
    f.nodes[self].child1 = (int32_t)kids.size();
    f.children.insert(f.children.end(), kids.begin(), kids.end());
    return self;
}

// ------------------------------------------------------------------ camera (B1/camera.h:6-48)
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
