// gfx950 kernels for general scene graphs (books 2 and 3: BVH, moving spheres, rectangles and boxes,
// translate / rotate_y instances, constant media, textures, emitters).
//
// Replaces, for every pixel and sample:
//   B2/RayTracing.cpp:369-373  jitter -> camera::get_ray (with shutter time) -> ray_color
//   B2/RayTracing.cpp:17-36    ray_color with background and emitted light
//   B2/bvh.h:96-103, B2/aabb.h:34-47, B2/hittable_list.h:28-41   traversal
//   B2/sphere.h:41-69, B2/moving_sphere.h:31-57, B2/aarect.h:28-130, B2/box.h:41-43
//   B2/hittable.h:43-52,113-140 (translate, rotate_y), B2/constant_medium.h:30-66
//   B2/material.h:25-131, B2/texture.h:22-104, B2/perlin.h:24-104
//
// Traversal keeps the reference's semantics exactly: the recursion is unrolled into a per-lane stack
// (node ids in LDS) that visits objects in the reference's order -- list children first to last,
// BVH left before right, both always visited if the box is hit -- with one closest-so-far value playing
// the role of the t_max the reference threads through its calls.  Every test is done in double with the
// reference's operation order, so winner, t, hit point and normal are the reference's, bit for bit.
// A hit only records (t, node, enclosing instance); the full hit record is rebuilt once, for the winner.
#include "rt_device_common.h"
#include "rt_internal.h"
#include "rt_render_common.h"

namespace rtd {

namespace {

// Stack entry: node id in the low 16 bits; high 16 bits = END_XF (leave that instance node) or, for a
// list / box, the index of the next child to visit (a list occupies one slot however long it is).
constexpr uint32_t END_XF = 0xFFFFu;
constexpr double T_MIN = 0.001;      // B2/RayTracing.cpp:25

struct Ray {
    d3 o, d;
};

__device__ __forceinline__ double comp(d3 v, int axis) { return axis == 0 ? v.x : (axis == 1 ? v.y : v.z); }

// Ray handed to the child of a translate / rotate_y node (B2/hittable.h:44, 114-123).
__device__ __forceinline__ Ray xf_apply(const SceneNode& n, Ray r) {
    if (n.kind == RT_NODE_TRANSLATE) {
        r.o = r.o - d3{n.p[0], n.p[1], n.p[2]};
        return r;
    }
    const double sin_theta = n.p[0], cos_theta = n.p[1];
    Ray q = r;
    q.o.x = cos_theta * r.o.x - sin_theta * r.o.z;
    q.o.z = sin_theta * r.o.x + cos_theta * r.o.z;
    q.d.x = cos_theta * r.d.x - sin_theta * r.d.z;
    q.d.z = sin_theta * r.d.x + cos_theta * r.d.z;
    return q;
}

// The world ray as seen inside instance node `xf` (-1: the world itself): the enclosing instances'
// transforms applied outermost first, recomputed from the world ray so that no ray stack is needed.
__device__ __forceinline__ Ray ray_in(const SceneDev& S, int xf, Ray world) {
    int c0 = -1, c1 = -1, c2 = -1, c3 = -1;
    int x = xf;
    if (x >= 0) { c0 = x; x = S.parent_xf[x]; }
    if (x >= 0) { c1 = x; x = S.parent_xf[x]; }
    if (x >= 0) { c2 = x; x = S.parent_xf[x]; }
    if (x >= 0) { c3 = x; }
    Ray r = world;
    if (c3 >= 0) r = xf_apply(S.nodes[c3], r);
    if (c2 >= 0) r = xf_apply(S.nodes[c2], r);
    if (c1 >= 0) r = xf_apply(S.nodes[c1], r);
    if (c0 >= 0) r = xf_apply(S.nodes[c0], r);
    return r;
}

// sphere::hit up to the accepted root (B2/sphere.h:42-60).
__device__ __forceinline__ bool hit_sphere(d3 center, double radius, const Ray& r, double a, double t_min, double t_max,
                                           double& t) {
    const d3 oc = r.o - center;
    const double half_b = dot(oc, r.d);
    const double c = len2(oc) - radius * radius;
    const double discriminant = half_b * half_b - a * c;
    if (discriminant < 0) return false;
    const double sqrtd = sqrt(discriminant);
    double root = (-half_b - sqrtd) / a;
    if (root < t_min || root > t_max) {
        root = (-half_b + sqrtd) / a;
        if (root < t_min || root > t_max) return false;
    }
    t = root;
    return true;
}

// moving_sphere::center (B2/moving_sphere.h:27-29)
__device__ __forceinline__ d3 moving_center(const SceneDev& S, const SceneNode& n, double tm) {
    const d3 c0{n.p[0], n.p[1], n.p[2]}, c1{n.p[3], n.p[4], n.p[5]};
    const double time0 = S.extra[n.c], time1 = S.extra[n.c + 1];
    return c0 + ((tm - time0) / (time1 - time0)) * (c1 - c0);
}

// xy_rect / xz_rect / yz_rect ::hit up to acceptance (B2/aarect.h:28-34, 70-76, 112-118)
__device__ __forceinline__ bool hit_rect(const SceneNode& n, const Ray& r, double t_min, double t_max, double& t,
                                         double& ua, double& vb) {
    const int ka = n.kind == RT_NODE_XY_RECT ? 2 : (n.kind == RT_NODE_XZ_RECT ? 1 : 0);
    const int aa = n.kind == RT_NODE_YZ_RECT ? 1 : 0;
    const int ba = n.kind == RT_NODE_XY_RECT ? 1 : 2;
    const double tt = (n.p[4] - comp(r.o, ka)) / comp(r.d, ka);
    if (tt < t_min || tt > t_max) return false;
    const double a = comp(r.o, aa) + tt * comp(r.d, aa);
    const double b = comp(r.o, ba) + tt * comp(r.d, ba);
    if (a < n.p[0] || a > n.p[1] || b < n.p[2] || b > n.p[3]) return false;
    t = tt;
    ua = a;
    vb = b;
    return true;
}

__device__ __forceinline__ bool prim_hit(const SceneDev& S, const SceneNode& n, const Ray& r, double a, double tm,
                                         double t_min, double t_max, double& t) {
    double ua, vb;
    switch (n.kind) {
    case RT_NODE_SPHERE: return hit_sphere(d3{n.p[0], n.p[1], n.p[2]}, n.p[3], r, a, t_min, t_max, t);
    case RT_NODE_MOVING_SPHERE: return hit_sphere(moving_center(S, n, tm), S.extra[n.c + 2], r, a, t_min, t_max, t);
    case RT_NODE_XY_RECT:
    case RT_NODE_XZ_RECT:
    case RT_NODE_YZ_RECT: return hit_rect(n, r, t_min, t_max, t, ua, vb);
    default: return false;
    }
}

// aabb::hit (B2/aabb.h:34-47) with the per-axis reciprocal computed once per ray (same value)
__device__ __forceinline__ bool hit_aabb(const double* p, const Ray& r, d3 inv, double t_min, double t_max) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double invD = comp(inv, a);
        double t0 = (p[a] - comp(r.o, a)) * invD;
        double t1 = (p[3 + a] - comp(r.o, a)) * invD;
        if (invD < 0.0f) {
            const double tmp = t0;
            t0 = t1;
            t1 = tmp;
        }
        t_min = t0 > t_min ? t0 : t_min;
        t_max = t1 < t_max ? t1 : t_max;
        if (t_max <= t_min) return false;
    }
    return true;
}

// boundary->hit(r, t_min, t_max, rec) as constant_medium::hit calls it; only rec.t is used there.
// Supported boundaries: a sphere, moving sphere, rectangle, a list/box of those, under up to
// XF_DEPTH_CAP translate / rotate_y levels (checked at upload).
__device__ __forceinline__ bool boundary_hit(const SceneDev& S, int node, Ray r, double tm, double t_min, double t_max,
                                             double& t_out) {
    SceneNode n = S.nodes[node];
    for (int lvl = 0; lvl < XF_DEPTH_CAP && (n.kind == RT_NODE_TRANSLATE || n.kind == RT_NODE_ROTATE_Y); ++lvl) {
        r = xf_apply(n, r);
        n = S.nodes[n.a];
    }
    const double a = len2(r.d);
    if (n.kind == RT_NODE_LIST || n.kind == RT_NODE_BOX) {
        bool any = false;
        double best = t_max;
        for (int i = 0; i < n.b; ++i) {
            const SceneNode c = S.nodes[S.children[n.a + i]];
            double t;
            if (prim_hit(S, c, r, a, tm, t_min, best, t)) {
                best = t;
                any = true;
            }
        }
        t_out = best;
        return any;
    }
    return prim_hit(S, n, r, a, tm, t_min, t_max, t_out);
}

struct Hit {
    double t;
    int32_t node;  // winning primitive or medium node, -1 = miss
    int32_t xf;    // instance node enclosing it, -1 = none
};

struct RngCtx {
    uint32_t pixel, sample, site, k0, k1;
};

// world.hit(r, 0.001, infinity, rec) for one ray per lane.  All 64 lanes must call it.
__device__ __forceinline__ Hit trace(const SceneDev& S, Ray w, double tm, bool on, uint32_t* stack, int lane,
                                     const RngCtx& g) {
    Hit h{__builtin_inf(), -1, -1};
    Ray cur = w;
    int cur_xf = -1;
    d3 inv{1.0f / cur.d.x, 1.0f / cur.d.y, 1.0f / cur.d.z};
    double a = len2(cur.d);
    int sp = 0;
    if (on) stack[(sp++) * 64 + lane] = (uint32_t)S.root;

    while (ballot(sp > 0) != 0ull) {
        if (sp > 0) {
            const uint32_t entry = stack[(--sp) * 64 + lane];
            const int e = (int)(entry & 0xFFFFu);
            const uint32_t tag = entry >> 16;
            if (tag == END_XF) {
                cur_xf = S.parent_xf[e];
                cur = ray_in(S, cur_xf, w);
                inv = d3{1.0f / cur.d.x, 1.0f / cur.d.y, 1.0f / cur.d.z};
                a = len2(cur.d);
                continue;
            }
            const SceneNode n = S.nodes[e];
            switch (n.kind) {
            case RT_NODE_BVH:
                if (hit_aabb(n.p, cur, inv, T_MIN, h.t)) {
                    stack[(sp++) * 64 + lane] = (uint32_t)n.b;
                    stack[(sp++) * 64 + lane] = (uint32_t)n.a;
                }
                break;
            case RT_NODE_LIST:
            case RT_NODE_BOX:  // visit child `tag` next; the list stays on the stack until its last child
                if ((int)tag < n.b) {
                    if ((int)tag + 1 < n.b) stack[(sp++) * 64 + lane] = (uint32_t)e | ((tag + 1u) << 16);
                    stack[(sp++) * 64 + lane] = (uint32_t)S.children[n.a + (int)tag];
                }
                break;
            case RT_NODE_TRANSLATE:
            case RT_NODE_ROTATE_Y:
                stack[(sp++) * 64 + lane] = (uint32_t)e | (END_XF << 16);
                stack[(sp++) * 64 + lane] = (uint32_t)n.a;
                cur = xf_apply(n, cur);
                cur_xf = e;
                inv = d3{1.0f / cur.d.x, 1.0f / cur.d.y, 1.0f / cur.d.z};
                a = len2(cur.d);
                break;
            case RT_NODE_MEDIUM: {  // B2/constant_medium.h:35-52
                double t1, t2;
                if (!boundary_hit(S, n.b, cur, tm, -__builtin_inf(), __builtin_inf(), t1)) break;
                if (!boundary_hit(S, n.b, cur, tm, t1 + 0.0001, __builtin_inf(), t2)) break;
                if (t1 < T_MIN) t1 = T_MIN;
                if (t2 > h.t) t2 = h.t;
                if (t1 >= t2) break;
                if (t1 < 0) t1 = 0;
                const double ray_length = sqrt(len2(cur.d));
                const double distance_inside_boundary = (t2 - t1) * ray_length;
                const u32x4 wd = philox4x32_10(g.pixel, g.sample, g.site, TRY_MEDIUM + (uint32_t)e, g.k0, g.k1);
                const double hit_distance = n.p[0] * log(u01(wd.x));
                if (hit_distance > distance_inside_boundary) break;
                h.t = t1 + hit_distance / ray_length;
                h.node = e;
                h.xf = cur_xf;
                break;
            }
            default: {  // primitives
                double t;
                if (prim_hit(S, n, cur, a, tm, T_MIN, h.t, t)) {
                    h.t = t;
                    h.node = e;
                    h.xf = cur_xf;
                }
                break;
            }
            }
        }
    }
    return h;
}

// ---------------------------------------------------------------- hit record of the winner
struct Surface {
    d3 p, normal, outward;  // outward: sphere's unit outward normal in its own space (for u,v)
    double u, v;
    bool front_face;
    bool sphere_uv;  // u,v still to be derived from `outward` (B2/sphere.h:25-38), done only if a texture needs it
    int32_t mat;
};

__device__ __forceinline__ void set_face_normal(Surface& s, const Ray& r, d3 outward_normal) {  // B2/hittable.h:19-22
    s.front_face = dot(r.d, outward_normal) < 0;
    s.normal = s.front_face ? outward_normal : -outward_normal;
}

__device__ __forceinline__ Surface make_surface(const SceneDev& S, const Hit& h, Ray w, double tm) {
    Surface s;
    s.u = s.v = 0.0;  // moving_sphere and constant_medium leave u,v untouched in the reference
    s.sphere_uv = false;
    s.outward = d3{0, 0, 0};
    const SceneNode n = S.nodes[h.node];
    const Ray local = ray_in(S, h.xf, w);
    s.p = local.o + h.t * local.d;
    s.mat = n.a;
    switch (n.kind) {
    case RT_NODE_SPHERE:
    case RT_NODE_MOVING_SPHERE: {
        const bool moving = n.kind == RT_NODE_MOVING_SPHERE;
        const d3 center = moving ? moving_center(S, n, tm) : d3{n.p[0], n.p[1], n.p[2]};
        const double radius = moving ? S.extra[n.c + 2] : n.p[3];
        const d3 outward_normal = div_by(s.p - center, radius);
        set_face_normal(s, local, outward_normal);
        s.outward = outward_normal;
        s.sphere_uv = !moving;
        break;
    }
    case RT_NODE_XY_RECT:
    case RT_NODE_XZ_RECT:
    case RT_NODE_YZ_RECT: {  // B2/aarect.h:36-43
        double t, ua, vb;
        hit_rect(n, local, h.t, h.t, t, ua, vb);  // same arithmetic as the accepted test: recovers a, b
        s.u = (ua - n.p[0]) / (n.p[1] - n.p[0]);
        s.v = (vb - n.p[2]) / (n.p[3] - n.p[2]);
        const d3 outward_normal = n.kind == RT_NODE_XY_RECT ? d3{0, 0, 1} : (n.kind == RT_NODE_XZ_RECT ? d3{0, 1, 0} : d3{1, 0, 0});
        set_face_normal(s, local, outward_normal);
        break;
    }
    default:  // RT_NODE_MEDIUM, B2/constant_medium.h:61-63
        s.normal = d3{1, 0, 0};
        s.front_face = true;
        break;
    }
    // back out through the enclosing instances, innermost first (B2/hittable.h:48-49, 127-137)
    for (int x = h.xf, lvl = 0; x >= 0 && lvl < XF_DEPTH_CAP; x = S.parent_xf[x], ++lvl) {
        const SceneNode xn = S.nodes[x];
        const Ray child_ray = ray_in(S, x, w);
        if (xn.kind == RT_NODE_TRANSLATE) {
            s.p = s.p + d3{xn.p[0], xn.p[1], xn.p[2]};
            set_face_normal(s, child_ray, s.normal);
        } else {
            const double sin_theta = xn.p[0], cos_theta = xn.p[1];
            d3 p = s.p, normal = s.normal;
            p.x = cos_theta * s.p.x + sin_theta * s.p.z;
            p.z = -sin_theta * s.p.x + cos_theta * s.p.z;
            normal.x = cos_theta * s.normal.x + sin_theta * s.normal.z;
            normal.z = -sin_theta * s.normal.x + cos_theta * s.normal.z;
            s.p = p;
            set_face_normal(s, child_ray, normal);
        }
    }
    return s;
}

// ---------------------------------------------------------------- textures
__device__ __forceinline__ double perlin_noise(const rt_perlin& pn, d3 p) {  // B2/perlin.h:24-48, 86-104
    const double u = p.x - floor(p.x), v = p.y - floor(p.y), w = p.z - floor(p.z);
    const int i = (int)floor(p.x), j = (int)floor(p.y), k = (int)floor(p.z);
    const double uu = u * u * (3 - 2 * u), vv = v * v * (3 - 2 * v), ww = w * w * (3 - 2 * w);
    double accum = 0.0;
#pragma unroll
    for (int di = 0; di < 2; ++di)
#pragma unroll
        for (int dj = 0; dj < 2; ++dj)
#pragma unroll
            for (int dk = 0; dk < 2; ++dk) {
                const int idx = pn.perm_x[(i + di) & 255] ^ pn.perm_y[(j + dj) & 255] ^ pn.perm_z[(k + dk) & 255];
                const d3 c{pn.ranvec[idx][0], pn.ranvec[idx][1], pn.ranvec[idx][2]};
                const d3 weight_v{u - di, v - dj, w - dk};
                accum += (di * uu + (1 - di) * (1 - uu)) * (dj * vv + (1 - dj) * (1 - vv)) *
                         (dk * ww + (1 - dk) * (1 - ww)) * dot(c, weight_v);
            }
    return accum;
}

__device__ __forceinline__ double perlin_turb(const rt_perlin& pn, d3 p) {  // B2/perlin.h:50-61
    double accum = 0.0, weight = 1.0;
    d3 temp_p = p;
    for (int i = 0; i < 7; ++i) {
        accum += weight * perlin_noise(pn, temp_p);
        weight *= 0.5;
        temp_p = d3{temp_p.x * 2, temp_p.y * 2, temp_p.z * 2};
    }
    return fabs(accum);
}

__device__ __forceinline__ double clampd(double x, double lo, double hi) {  // B2/rtweekend.h:47-51
    if (x < lo) return lo;
    if (x > hi) return hi;
    return x;
}

// texture::value (B2/texture.h).  Checker textures are followed iteratively (at most 4 levels).
__device__ __forceinline__ d3 texture_value(const SceneDev& S, int ti, Surface& s) {
    for (int level = 0; level < 4; ++level) {
        const TextureDev t = S.textures[ti];
        if (t.kind == RT_TEX_CHECKER) {
            const double sines = sin(10 * s.p.x) * sin(10 * s.p.y) * sin(10 * s.p.z);
            ti = sines < 0 ? t.b : t.a;
            continue;
        }
        if (t.kind == RT_TEX_SOLID) return d3{t.p[0], t.p[1], t.p[2]};
        if (t.kind == RT_TEX_NOISE) {
            const double f = 1 + sin(t.p[0] * s.p.z + 10 * perlin_turb(S.perlins[t.a], s.p));
            return f * (0.5 * d3{1, 1, 1});
        }
        // RT_TEX_IMAGE, B2/texture.h:85-104
        if (t.a < 0 || S.images[t.a].rgb == nullptr) return d3{0, 1, 1};
        if (s.sphere_uv) {  // get_sphere_uv, B2/sphere.h:33-37
            const double pi = 3.14159265358979323846;
            const double theta = acos(-s.outward.y);
            const double phi = atan2(-s.outward.z, s.outward.x) + pi;
            s.u = phi / (2 * pi);
            s.v = theta / pi;
            s.sphere_uv = false;
        }
        const ImageDev im = S.images[t.a];
        const double u = clampd(s.u, 0.0, 1.0);
        const double v = 1.0 - clampd(s.v, 0.0, 1.0);
        int i = (int)(u * im.width), j = (int)(v * im.height);
        if (i >= im.width) i = im.width - 1;
        if (j >= im.height) j = im.height - 1;
        const double color_scale = 1.0 / 255.0;
        const uint8_t* px = im.rgb + (size_t)j * 3 * (size_t)im.width + (size_t)i * 3;
        return d3{color_scale * px[0], color_scale * px[1], color_scale * px[2]};
    }
    return d3{0, 0, 0};
}

__device__ __forceinline__ Ray primary_ray(const CameraDev& C, int W, int H, uint32_t pixel, uint32_t sample, uint32_t k0,
                                           uint32_t k1, double& tm) {
    const int i = (int)(pixel % (uint32_t)W);
    const int j = H - 1 - (int)(pixel / (uint32_t)W);
    Ray r;
    camera_ray(C, W, H, i, j, pixel, sample, k0, k1, r.o, r.d);
    tm = C.time0;
    if (C.time0 != C.time1) {  // B2/camera.h:44; with equal times the draw would not change the value
        const u32x4 w = philox4x32_10(pixel, sample, 0u, TRY_TIME, k0, k1);
        tm = u_range(w.x, C.time0, C.time1);
    }
    return r;
}

}  // namespace

// ---------------------------------------------------------------- render kernel
__global__ void __launch_bounds__(256) k_render_scene(const SceneRenderArgs A) {
    __shared__ uint32_t s_stack[4][STACK_CAP * 64];
    const int lane = threadIdx.x & 63;
    uint32_t* stack = s_stack[threadIdx.x >> 6];
    const SceneDev& S = A.scene;
    const uint32_t npix = (uint32_t)A.width * (uint32_t)A.height;
    const ItemPlan plan = make_item_plan(A.sample_count, A.item_cap);

    bool active = false;
    uint32_t bounce = 0;
    int depth_left = 0;
    Ray r{d3{0, 0, 0}, d3{0, 0, 1}};
    d3 T{1, 1, 1};
    double tm = 0.0;
    LaneWork w{};
    WaveWork wq{};
    const WorkQueues Q{A.item_counter, A.n_items, A.accum};
    uint32_t n_samples = 0, n_rays = 0, n_bad = 0;

    for (;;) {
        if (!refill_work(Q, plan, npix, A.sample_begin, lane, active, w, wq)) break;
        if (!active && w.has_item) {
            r = primary_ray(A.cam, A.width, A.height, w.pixel, (uint32_t)w.s_cur, A.key0, A.key1, tm);
            T = d3{1, 1, 1};
            depth_left = A.max_depth;
            bounce = 0;
            active = true;
            ++n_samples;
        }
        const bool on = active;
        const RngCtx g{w.pixel, (uint32_t)w.s_cur, 1u + bounce, A.key0, A.key1};
        const Hit h = trace(S, r, tm, on, stack, lane, g);

        if (on) {
            ++n_rays;
            bool done = false;
            d3 L_end{0, 0, 0};
            if (h.node < 0) {
                if (S.sky_mode == RT_SKY_GRADIENT) {
                    const d3 unit_direction = unit_vector(r.d);
                    const double t = 0.5 * (unit_direction.y + 1.0);
                    L_end = (1.0 - t) * d3{1.0, 1.0, 1.0} + t * d3{0.5, 0.7, 1.0};
                } else {
                    L_end = d3{S.background[0], S.background[1], S.background[2]};
                }
                done = true;
            } else {
                Surface s = make_surface(S, h, r, tm);
                const MaterialDev m = S.materials[s.mat];
                d3 attenuation{m.ax, m.ay, m.az};
                d3 nd{0, 0, 0};
                const bool wants_sphere = m.kind == RT_MAT_LAMBERTIAN || m.kind == RT_MAT_METAL || m.kind == RT_MAT_ISOTROPIC;
                const d3 rs = random_in_unit_sphere(wants_sphere, g.pixel, g.sample, g.site, A.key0, A.key1);
                if (m.kind == RT_MAT_LAMBERTIAN) {  // B2/material.h:25-36
                    nd = s.normal + unit_vector(rs);
                    const double eps = 1e-8;
                    if ((fabs(nd.x) < eps) && (fabs(nd.y) < eps) && (fabs(nd.z) < eps)) nd = s.normal;
                    attenuation = texture_value(S, m.tex, s);
                } else if (m.kind == RT_MAT_METAL) {  // B2/material.h:46-53
                    const d3 reflected = reflect(unit_vector(r.d), s.normal);
                    nd = reflected + m.param * rs;
                    if (!(dot(nd, s.normal) > 0)) done = true;  // no scatter, nothing emitted
                } else if (m.kind == RT_MAT_DIELECTRIC) {  // B2/material.h:64-83
                    attenuation = d3{1.0, 1.0, 1.0};
                    const double refraction_ratio = s.front_face ? (1.0 / m.param) : m.param;
                    const d3 unit_direction = unit_vector(r.d);
                    const double cos_theta = fmin(dot(-unit_direction, s.normal), 1.0);
                    const double sin_theta = sqrt(1.0 - cos_theta * cos_theta);
                    bool do_reflect = refraction_ratio * sin_theta > 1.0;
                    if (!do_reflect) {
                        const u32x4 w = philox4x32_10(g.pixel, g.sample, g.site, 0u, A.key0, A.key1);
                        do_reflect = reflectance(cos_theta, refraction_ratio) > u01(w.x);
                    }
                    nd = do_reflect ? reflect(unit_direction, s.normal) : refract(unit_direction, s.normal, refraction_ratio);
                } else if (m.kind == RT_MAT_ISOTROPIC) {  // B2/material.h:121-127
                    nd = rs;
                    attenuation = texture_value(S, m.tex, s);
                } else {  // diffuse_light: emits, does not scatter (B2/material.h:102-110)
                    L_end = texture_value(S, m.tex, s);
                    done = true;
                }
                if (!done) {
                    T = T * attenuation;
                    r.o = s.p;
                    r.d = nd;
                    ++bounce;
                    if (--depth_left <= 0) done = true;
                }
            }
            if (done) {
                add_sample(T * L_end, w.fx, w.fy, w.fz, n_bad);
                active = false;
                ++w.s_cur;
            }
        }
    }

    uint64_t v0 = n_samples, v1 = n_rays, v2 = n_bad;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        v0 += __shfl_xor(v0, off);
        v1 += __shfl_xor(v1, off);
        v2 += __shfl_xor(v2, off);
    }
    if (lane == 0) {
        atomicAdd((unsigned long long*)&A.stats[0], (unsigned long long)v0);
        atomicAdd((unsigned long long*)&A.stats[1], (unsigned long long)v1);
        atomicAdd((unsigned long long*)&A.stats[2], (unsigned long long)v2);
    }
}

// ---------------------------------------------------------------- first-hit kernel
__global__ void __launch_bounds__(256) k_first_hit_scene(const SceneFirstHitArgs A) {
    __shared__ uint32_t s_stack[4][STACK_CAP * 64];
    const int lane = threadIdx.x & 63;
    uint32_t* stack = s_stack[threadIdx.x >> 6];
    const uint32_t npix = (uint32_t)A.width * (uint32_t)A.height;
    const uint32_t pixel = blockIdx.x * 256u + threadIdx.x;
    const bool on = pixel < npix;
    const uint32_t pc = on ? pixel : 0u;
    const int i = (int)(pc % (uint32_t)A.width);
    const int j = A.height - 1 - (int)(pc / (uint32_t)A.width);
    const CameraDev& C = A.cam;
    const double s = (i + 0.5) / (A.width - 1);
    const double t = (j + 0.5) / (A.height - 1);
    const d3 cu{C.u[0], C.u[1], C.u[2]}, cv{C.v[0], C.v[1], C.v[2]};
    const d3 offset = 0.0 * cu + 0.0 * cv;
    const d3 origin{C.origin[0], C.origin[1], C.origin[2]};
    const d3 llc{C.llc[0], C.llc[1], C.llc[2]};
    const d3 hor{C.horizontal[0], C.horizontal[1], C.horizontal[2]};
    const d3 ver{C.vertical[0], C.vertical[1], C.vertical[2]};
    const Ray r{origin + offset, llc + s * hor + t * ver - origin - offset};
    const RngCtx g{pc, 0u, 1u, 0u, 0u};
    const Hit h = trace(A.scene, r, C.time0, on, stack, lane, g);
    if (on) A.ids[pixel] = h.node < 0 ? -1 : A.scene.top_of[h.node];
}

hipError_t launch_render_scene(const SceneRenderArgs& args, int blocks, hipStream_t stream) {
    hipLaunchKernelGGL(k_render_scene, dim3((unsigned)blocks), dim3(256), 0, stream, args);
    return hipGetLastError();
}

hipError_t launch_first_hit_scene(const SceneFirstHitArgs& args, hipStream_t stream) {
    const uint32_t npix = (uint32_t)args.width * (uint32_t)args.height;
    hipLaunchKernelGGL(k_first_hit_scene, dim3((npix + 255u) / 256u), dim3(256), 0, stream, args);
    return hipGetLastError();
}

int render_scene_blocks_per_cu() {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_render_scene, 256, 0) != hipSuccess || n < 1) n = 2;
    return n;
}

}  // namespace rtd
