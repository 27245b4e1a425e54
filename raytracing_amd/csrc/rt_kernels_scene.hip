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
//
// Three levels of the walk, from the outside in:
//   1. the world list is walked by the wave as a whole (all lanes enter and leave its k-th object together);
//   2. an object that is a primitive, a box, a medium or a leaf set -- by itself or under an instance chain -- is
//      evaluated right there from a per-object table read with scalar loads; a leaf set (a BVH subtree of spheres and
//      boxes whose node boxes enclose their leaves) is screened in float by the whole wave and only its few
//      candidates are tested in double (rt_sphere_screen.h, rt_box_screen.h);
//   3. everything else (lists, BVH trees with other leaves) goes through the per-lane stack.
#include <cstdio>

#include "rt_device_common.h"
#include "rt_internal.h"
#include "rt_render_common.h"
#include "rt_box_screen.h"
#include "rt_sphere_screen.h"

namespace rtd {

namespace {

// Stack entry: node id in the low 16 bits; high 16 bits = END_XF (leave that instance node) or, for a
// list / box, the index of the next child to visit (a list occupies one slot however long it is).
constexpr uint32_t END_XF = 0xFFFFu;            // leave a rotate_y node
constexpr uint32_t END_XF_TRANSLATE = 0xFFFEu;  // leave a translate node

// Scene features a kernel variant is compiled for (template parameter F).  A world that has no instances,
// media, rectangles ... runs a variant without that code: fewer registers (more waves per SIMD to hide
// the latency of the dependent node loads) and fewer divergent cases in the traversal loop.
enum : int {
    F_MOVING = 1,    // moving spheres
    F_RECT = 2,      // rectangles and boxes
    F_XF = 4,        // translate / rotate_y instances
    F_MEDIUM = 8,    // constant media and the isotropic material
    F_CHECKER = 16,  // checker textures
    F_NOISE = 32,    // Perlin noise textures
    F_IMAGE = 64,    // image textures
    F_ALL = 127,
    F_IS = 128,      // ray_color of book 3: light sampled by importance (RT_RENDER_IMPORTANCE); not part of F_ALL
    F_FLIP = 256,    // flip_face wrappers (book 3); not part of F_ALL, implies the instance machinery (F_XF)
    F_SET = 512,     // BVH subtrees of spheres / boxes evaluated as leaf sets (DEV_NODE_LEAF_SET); not part of F_ALL
    F_STACK = 1024,  // some object of the world is walked through the per-lane stack (trace()); not part of F_ALL
    F_EVERY = F_ALL | F_FLIP | F_SET | F_STACK,
};
static_assert(CAND_CAP * 32 + ROW_WORDS * 64 <= STACK_CAP * 64, "the candidate lists of a leaf set use the stack's memory (a set is evaluated while the stack is empty)");
constexpr double T_MIN = 0.001;      // B2/RayTracing.cpp:25
// In the variants whose worlds are walked by the wave as a whole (no per-lane stack), Hit::xf names the world's object
// the hit lies in (an index into SceneDev::tops; -1 when no instance stands above it) instead of the innermost instance node.
template <int F>
constexpr bool TOP_CONTEXT = (F & F_XF) != 0 && (F & F_STACK) == 0;

struct Ray {
    d3 o, d;
};

__device__ __forceinline__ double comp(d3 v, int axis) { return axis == 0 ? v.x : (axis == 1 ? v.y : v.z); }

// Ray handed to the child of a translate / rotate_y node (B2/hittable.h:44, 114-123).
// (A flip_face node hands its ray on unchanged: callers compiled with F_FLIP check for it before calling.)
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
template <int F>
__device__ __forceinline__ Ray ray_in(const SceneDev& S, int xf, Ray world) {
    if constexpr (!(F & F_XF)) return world;
    int c0 = -1, c1 = -1, c2 = -1, c3 = -1;
    int x = xf;
    if (x >= 0) { c0 = x; x = S.parent_xf[x]; }
    if (x >= 0) { c1 = x; x = S.parent_xf[x]; }
    if (x >= 0) { c2 = x; x = S.parent_xf[x]; }
    if (x >= 0) { c3 = x; }
    Ray r = world;
    const int chain[4] = {c3, c2, c1, c0};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (chain[k] < 0) continue;
        const SceneNode n = S.nodes[chain[k]];
        if ((F & F_FLIP) != 0 && n.kind == RT_NODE_FLIP_FACE) continue;
        r = xf_apply(n, r);
    }
    return r;
}

// sphere::hit up to the accepted root (B2/sphere.h:42-60), in two steps so that callers that ask twice about the
// same ray and sphere (constant_medium) share the arithmetic: both roots, then the reference's choice between them.
struct SphereRoots {
    bool real;  // discriminant >= 0
    double near_root, far_root;
};
__device__ __forceinline__ SphereRoots sphere_roots(d3 center, double radius, const Ray& r, double a) {
    const d3 oc = r.o - center;
    const double half_b = dot(oc, r.d);
    const double c = len2(oc) - radius * radius;
    const double discriminant = half_b * half_b - a * c;
    SphereRoots s{false, 0.0, 0.0};
    if (discriminant < 0) return s;
    const double sqrtd = sqrt(discriminant);
    s.real = true;
    s.near_root = (-half_b - sqrtd) / a;
    s.far_root = (-half_b + sqrtd) / a;
    return s;
}
__device__ __forceinline__ bool pick_root(const SphereRoots& s, double t_min, double t_max, double& t) {
    if (!s.real) return false;
    double root = s.near_root;
    if (root < t_min || root > t_max) {
        root = s.far_root;
        if (root < t_min || root > t_max) return false;
    }
    t = root;
    return true;
}
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
    return c0 + motion_fraction(tm, time0, time1) * (c1 - c0);
}

// xy_rect / xz_rect / yz_rect ::hit up to acceptance (B2/aarect.h:28-34, 70-76, 112-118): the plane is axis ka = k,
// the rectangle spans [a0,a1] on axis aa and [b0,b1] on axis ba.
__device__ __forceinline__ bool hit_rect_raw(int ka, int aa, int ba, double a0, double a1, double b0, double b1, double k,
                                             const Ray& r, double t_min, double t_max, double& t, double& ua, double& vb) {
    const double tt = (k - comp(r.o, ka)) / comp(r.d, ka);
    if (tt < t_min || tt > t_max) return false;
    const double a = comp(r.o, aa) + tt * comp(r.d, aa);
    const double b = comp(r.o, ba) + tt * comp(r.d, ba);
    if (a < a0 || a > a1 || b < b0 || b > b1) return false;
    t = tt;
    ua = a;
    vb = b;
    return true;
}
__device__ __forceinline__ bool hit_rect(const SceneNode& n, const Ray& r, double t_min, double t_max, double& t,
                                         double& ua, double& vb) {
    const int ka = n.kind == RT_NODE_XY_RECT ? 2 : (n.kind == RT_NODE_XZ_RECT ? 1 : 0);
    const int aa = n.kind == RT_NODE_YZ_RECT ? 1 : 0;
    const int ba = n.kind == RT_NODE_XY_RECT ? 1 : 2;
    return hit_rect_raw(ka, aa, ba, n.p[0], n.p[1], n.p[2], n.p[3], n.p[4], r, t_min, t_max, t, ua, vb);
}

// box::hit (B2/box.h:41-43 = hittable_list::hit over the six sides of B2/box.h:31-38) for a box whose sides are
// exactly those the reference's constructor derives from (p0, p1) and whose side nodes are numbered consecutively
// (checked at upload): the six rectangle tests in the reference's order with the closest-so-far value, from the two
// corners alone.  Returns the index 0..5 of the winning side, or -1.
__device__ __forceinline__ int hit_box6(const double* p, const Ray& r, double t_min, double& t_max) {
    const double x0 = p[0], y0 = p[1], z0 = p[2], x1 = p[3], y1 = p[4], z1 = p[5];
    int side = -1;
    double t, ua, vb;
    if (hit_rect_raw(2, 0, 1, x0, x1, y0, y1, z1, r, t_min, t_max, t, ua, vb)) { t_max = t; side = 0; }
    if (hit_rect_raw(2, 0, 1, x0, x1, y0, y1, z0, r, t_min, t_max, t, ua, vb)) { t_max = t; side = 1; }
    if (hit_rect_raw(1, 0, 2, x0, x1, z0, z1, y1, r, t_min, t_max, t, ua, vb)) { t_max = t; side = 2; }
    if (hit_rect_raw(1, 0, 2, x0, x1, z0, z1, y0, r, t_min, t_max, t, ua, vb)) { t_max = t; side = 3; }
    if (hit_rect_raw(0, 1, 2, y0, y1, z0, z1, x1, r, t_min, t_max, t, ua, vb)) { t_max = t; side = 4; }
    if (hit_rect_raw(0, 1, 2, y0, y1, z0, z1, x0, r, t_min, t_max, t, ua, vb)) { t_max = t; side = 5; }
    return side;
}

template <int F>
__device__ __forceinline__ bool prim_hit(const SceneDev& S, const SceneNode& n, const Ray& r, double a, double tm,
                                         double t_min, double t_max, double& t) {
    if (n.kind == RT_NODE_SPHERE || n.kind == RT_NODE_MOVING_SPHERE) {  // one copy of the quadratic for both
        d3 center{n.p[0], n.p[1], n.p[2]};
        double radius = n.p[3];
        if constexpr ((F & F_MOVING) != 0) {
            if (n.kind == RT_NODE_MOVING_SPHERE) {
                center = moving_center(S, n, tm);
                radius = S.extra[n.c + 2];
            }
        }
        return hit_sphere(center, radius, r, a, t_min, t_max, t);
    }
    if constexpr ((F & F_RECT) != 0) {
        if (n.kind == RT_NODE_XY_RECT || n.kind == RT_NODE_XZ_RECT || n.kind == RT_NODE_YZ_RECT) {
            double ua, vb;
            return hit_rect(n, r, t_min, t_max, t, ua, vb);
        }
    }
    return false;
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
template <int F>
__device__ __forceinline__ bool boundary_hit(const SceneDev& S, int node, Ray r, double tm, double t_min, double t_max,
                                             double& t_out) {
    SceneNode n = S.nodes[node];
    for (int lvl = 0; lvl < XF_DEPTH_CAP && (n.kind == RT_NODE_TRANSLATE || n.kind == RT_NODE_ROTATE_Y); ++lvl) {
        r = xf_apply(n, r);
        n = S.nodes[n.a];
    }
    const double a = len2(r.d);
    if constexpr ((F & F_RECT) != 0) {
        if (n.kind == DEV_NODE_BOX6) {  // the six sides from the corners: the list's tests, order and closest-so-far rule
            double best = t_max;
            const bool any = hit_box6(n.p, r, t_min, best) >= 0;
            t_out = best;
            return any;
        }
    }
    if (n.kind == RT_NODE_LIST || n.kind == RT_NODE_BOX || n.kind == DEV_NODE_BOX6) {
        bool any = false;
        double best = t_max;
        for (int i = 0; i < n.b; ++i) {
            const SceneNode c = S.nodes[S.children[n.a + i]];
            double t;
            if (prim_hit<F>(S, c, r, a, tm, t_min, best, t)) {
                best = t;
                any = true;
            }
        }
        t_out = best;
        return any;
    }
    return prim_hit<F>(S, n, r, a, tm, t_min, t_max, t_out);
}

struct Hit {
    double t;
    int32_t node;  // winning primitive or medium node, -1 = miss
    int32_t xf;    // instance node enclosing it, -1 = none
};

struct RngCtx {
    uint32_t pixel, sample, site, k0, k1;
};

// constant_medium::hit (B2/constant_medium.h:30-66) for medium node `e`; updates h when the ray scatters inside.
template <int F>
__device__ __forceinline__ void medium_hit(const SceneDev& S, const SceneNode& n, int e, const Ray& cur, int cur_xf,
                                           double tm, const RngCtx& g, Hit& h) {
    double t1, t2;
    const SceneNode bn = S.nodes[n.b];
    if (bn.kind == RT_NODE_SPHERE) {  // both boundary searches from one evaluation of the quadratic
        const SphereRoots sr = sphere_roots(d3{bn.p[0], bn.p[1], bn.p[2]}, bn.p[3], cur, len2(cur.d));
        if (!pick_root(sr, -__builtin_inf(), __builtin_inf(), t1)) return;
        if (!pick_root(sr, t1 + 0.0001, __builtin_inf(), t2)) return;
    } else {
        if (!boundary_hit<F>(S, n.b, cur, tm, -__builtin_inf(), __builtin_inf(), t1)) return;
        if (!boundary_hit<F>(S, n.b, cur, tm, t1 + 0.0001, __builtin_inf(), t2)) return;
    }
    if (t1 < T_MIN) t1 = T_MIN;
    if (t2 > h.t) t2 = h.t;
    if (t1 >= t2) return;
    if (t1 < 0) t1 = 0;
    const double ray_length = sqrt(len2(cur.d));
    const double distance_inside_boundary = (t2 - t1) * ray_length;
    const u32x4 wd = philox4x32_10(g.pixel, g.sample, g.site, TRY_MEDIUM + (uint32_t)e, g.k0, g.k1);
    const double hit_distance = n.p[0] * log(u01(wd.x));
    if (hit_distance > distance_inside_boundary) return;
    h.t = t1 + hit_distance / ray_length;
    h.node = e;
    h.xf = cur_xf;
}

// world.hit(r, 0.001, infinity, rec) for one ray per lane.  All 64 lanes must call it.
// One entry per lane and trip, whatever its kind: letting lanes with a leaf wait for the others to reach
// one (separate box and leaf phases) was measured 1.2-1.6x slower -- the loop is bound by the latency of
// the dependent stack and node loads, not by the instructions of the cases.  (Testing a box's six sides in one
// trip instead of walking them through the stack was also measured slower, for the same reason.)
#ifdef RT_SCENE_PROBE
// debug build: per-lane counters of the traversal loop, summed and printed by the render kernel
static __device__ unsigned long long g_probe_total[16];  // [kind] entries, [12] trips, [13] lane-trips, [14] traces
// wave cycles per section of the render loop
enum { PS_REFILL = 0, PS_TOP_PRIM, PS_TOP_BOX, PS_TOP_MEDIUM, PS_SET_SPHERES, PS_SET_BOXES, PS_STACK_WALK, PS_SURFACE, PS_RANDOM, PS_TEXTURE, PS_SHADE, PS_COUNT };
static __device__ unsigned long long g_probe_cycles[PS_COUNT];
static __device__ unsigned long long g_probe_mine[2][6];  // visits of a leaf set by the lanes that reached it: 1-4, 5-8, 9-16, 17-32, 33-64; [5] = lanes
// The cycles are summed per wave in registers and added to the device-wide counters once, when the wave ends: an
// atomic per section and trip from thousands of waves to one address takes longer than the sections themselves and
// lands in whichever section waits for memory next.  -DRT_SCENE_PROBE=2 adds the event counts (atomics: small frames).
struct ProbeAcc {
    unsigned long long cycles[PS_COUNT];
    unsigned long long t;
};
#define PROBE_PARAM , ProbeAcc& probe
#define PROBE_PASS , probe
#define PROBE_MARK(k) do { const unsigned long long now_ = clock64(); probe.cycles[k] += now_ - probe.t; probe.t = now_; } while (0)
#define PROBE_COUNTS (RT_SCENE_PROBE + 0 >= 2)
#else
#define PROBE_PARAM
#define PROBE_PASS
#define PROBE_MARK(k) do { } while (0)
#define PROBE_COUNTS 0
#endif
// bvh_node::hit (B2/bvh.h:97-106) over the subtree of leaf set `set_index`, without the recursion: its nodes in
// depth-first order, a node whose box test fails left by its skip link.  For the rays the flat form is not for.  Out of
// line: it runs for a handful of rays in a launch, and inline its registers would be the render loop's.
template <int F>
__device__ __noinline__ Hit replay_subtree(const LeafSetDev* sets, const int32_t* set_nodes, int set_index, int first_slot, Ray loc,
                                           int loc_xf, Hit h) {
    const LeafSetDev set = sets[set_index];
    const d3 loc_inv{1.0f / loc.d.x, 1.0f / loc.d.y, 1.0f / loc.d.z};
    const double loc_a = len2(loc.d);
    for (int at = 0; at < set.n_replay;) {
        const ReplayNode rn = set.replay[at];
        if (rn.member < 0) {
            at = hit_aabb(rn.box, loc, loc_inv, T_MIN, h.t) ? at + 1 : rn.skip;
            continue;
        }
        ++at;
        if (rn.member < set.spheres.n) {
            const SphereExact se = set.spheres.exact[rn.member];
            double t;
            if (hit_sphere(d3{se.cx, se.cy, se.cz}, se.radius, loc, loc_a, T_MIN, h.t, t)) {
                h.t = t;
                h.node = set_nodes[first_slot + rn.member];
                h.xf = loc_xf;
            }
        } else if constexpr ((F & F_RECT) != 0) {
            const BoxExact be = set.boxes.exact[rn.member - set.spheres.n];
            const int side = hit_box6(be.corner, loc, T_MIN, h.t);
            if (side >= 0) {
                h.node = be.first_side + side;
                h.xf = loc_xf;
            }
        }
    }
    return h;
}

template <int F>
__device__ __forceinline__ Hit trace(const SceneDev& S, Ray w, double tm, bool on, bool exact_only, uint32_t* stack,
                                     int lane, const RngCtx& g PROBE_PARAM) {
#ifdef RT_SCENE_PROBE
    probe.t = clock64();
#endif
    Hit h{__builtin_inf(), -1, -1};
    Ray cur = w;
    int cur_xf = -1;
    d3 inv{1.0f / cur.d.x, 1.0f / cur.d.y, 1.0f / cur.d.z};
    double a = len2(cur.d);
#if PROBE_COUNTS
    if (lane == 0) atomicAdd(&g_probe_total[14], 1ull);
#endif
    // The world (a list, checked at upload) is walked by the wave as a whole: all lanes enter its k-th object
    // together and leave it together.  Each lane still meets the objects in list order with its own
    // closest-so-far value, but the lanes of a trip are now inside the same object: a sphere or a medium of the
    // world is tested by all lanes at once, and inside a BVH only box code and its leaves' code are live.
    typedef __attribute__((address_space(4))) TopEntry ConstTopEntry;  // immutable for the launch: scalar loads
    const ConstTopEntry* tops = (const ConstTopEntry*)S.tops;
    for (int top = 0; top < S.n_tops; ++top) {
    const int top_node = tops[top].first;
    {   // An object of the world that is a primitive, a box, a medium or a leaf set, by itself or under a chain of
        // instances, is the same node for every lane: its data come through scalar loads and nothing goes through
        // the stack.
        const int node = tops[top].node, chain = tops[top].chain_len;
        const int kind = tops[top].end.kind;
        const bool plain = kind >= RT_NODE_SPHERE && kind <= RT_NODE_YZ_RECT;
        const bool box6 = (F & F_RECT) != 0 && kind == DEV_NODE_BOX6;
        const bool medium = (F & F_MEDIUM) != 0 && kind == RT_NODE_MEDIUM;
        const bool leaf_set = (F & F_SET) != 0 && kind == DEV_NODE_LEAF_SET;
        if (plain || box6 || medium || leaf_set) {
            Ray loc = w;
            int loc_xf = -1;
            if constexpr ((F & F_XF) != 0) {
                for (int i = 0; i < chain; ++i) {  // the rays handed down by translate::hit / rotate_y::hit, outermost first
                    SceneNode xn;
                    xn.kind = tops[top].chain_kind[i];
                    xn.p[0] = tops[top].chain_p[i][0];
                    xn.p[1] = tops[top].chain_p[i][1];
                    xn.p[2] = tops[top].chain_p[i][2];
                    if (!((F & F_FLIP) != 0 && xn.kind == RT_NODE_FLIP_FACE)) loc = xf_apply(xn, loc);
                    loc_xf = tops[top].chain[i];
                }
            }
            const int ctx = TOP_CONTEXT<F> ? (chain > 0 ? top : -1) : loc_xf;  // what Hit::xf records
            SceneNode n;
            n.kind = kind;
            n.a = tops[top].end.a;
            n.b = tops[top].end.b;
            n.c = tops[top].end.c;
#pragma unroll
            for (int k = 0; k < 6; ++k) n.p[k] = tops[top].end.p[k];
            if (plain) {
                double t;
                if (on && prim_hit<F>(S, n, loc, len2(loc.d), tm, T_MIN, h.t, t)) {
                    h.t = t;
                    h.node = node;
                    h.xf = ctx;
                }
                PROBE_MARK(PS_TOP_PRIM);
            } else if (box6) {
                if constexpr ((F & F_RECT) != 0) {
                    const int side = on ? hit_box6(n.p, loc, T_MIN, h.t) : -1;
                    if (side >= 0) {
                        h.node = n.c + side;
                        h.xf = ctx;
                    }
                }
                PROBE_MARK(PS_TOP_BOX);
            } else if (medium) {
                if constexpr ((F & F_MEDIUM) != 0) {
                    if (on) medium_hit<F>(S, n, node, loc, ctx, tm, g, h);
                }
                PROBE_MARK(PS_TOP_MEDIUM);
            } else {
                if constexpr ((F & F_SET) != 0) {
                    // bvh_node::hit's own box test (B2/bvh.h:98-99), then the subtree's leaves as a flat set: the
                    // subtree keeps the smallest root in [t_min, closest so far] whatever order its leaves are tested
                    // in (B2/bvh.h:97-106), and so does the list
                    const d3 loc_inv{1.0f / loc.d.x, 1.0f / loc.d.y, 1.0f / loc.d.z};
                    const bool reached = on && hit_aabb(n.p, loc, loc_inv, T_MIN, h.t);
                    // A ray the float screens cannot hold -- a NaN, an infinity, products that overflow: the reference's
                    // comparisons then come out false and the order of the tests decides everything -- walks the subtree
                    // itself (replay_subtree); the flat form is for the others.  Only in the variants that carry the stack
                    // loop as well: a call in the stack-free variants costs their render loop registers (final scene 42 ->
                    // 46 ms), and their sets hold one kind of leaf (rt_scene_upload sends worlds with mixed sets here), for
                    // which testing every member in order, as closest_hit_with does for such rays, gives the walk's answer.
                    bool mine = reached;
                    if constexpr ((F & F_STACK) != 0) {
                        const double loc_a_ = len2(loc.d);
                        const bool sane = fabs(loc.o.x) <= 1e15 && fabs(loc.o.y) <= 1e15 && fabs(loc.o.z) <= 1e15 && loc_a_ >= 1e-200 && loc_a_ <= 1e200;
                        mine = reached && sane;
                        if (ballot(reached && !sane) != 0ull) {
                            if (reached && !sane) h = replay_subtree<F>(S.sets, S.set_nodes, n.a, n.b, loc, loc_xf, h);
                        }
                    }
                    if (ballot(mine) != 0ull) {
                        const LeafSetDev set = S.sets[n.a];
#if PROBE_COUNTS
                        const int np = __popcll(ballot(mine));
                        if (lane == 0) {
                            const int kind_of_set = set.spheres.n > 0 ? 0 : 1;
                            atomicAdd(&g_probe_mine[kind_of_set][np <= 4 ? 0 : np <= 8 ? 1 : np <= 16 ? 2 : np <= 32 ? 3 : 4], 1ull);
                            atomicAdd(&g_probe_mine[kind_of_set][5], (unsigned long long)np);
                        }
#endif
                        const int first_slot = n.b;
                        const double loc_a = len2(loc.d);
                        uint16_t* list = (uint16_t*)stack;
                        uint32_t* rows = stack + CAND_CAP * 32;
                        int winner = -1, sphere_j = 0;
                        double best_t = h.t;
                        if (set.spheres.n > 0) {
                            uint32_t overflowed = 0;
                            ProbeClock clk;
                            const SetHit sh = closest_hit_with(set.spheres, loc.o, loc.d, best_t, mine, exact_only, list, rows, lane, overflowed, clk,
                                                               [&](int j, SetHit& best) {
                                // (the set's own copies of centre and radius: no walk from the node list to the node)
#if PROBE_COUNTS
                                atomicAdd(&g_probe_total[1], 1ull);  // double-precision sphere tests of leaf sets
#endif
                                const SphereExact se = set.spheres.exact[j];
                                const int id = S.set_nodes[first_slot + j];
                                double t;
                                if (hit_sphere(d3{se.cx, se.cy, se.cz}, se.radius, loc, loc_a, T_MIN, best.t, t)) {
                                    best.t = t;
                                    best.id = j;
                                    best.aux = id;
                                }
                            }, [&](int j, d3 ro, d3 rd, double t_max, double, double& t) -> int {  // the same for a ray of another lane
#if PROBE_COUNTS
                                atomicAdd(&g_probe_total[1], 1ull);
#endif
                                const SphereExact se = set.spheres.exact[j];
                                const Ray ray{ro, rd};
                                return hit_sphere(d3{se.cx, se.cy, se.cz}, se.radius, ray, len2(rd), T_MIN, t_max, t) ? S.set_nodes[first_slot + j] : -1;
                            });
                            best_t = sh.t;
                            if (sh.id >= 0) {
                                winner = sh.aux;
                                sphere_j = sh.id;
                            }
#if PROBE_COUNTS
                            if (overflowed) atomicAdd(&g_probe_total[15], 1ull);  // rays with more candidates than slots
#endif
                            PROBE_MARK(PS_SET_SPHERES);
                        }
                        if constexpr ((F & F_RECT) != 0) {
                            if (set.boxes.n > 0) {
                                const int first_box = first_slot + set.spheres.n;
                                const SetHit bh = closest_box_hit(set.boxes, loc.o, loc.d, best_t, mine, exact_only, list, rows, lane,
                                                                  [&](int j, d3 ro, d3 rd, double t_max, double& t) -> int {
#if PROBE_COUNTS
                                    atomicAdd(&g_probe_total[6], 1ull);  // double-precision box tests of leaf sets
#endif
                                    const BoxExact be = set.boxes.exact[j];
                                    const Ray ray{ro, rd};
                                    t = t_max;
                                    const int side = hit_box6(be.corner, ray, T_MIN, t);
                                    return side >= 0 ? be.first_side + side : -1;
                                });
                                // A box side hit at exactly the sphere's t: bvh_node::hit accepts an equal root
                                // (B2/bvh.h:99-100, B2/sphere.h:54), so the leaf the tree reaches LATER wins -- the
                                // sphere keeps the hit if it comes after the box in the depth-first walk.
                                bool box_wins = bh.id >= 0;
                                if (box_wins && winner >= 0 && bh.t == best_t)
                                    box_wins = S.set_order[first_box + bh.id] > S.set_order[first_slot + sphere_j];
                                if (box_wins) {
                                    best_t = bh.t;
                                    winner = bh.aux;
                                }
                                PROBE_MARK(PS_SET_BOXES);
                            }
                        }
                        if (mine && winner >= 0) {
                            h.t = best_t;
                            h.node = winner;
                            h.xf = ctx;
                        }
                    }
                }
            }
            continue;
        }
    }
    if constexpr ((F & F_STACK) != 0) {
    int sp = 0;
    if (on) stack[(sp++) * 64 + lane] = (uint32_t)top_node;

    while (ballot(sp > 0) != 0ull) {
#if PROBE_COUNTS
        if (lane == 0) {
            atomicAdd(&g_probe_total[12], 1ull);
            atomicAdd(&g_probe_total[13], (unsigned long long)__popcll(ballot(sp > 0)));
        }
#endif
        if (sp > 0) {
            const uint32_t entry = stack[(--sp) * 64 + lane];
            const int e = (int)(entry & 0xFFFFu);
            const uint32_t tag = entry >> 16;
            if constexpr ((F & F_XF) != 0) {
                if (tag >= END_XF_TRANSLATE) {  // leave instance node e: back to the ray of the level above
                    cur_xf = S.parent_xf[e];
                    cur = ray_in<F>(S, cur_xf, w);
                    if (tag == END_XF) {  // a translation leaves the direction, hence inv and a, as they were
                        inv = d3{1.0f / cur.d.x, 1.0f / cur.d.y, 1.0f / cur.d.z};
                        a = len2(cur.d);
                    }
                    continue;
                }
            }
            const SceneNode n = S.nodes[e];
            const int kind = n.kind;
#if PROBE_COUNTS
            atomicAdd(&g_probe_total[kind < 12 ? kind : 0], 1ull);
#endif
            if (kind == RT_NODE_BVH) {
                if (hit_aabb(n.p, cur, inv, T_MIN, h.t)) {
                    stack[(sp++) * 64 + lane] = (uint32_t)n.b;
                    stack[(sp++) * 64 + lane] = (uint32_t)n.a;
                }
            } else if (kind == RT_NODE_LIST || kind == RT_NODE_BOX) {
                // visit child `tag` next; the list stays on the stack until its last child
                if ((int)tag < n.b) {
                    if ((int)tag + 1 < n.b) stack[(sp++) * 64 + lane] = (uint32_t)e | ((tag + 1u) << 16);
                    stack[(sp++) * 64 + lane] = (uint32_t)S.children[n.a + (int)tag];
                }
            } else if ((F & F_XF) != 0 && (kind == RT_NODE_TRANSLATE || kind == RT_NODE_ROTATE_Y ||
                                           ((F & F_FLIP) != 0 && kind == RT_NODE_FLIP_FACE))) {
                const bool rotates = kind == RT_NODE_ROTATE_Y;
                stack[(sp++) * 64 + lane] = (uint32_t)e | ((rotates ? END_XF : END_XF_TRANSLATE) << 16);
                stack[(sp++) * 64 + lane] = (uint32_t)n.a;
                if (!((F & F_FLIP) != 0 && kind == RT_NODE_FLIP_FACE)) cur = xf_apply(n, cur);
                cur_xf = e;
                if (rotates) {
                    inv = d3{1.0f / cur.d.x, 1.0f / cur.d.y, 1.0f / cur.d.z};
                    a = len2(cur.d);
                }
            } else if ((F & F_RECT) != 0 && kind == DEV_NODE_BOX6) {
                const int side = hit_box6(n.p, cur, T_MIN, h.t);
                if (side >= 0) {
                    h.node = n.c + side;
                    h.xf = cur_xf;
                }
            } else if ((F & F_MEDIUM) != 0 && kind == RT_NODE_MEDIUM) {
                medium_hit<F>(S, n, e, cur, cur_xf, tm, g, h);
            } else {
                double t;
                if (prim_hit<F>(S, n, cur, a, tm, T_MIN, h.t, t)) {
                    h.t = t;
                    h.node = e;
                    h.xf = cur_xf;
                }
            }
        }
    }
    PROBE_MARK(PS_STACK_WALK);
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

// The hit record on its way out of instance node xn, whose child saw the ray `child_ray` (B2/hittable.h:48-49, 127-137;
// a flip_face node turns front_face over and leaves the rest).
template <int F>
__device__ __forceinline__ void leave_instance(Surface& s, const SceneNode& xn, const Ray& child_ray) {
    if ((F & F_FLIP) != 0 && xn.kind == RT_NODE_FLIP_FACE) {
        s.front_face = !s.front_face;
    } else if (xn.kind == RT_NODE_TRANSLATE) {
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

template <int F>
__device__ __forceinline__ Surface make_surface(const SceneDev& S, const Hit& h, Ray w, double tm, bool wants_uv) {
    Surface s;
    s.u = s.v = 0.0;  // moving_sphere and constant_medium leave u,v untouched in the reference
    s.sphere_uv = false;
    s.outward = d3{0, 0, 0};
    const SceneNode n = S.nodes[h.node];
    // Variants without the per-lane stack: h.xf is the index of the world's object the hit lies in, and that object's
    // entry holds the instance chain above it -- kinds and parameters of up to two levels come back together with the
    // node in one round of loads, instead of a walk from parent to parent with a node load per level.
    int xf = h.xf, chain_len = 0;
    SceneNode lv0, lv1;
    lv0.kind = lv1.kind = RT_NODE_TRANSLATE;
    bool by_entry = false;
    if constexpr (TOP_CONTEXT<F>) {
        by_entry = true;
        if (h.xf >= 0) {  // (-1: an object with no instance above it)
            const TopEntry* te = S.tops + h.xf;
            chain_len = te->chain_len;
            lv0.kind = te->chain_kind[0];
            lv1.kind = te->chain_kind[1];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                lv0.p[k] = te->chain_p[0][k];
                lv1.p[k] = te->chain_p[1][k];
            }
            by_entry = chain_len <= 2;
            if (!by_entry) xf = te->chain[chain_len - 1];  // deeper chains: the walk from the innermost instance outwards
        }
    }
    Ray local = w;
    if (by_entry) {
        if (chain_len > 0 && !((F & F_FLIP) != 0 && lv0.kind == RT_NODE_FLIP_FACE)) local = xf_apply(lv0, local);
        if (chain_len > 1 && !((F & F_FLIP) != 0 && lv1.kind == RT_NODE_FLIP_FACE)) local = xf_apply(lv1, local);
    } else {
        local = ray_in<F>(S, xf, w);
    }
    s.p = local.o + h.t * local.d;
    s.mat = n.a;
    switch (n.kind) {
    case RT_NODE_SPHERE:
    case RT_NODE_MOVING_SPHERE: {
        const bool moving = (F & F_MOVING) != 0 && n.kind == RT_NODE_MOVING_SPHERE;
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
        if constexpr (!(F & F_RECT)) break;
        if (wants_uv) {  // (only an image texture reads them)
            // the rectangle's own coordinates of the hit are those of rec.p = r.at(t): the same sum of the same products
            const double ua = comp(s.p, n.kind == RT_NODE_YZ_RECT ? 1 : 0), vb = comp(s.p, n.kind == RT_NODE_XY_RECT ? 1 : 2);
            s.u = (ua - n.p[0]) / (n.p[1] - n.p[0]);
            s.v = (vb - n.p[2]) / (n.p[3] - n.p[2]);
        }
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
    if constexpr ((F & F_XF) != 0) {
        if (by_entry) {
            // the ray inside level 0 once more (level 1's is `local`): a few products, no register held across the switch
            Ray in0 = w;
            if (chain_len > 0 && !((F & F_FLIP) != 0 && lv0.kind == RT_NODE_FLIP_FACE)) in0 = xf_apply(lv0, in0);
            if (chain_len > 1) leave_instance<F>(s, lv1, local);
            if (chain_len > 0) leave_instance<F>(s, lv0, in0);
        } else {
            for (int x = xf, lvl = 0; x >= 0 && lvl < XF_DEPTH_CAP; x = S.parent_xf[x], ++lvl) leave_instance<F>(s, S.nodes[x], ray_in<F>(S, x, w));
        }
    }
    return s;
}

// ---------------------------------------------------------------- textures
// static_cast<int>(floor(x)) as the reference's x86-64 build computes it: beyond the range of int (and for a NaN) the
// conversion instruction yields INT_MIN, where the device's own conversion would saturate.  Perlin noise meets such
// coordinates on very large worlds (its turbulence doubles them six times).
__device__ __forceinline__ int floor_to_int(double x) {
    const double f = floor(x);
    return (f >= -2147483648.0 && f < 2147483648.0) ? (int)f : (int)0x80000000;
}

__device__ __forceinline__ double perlin_noise(const rt_perlin& pn, d3 p) {  // B2/perlin.h:24-48, 86-104
    const double u = p.x - floor(p.x), v = p.y - floor(p.y), w = p.z - floor(p.z);
    const int i = floor_to_int(p.x), j = floor_to_int(p.y), k = floor_to_int(p.z);
    const double uu = u * u * (3 - 2 * u), vv = v * v * (3 - 2 * v), ww = w * w * (3 - 2 * w);
    double accum = 0.0;
#pragma unroll
    for (int di = 0; di < 2; ++di)
#pragma unroll
        for (int dj = 0; dj < 2; ++dj)
#pragma unroll
            for (int dk = 0; dk < 2; ++dk) {
                const int idx = pn.perm_x[((uint32_t)i + (uint32_t)di) & 255u] ^ pn.perm_y[((uint32_t)j + (uint32_t)dj) & 255u] ^ pn.perm_z[((uint32_t)k + (uint32_t)dk) & 255u];
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

// perlin::turb for the lanes with `want` set, computed by the whole wave.  A noise texture is met by one or two lanes
// of a wave at a time, and evaluating it there keeps the other lanes waiting through 7 octaves x 8 lattice corners
// (about 1400 vector instructions and 340 loads).  Here the 56 terms of one lookup are spread over 56 lanes -- lane
// 8 o + c computes corner c (in the loop order of B2/perlin.h:92-101) of octave o, whose point is p 2^o (six exact
// doublings) -- and then added up in the reference's order: the eight terms of an octave one after the other into 0.0,
// the seven octaves one after the other with their weights (B2/perlin.h:55-59).  Same operations on the same values as
// perlin_turb: the result is the same bit for bit.  One lookup after the other; beyond RT_PERLIN_SPREAD_MAX wanting
// lanes every lane evaluates its own.  All 64 lanes must call it.
#ifndef RT_PERLIN_SPREAD_MAX
#define RT_PERLIN_SPREAD_MAX 12
#endif
__device__ __forceinline__ double perlin_turb_for_lanes(const SceneDev& S, bool want, int perlin_index, d3 p, int lane) {
    double result = 0.0;
    const uint64_t wanting = ballot(want);
    if (__popcll(wanting) > RT_PERLIN_SPREAD_MAX) {
        if (want) result = perlin_turb(S.perlins[perlin_index], p);
        return result;
    }
    for (uint64_t m = wanting; m != 0ull; m &= m - 1ull) {
        const int src = __builtin_ctzll(m);
        const rt_perlin& pn = S.perlins[__builtin_amdgcn_readlane(perlin_index, src)];
        const d3 q{lane_value(p.x, src), lane_value(p.y, src), lane_value(p.z, src)};
        const int octave = lane >> 3, di = (lane >> 2) & 1, dj = (lane >> 1) & 1, dk = lane & 1;
        double term = 0.0;
        if (lane < 56) {  // one term of perlin_noise at q 2^octave, B2/perlin.h:24-48, 86-104
            const double scale = (double)(1 << octave);
            const d3 tp{q.x * scale, q.y * scale, q.z * scale};
            const double u = tp.x - floor(tp.x), v = tp.y - floor(tp.y), w = tp.z - floor(tp.z);
            const int i = floor_to_int(tp.x), j = floor_to_int(tp.y), k = floor_to_int(tp.z);
            const double uu = u * u * (3 - 2 * u), vv = v * v * (3 - 2 * v), ww = w * w * (3 - 2 * w);
            const int idx = pn.perm_x[((uint32_t)i + (uint32_t)di) & 255u] ^ pn.perm_y[((uint32_t)j + (uint32_t)dj) & 255u] ^
                            pn.perm_z[((uint32_t)k + (uint32_t)dk) & 255u];
            const d3 c{pn.ranvec[idx][0], pn.ranvec[idx][1], pn.ranvec[idx][2]};
            const d3 weight_v{u - di, v - dj, w - dk};
            term = (di * uu + (1 - di) * (1 - uu)) * (dj * vv + (1 - dj) * (1 - vv)) * (dk * ww + (1 - dk) * (1 - ww)) * dot(c, weight_v);
        }
        double noise = 0.0;  // of this lane's octave: its eight terms in order
#pragma unroll
        for (int c = 0; c < 8; ++c) noise += lane_value_of(term, (lane & ~7) + c);
        double accum = 0.0, weight = 1.0;
#pragma unroll
        for (int o = 0; o < 7; ++o) {
            accum += weight * lane_value(noise, 8 * o);
            weight *= 0.5;
        }
        if (lane == src) result = fabs(accum);
    }
    return result;
}

__device__ __forceinline__ double clampd(double x, double lo, double hi) {  // B2/rtweekend.h:47-51
    if (x < lo) return lo;
    if (x > hi) return hi;
    return x;
}

// texture::value (B2/texture.h).  Checker textures are followed iteratively (at most 4 levels).
// The texture that answers for texture ti at point p: checker textures followed down to the first that is none
// (at most 4 levels: -1 beyond, value() is black there).
template <int F>
__device__ __forceinline__ int texture_resolve(const SceneDev& S, int ti, d3 p) {
    if constexpr ((F & F_CHECKER) != 0) {
        for (int level = 0; level < 4; ++level) {
            const TextureDev t = S.textures[ti];
            if (t.kind != RT_TEX_CHECKER) return ti;
            ti = checker_sines_negative(p) ? t.b : t.a;
        }
        return -1;
    }
    return ti;
}

// texture::value (B2/texture.h) of a texture that is no checker (texture_resolve); turb = perlin::turb at s.p if it is a
// noise texture (perlin_turb_for_lanes).
template <int F>
__device__ __forceinline__ d3 texture_value(const SceneDev& S, int ti, Surface& s, double turb) {
    if (ti < 0) return d3{0, 0, 0};
    {
        const TextureDev t = S.textures[ti];
        if (t.kind == RT_TEX_SOLID) return d3{t.p[0], t.p[1], t.p[2]};
        if ((F & F_NOISE) != 0 && t.kind == RT_TEX_NOISE) {
            const double f = 1 + sin(t.p[0] * s.p.z + 10 * turb);
            return f * (0.5 * d3{1, 1, 1});
        }
        // RT_TEX_IMAGE, B2/texture.h:85-104
        if constexpr (!(F & F_IMAGE)) return d3{0, 1, 1};
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
}

// ---------------------------------------------------------------- "The Rest Of Your Life"
// Restated from the final listing of the public book text (the reference repository holds no book-3 code):
// orthonormal basis, pdf_value / random of the two shapes a light list may hold, cosine density.
struct Onb {
    d3 u, v, w;
};
__device__ __forceinline__ d3 cross(d3 a, d3 b) {
    return d3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ Onb onb_from_w(d3 n) {  // onb::build_from_w
    Onb o;
    o.w = unit_vector(n);
    const d3 a = (fabs(o.w.x) > 0.9) ? d3{0, 1, 0} : d3{1, 0, 0};
    o.v = unit_vector(cross(o.w, a));
    o.u = cross(o.w, o.v);
    return o;
}
__device__ __forceinline__ d3 onb_local(const Onb& o, d3 a) { return a.x * o.u + a.y * o.v + a.z * o.w; }

constexpr double PI_D = 3.14159265358979323846;

// hittable::pdf_value of xz_rect and sphere
__device__ __forceinline__ double shape_pdf_value(const SceneNode& n, d3 origin, d3 v) {
    const Ray r{origin, v};
    double t, ua, vb;
    if (n.kind == RT_NODE_XZ_RECT) {
        if (!hit_rect_raw(1, 0, 2, n.p[0], n.p[1], n.p[2], n.p[3], n.p[4], r, 0.001, __builtin_inf(), t, ua, vb)) return 0.0;
        const d3 outward{0, 1, 0};
        const d3 normal = dot(v, outward) < 0 ? outward : -outward;  // set_face_normal
        const double area = (n.p[1] - n.p[0]) * (n.p[3] - n.p[2]);
        const double distance_squared = t * t * len2(v);
        const double cosine = fabs(dot(v, normal) / sqrt(len2(v)));
        return distance_squared / (cosine * area);
    }
    const d3 center{n.p[0], n.p[1], n.p[2]};
    if (!hit_sphere(center, n.p[3], r, len2(v), 0.001, __builtin_inf(), t)) return 0.0;
    const double cos_theta_max = sqrt(1 - n.p[3] * n.p[3] / len2(center - origin));
    const double solid_angle = 2 * PI_D * (1 - cos_theta_max);
    return 1 / solid_angle;
}
// hittable::random of xz_rect and sphere from the third and fourth word of the bounce's block
__device__ __forceinline__ d3 shape_random(const SceneNode& n, d3 origin, uint32_t w2, uint32_t w3) {
    if (n.kind == RT_NODE_XZ_RECT) {
        const double z = u_range(w3, n.p[2], n.p[3]);
        const double x = u_range(w2, n.p[0], n.p[1]);
        return d3{x, n.p[4], z} - origin;
    }
    const d3 center{n.p[0], n.p[1], n.p[2]};
    const d3 direction = center - origin;
    const double distance_squared = len2(direction);
    const Onb uvw = onb_from_w(direction);
    const double r1 = u01(w2), r2 = u01(w3);  // random_to_sphere
    const double z = 1 + r2 * (sqrt(1 - n.p[3] * n.p[3] / distance_squared) - 1);
    const double phi = 2 * PI_D * r1;
    double sin_phi, cos_phi;  // one argument reduction for both; the device library's sincos returns what its sin and cos
    sincos(phi, &sin_phi, &cos_phi);  // return (compared over 1.3e8 rays of config 5: not one sum differs)
    const double x = cos_phi * sqrt(1 - z * z);
    const double y = sin_phi * sqrt(1 - z * z);
    return onb_local(uvw, d3{x, y, z});
}
// One lambertian bounce of the book's ray_color: direction from the 50/50 mixture of the light density and the
// cosine density; returns  (1 / pdf_val) * (scattering_pdf * attenuation)  as the factor of the throughput.
// ph = Philox block 0 of the site.
__device__ __forceinline__ d3 importance_bounce(const SceneDev& S, d3 p, d3 normal, d3 attenuation, const RngCtx& g, u32x4 ph, d3& dir) {
    // (of the normal's basis only w is read by the densities below; the other two axes are completed further down, for
    // whichever w the lane samples around: onb::build_from_w in two steps)
    Onb uvw;
    uvw.w = unit_vector(normal);
    // Three ways to a direction -- towards a rectangle (a point on it), towards a sphere (random_to_sphere in the basis of
    // the direction to its centre), by the cosine density (random_cosine_direction in the basis of the normal) -- and
    // the last two are the same arithmetic on different inputs: z from one square root, the angle from one sincos, the
    // disc radius from one more square root, onb::local.  The lanes of both take that path together, each on its own
    // inputs, instead of one after the other (the expressions per lane are those of shape_random and of the book's
    // random_cosine_direction, unchanged).
    const bool to_light = u01(ph.x) < 0.5;
    bool to_sphere = false;
    d3 axis = uvw.w;
    const double r1 = u01(ph.z), r2 = u01(ph.w);
    double inner = 1 - r2;
    if (to_light) {
        const int idx = (int)u_range(ph.y, 0, (S.n_lights - 1) + 1);
        const SceneNode n = S.nodes[S.lights[idx < MAX_LIGHT_SHAPES ? idx : 0]];
        if (n.kind == RT_NODE_XZ_RECT) {
            dir = shape_random(n, p, ph.z, ph.w);
        } else {
            to_sphere = true;
            const d3 direction = d3{n.p[0], n.p[1], n.p[2]} - p;
            axis = unit_vector(direction);
            inner = 1 - n.p[3] * n.p[3] / len2(direction);
        }
    }
    if (!to_light || to_sphere) {
        const double q = sqrt(inner);
        const double z = to_sphere ? 1 + r2 * (q - 1) : q;
        const double phi = 2 * PI_D * r1;
        double sin_phi, cos_phi;  // one argument reduction for both; the device library's sincos returns what its sin and cos
        sincos(phi, &sin_phi, &cos_phi);  // return (compared over 1.3e8 rays of config 5: not one sum differs)
        const double s = sqrt(to_sphere ? 1 - z * z : r2);
        Onb basis;
        basis.w = axis;
        const d3 a = (fabs(axis.x) > 0.9) ? d3{0, 1, 0} : d3{1, 0, 0};
        basis.v = unit_vector(cross(axis, a));
        basis.u = cross(axis, basis.v);
        dir = onb_local(basis, d3{cos_phi * s, sin_phi * s, z});
    }
    const d3 ud = unit_vector(dir);
    const double cosine_w = dot(ud, uvw.w);
    const double cos_pdf = (cosine_w <= 0) ? 0 : cosine_w / PI_D;
    const double weight = 1.0 / S.n_lights;
    double sum = 0.0;
    for (int i = 0; i < S.n_lights; ++i) sum += weight * shape_pdf_value(S.nodes[S.lights[i]], p, dir);
    const double pdf_val = 0.5 * sum + 0.5 * cos_pdf;
    const double cosine_n = dot(normal, ud);
    const double spdf = cosine_n < 0 ? 0 : cosine_n / PI_D;
    return (1 / pdf_val) * (spdf * attenuation);
}

}  // namespace

// ---------------------------------------------------------------- render kernel
#ifndef RT_SCENE_MIN_WAVES
#define RT_SCENE_MIN_WAVES 3
#endif
// Philox blocks of the rejection loops computed by the whole wave (philox_blocks_for_lanes)?  Measured on these
// kernels, which wait on dependent loads rather than on vector issue: no gain (final scene -1 %, Cornell box +3 %).
#ifndef RT_PARK_PATH_STATE
#define RT_PARK_PATH_STATE 1
#endif
#ifndef RT_SCENE_WAVE_BLOCKS
#define RT_SCENE_WAVE_BLOCKS 0
#endif
constexpr bool SCENE_WAVE_BLOCKS = RT_SCENE_WAVE_BLOCKS != 0;
template <int F, int MIN_WAVES>
__global__ void __launch_bounds__(256, MIN_WAVES) k_render_scene(const SceneRenderArgs A) {
    // LDS per wave: the traversal stack; without it the candidate lists of the leaf sets; without those nothing (a
    // variant that needs neither is not held back by LDS: 4 workgroups of 40 KB would not fit a CU's 160 KB)
    constexpr int STACK_WORDS = (F & F_STACK) != 0 ? STACK_CAP * 64 : ((F & F_SET) != 0 ? CAND_CAP * 32 + ROW_WORDS * 64 : 64);
    static_assert((STACK_WORDS * 4) % 8 == 0, "a wave's slice of the stack holds whole 8-byte words (64-bit LDS operations of spread_exact_tests)");
    __shared__ __align__(16) uint32_t s_stack[4][STACK_WORDS];
#if RT_PARK_PATH_STATE
    constexpr bool PARK = MIN_WAVES >= 3 && (F & F_SET) != 0 && (F & F_STACK) == 0;  // the variant that spills most and has the LDS to spare
    __shared__ double s_park[4][PARK ? 6 : 1][64];
#endif
    // The camera (21 doubles) is needed once per sample; read from the kernel arguments it would sit in 42 scalar
    // registers for the whole kernel and push hot values out.  One copy per wave in LDS, read back where it is used.
    __shared__ CameraDev s_cam[4];
    const int lane = threadIdx.x & 63;
    uint32_t* stack = s_stack[threadIdx.x >> 6];
    {
        const double* src = (const double*)&A.cam;
        double* dst = (double*)&s_cam[threadIdx.x >> 6];
        if (lane < (int)(sizeof(CameraDev) / sizeof(double))) dst[lane] = src[lane];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    const CameraDev& cam = s_cam[threadIdx.x >> 6];
    const SceneDev& S = A.scene;
    const uint32_t npix = (uint32_t)A.width * (uint32_t)A.height;
    const ItemPlan plan = make_item_plan(A.sample_count, A.item_cap, npix, false);

    bool active = false;
    uint32_t bounce = 0;
    Ray r{d3{0, 0, 0}, d3{0, 0, 1}};
    d3 T{1, 1, 1};
    double tm = 0.0;
    LaneWork w{};
    WaveWork wq{};
    const WorkQueues Q{A.item_counter, A.n_items, A.accum, nullptr, plan.batch, false};  // no accumulation cache: these kernels are short of registers
    uint32_t n_samples = 0, n_rays = 0;  // of the wave (population counts of lane masks: scalar registers)
    uint32_t n_bad = 0;                  // per lane

#ifdef RT_SCENE_PROBE
    ProbeAcc probe{};
    probe.t = clock64();
#endif
    for (;;) {
        if (!refill_work(Q, plan, A.sample_begin, lane, active, w, wq)) break;
        {   // the lanes that start a sample; their camera rays come from Philox blocks the whole wave computes
            const bool start = !active && w.has_item;
            const uint32_t row = div_by_width(w.pixel, A.wdiv.mul, A.wdiv.shift);
            camera_ray_for_lanes<true, SCENE_WAVE_BLOCKS>(start, cam, A.width_m1, A.height_m1, (int)(w.pixel - row * (uint32_t)A.width), A.height - 1 - (int)row,
                                       w.pixel, (uint32_t)w.s_cur, A.key0, A.key1, lane, nullptr, r.o, r.d, tm);
            n_samples += (uint32_t)__popcll(ballot(start));
            if (start) {
                T = d3{1, 1, 1};
                bounce = 0;
                active = true;
            }
        }
        const bool on = active;
        const RngCtx g{w.pixel, (uint32_t)w.s_cur, 1u + bounce, A.key0, A.key1};
        PROBE_MARK(PS_REFILL);
#if RT_PARK_PATH_STATE
        // What the traversal does not touch -- the path's throughput and the item's sums, 12 registers -- waits in LDS
        // meanwhile: the traversal is where registers run out, and what the compiler spills goes to scratch memory.
        if constexpr (PARK) {
            volatile double* pk = s_park[threadIdx.x >> 6][0] + lane;
            pk[0 * 64] = T.x;
            pk[1 * 64] = T.y;
            pk[2 * 64] = T.z;
            pk[3 * 64] = __longlong_as_double((long long)w.fx);
            pk[4 * 64] = __longlong_as_double((long long)w.fy);
            pk[5 * 64] = __longlong_as_double((long long)w.fz);
        }
#endif
        const Hit h = trace<F>(S, r, tm, on, A.exact_only != 0, stack, lane, g PROBE_PASS);
#if RT_PARK_PATH_STATE
        if constexpr (PARK) {
            volatile double* pk = s_park[threadIdx.x >> 6][0] + lane;
            T = d3{pk[0 * 64], pk[1 * 64], pk[2 * 64]};
            w.fx = (int64_t)__double_as_longlong(pk[3 * 64]);
            w.fy = (int64_t)__double_as_longlong(pk[4 * 64]);
            w.fz = (int64_t)__double_as_longlong(pk[5 * 64]);
        }
#endif
#ifdef RT_SCENE_PROBE
        probe.t = clock64();  // trace() accounts for its own sections
#endif

        // the hit record and the material of the lanes that hit something; then, for all lanes together, the first Philox
        // block of the segment's site (every material's first draw: sphere sampler, reflectance, light mixture) and the
        // unit-sphere rejection loop
        const bool hit = on && h.node >= 0;
        constexpr bool importance = (F & F_IS) != 0;  // variants with F_IS only run with RT_RENDER_IMPORTANCE
        Surface s{};
        MaterialDev m{RT_MAT_DIFFUSE_LIGHT, -1, 0, 0, 0, 0};
        if (hit) {
            m = S.node_materials[h.node];
            const bool wants_uv = (F & F_IMAGE) != 0 && m.tex >= 0 && (m.tex & MAT_TEX_WANTS_UV) != 0;
            s = make_surface<F>(S, h, r, tm, wants_uv);
            if (m.tex >= 0) m.tex &= ~MAT_TEX_WANTS_UV;
        }
        PROBE_MARK(PS_SURFACE);
        const bool wants_sphere = hit && ((!importance && m.kind == RT_MAT_LAMBERTIAN) || m.kind == RT_MAT_METAL ||
                                          (!importance && m.kind == RT_MAT_ISOTROPIC));
        const u32x4 block0 = philox4x32_10(g.pixel, g.sample, g.site, 0u, A.key0, A.key1);
        const d3 rs = random_in_unit_sphere_for_lanes<SCENE_WAVE_BLOCKS>(wants_sphere, block0, g.pixel, g.sample, g.site, A.key0, A.key1, lane, nullptr);
        PROBE_MARK(PS_RANDOM);
        // texture::value once, for whichever material wants it (one copy of the Perlin / image / checker code); the
        // turbulence of the lanes that met a noise texture is computed by the whole wave
        const bool wants_texture = hit && m.tex != MAT_TEX_SOLID && (m.kind == RT_MAT_LAMBERTIAN || ((F & F_MEDIUM) != 0 && !importance && m.kind == RT_MAT_ISOTROPIC) ||
                                           (m.kind == RT_MAT_DIFFUSE_LIGHT && (!importance || s.front_face)));
        int tex_index = -1;
        double turb = 0.0;
        if (wants_texture) tex_index = texture_resolve<F>(S, m.tex, s.p);
        if constexpr ((F & F_NOISE) != 0) {
            int perlin_index = 0;
            bool noise = false;
            if (tex_index >= 0) {
                const TextureDev t = S.textures[tex_index];
                noise = t.kind == RT_TEX_NOISE;
                perlin_index = t.a;
            }
            turb = perlin_turb_for_lanes(S, noise, perlin_index, s.p, lane);
        }
        PROBE_MARK(PS_TEXTURE);
        n_rays += (uint32_t)__popcll(ballot(on));
        if (on) {
            bool done = false;
            d3 L_end{0, 0, 0};
            if (!hit) {
                if (S.sky_mode == RT_SKY_GRADIENT) {
                    const d3 unit_direction = unit_vector(r.d);
                    const double t = 0.5 * (unit_direction.y + 1.0);
                    L_end = (1.0 - t) * d3{1.0, 1.0, 1.0} + t * d3{0.5, 0.7, 1.0};
                } else {
                    L_end = d3{S.background[0], S.background[1], S.background[2]};
                }
                done = true;
            } else {
                d3 attenuation{m.ax, m.ay, m.az};
                d3 nd{0, 0, 0};
                d3 tex{m.ax, m.ay, m.az};  // a solid colour; any other texture is looked up
                if (wants_texture) tex = texture_value<F>(S, tex_index, s, turb);
                else if (m.tex != MAT_TEX_SOLID) tex = d3{0, 0, 0};
                if (importance && m.kind == RT_MAT_LAMBERTIAN) {  // book 3: mixture of light and cosine densities
                    attenuation = importance_bounce(S, s.p, s.normal, tex, g, block0, nd);
                } else if (importance && m.kind != RT_MAT_METAL && m.kind != RT_MAT_DIELECTRIC) {
                    // book 3: a light emits from its front face only and ends the path; nothing else scatters
                    if (m.kind == RT_MAT_DIFFUSE_LIGHT && s.front_face) L_end = tex;
                    done = true;
                } else if (m.kind == RT_MAT_LAMBERTIAN) {  // B2/material.h:25-36
                    nd = s.normal + unit_vector(rs);
                    const double eps = 1e-8;
                    if ((fabs(nd.x) < eps) && (fabs(nd.y) < eps) && (fabs(nd.z) < eps)) nd = s.normal;
                    attenuation = tex;
                } else if (m.kind == RT_MAT_METAL) {  // B2/material.h:46-53
                    const d3 reflected = reflect(unit_vector(r.d), s.normal);
                    nd = reflected + m.param * rs;
                    // books 1/2: no scatter below the surface, nothing emitted; book 3 follows the ray regardless
                    if (!importance && !(dot(nd, s.normal) > 0)) done = true;
                } else if (m.kind == RT_MAT_DIELECTRIC) {  // B2/material.h:64-83
                    attenuation = d3{1.0, 1.0, 1.0};
                    const double refraction_ratio = s.front_face ? (1.0 / m.param) : m.param;
                    const d3 unit_direction = unit_vector(r.d);
                    const double cos_theta = fmin(dot(-unit_direction, s.normal), 1.0);
                    const double sin_theta = sqrt(1.0 - cos_theta * cos_theta);
                    bool do_reflect = refraction_ratio * sin_theta > 1.0;
                    if (!do_reflect) do_reflect = reflectance(cos_theta, refraction_ratio) > u01(block0.x);
                    nd = do_reflect ? reflect(unit_direction, s.normal) : refract(unit_direction, s.normal, refraction_ratio);
                } else if ((F & F_MEDIUM) != 0 && m.kind == RT_MAT_ISOTROPIC) {  // B2/material.h:121-127
                    nd = rs;
                    attenuation = tex;
                } else {  // diffuse_light: emits, does not scatter (B2/material.h:102-110)
                    L_end = tex;
                    done = true;
                }
                if (!done) {
                    T = T * attenuation;
                    r.o = s.p;
                    r.d = nd;
                    ++bounce;
                    if ((int)bounce >= A.max_depth) done = true;  // B2/RayTracing.cpp:21-22: the next call (depth 0) returns black
                }
            }
            if (done) {
                add_sample(T * L_end, w.fx, w.fy, w.fz, n_bad);
                active = false;
                ++w.s_cur;
            }
        }
        PROBE_MARK(PS_SHADE);
    }

#ifdef RT_SCENE_PROBE
    if (lane == 0)
        for (int k = 0; k < PS_COUNT; ++k) atomicAdd(&g_probe_cycles[k], probe.cycles[k]);
#endif
    uint64_t v0 = n_samples, v1 = n_rays, v2 = n_bad;  // v0, v1: already the wave's totals
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v2 += __shfl_xor(v2, off);
    if (lane == 0) {
        atomicAdd((unsigned long long*)&A.stats[0], (unsigned long long)v0);
        atomicAdd((unsigned long long*)&A.stats[1], (unsigned long long)v1);
        atomicAdd((unsigned long long*)&A.stats[2], (unsigned long long)v2);
    }
}

// ---------------------------------------------------------------- first-hit kernel
__global__ void __launch_bounds__(256) k_first_hit_scene(const SceneFirstHitArgs A) {
    __shared__ __align__(16) uint32_t s_stack[4][STACK_CAP * 64];
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
#ifdef RT_SCENE_PROBE
    ProbeAcc probe{};
#endif
    const Hit h = trace<F_EVERY>(A.scene, r, C.time0, on, A.exact_only != 0, stack, lane, g PROBE_PASS);
    if (on) A.ids[pixel] = h.node < 0 ? -1 : A.scene.top_of[h.node];
}

// Compiled variants, most specialised first; a world runs the first one that covers its features.
// MIN_WAVES is the occupancy the register allocator is held to (measured best per variant).
#ifndef RT_SCENE_WAVES_A
#define RT_SCENE_WAVES_A 3
#endif
#ifndef RT_SCENE_WAVES_B
#define RT_SCENE_WAVES_B 3
#endif
#ifndef RT_SCENE_WAVES_SET
#define RT_SCENE_WAVES_SET 2
#endif
constexpr int V_A = F_MOVING | F_CHECKER | F_STACK;                    // spheres under a BVH (bouncing spheres)
constexpr int V_B = F_RECT | F_XF | F_MEDIUM | F_STACK;                // boxes, instances, smoke (Cornell boxes)
constexpr int V_C = F_RECT | F_NOISE | F_CHECKER | F_IMAGE | F_STACK;  // textured spheres and lights
constexpr int V_IS = F_RECT | F_XF | F_FLIP | F_IS | F_STACK;          // book 3's Cornell box, light sampled by importance
#ifndef RT_SCENE_WAVES_FLAT
#define RT_SCENE_WAVES_FLAT 3
#endif
#ifndef RT_SCENE_WAVES_B_FLAT
#define RT_SCENE_WAVES_B_FLAT 4
#endif
#ifndef RT_SCENE_WAVES_C_FLAT
#define RT_SCENE_WAVES_C_FLAT 3
#endif
#ifndef RT_SCENE_WAVES_IS_FLAT
#define RT_SCENE_WAVES_IS_FLAT 4
#endif
typedef void (*scene_kernel_t)(const SceneRenderArgs);
struct SceneVariant {
    int features;
    scene_kernel_t kernel;
};
static const SceneVariant g_variants[] = {
    {V_A, k_render_scene<V_A, RT_SCENE_WAVES_A>},
    {V_B & ~F_STACK, k_render_scene<V_B & ~F_STACK, RT_SCENE_WAVES_B_FLAT>},  // ... every object of the world in the wave-level walk
    {V_B, k_render_scene<V_B, RT_SCENE_WAVES_B>},
    {V_C & ~F_STACK, k_render_scene<V_C & ~F_STACK, RT_SCENE_WAVES_C_FLAT>},
    {V_C, k_render_scene<V_C, 2>},
#ifndef RT_SCENE_NO_CHECKER_VARIANT
#define RT_SCENE_NO_CHECKER_VARIANT 1
#endif
#if RT_SCENE_NO_CHECKER_VARIANT
    {(F_ALL & ~F_CHECKER) | F_SET, k_render_scene<(F_ALL & ~F_CHECKER) | F_SET, RT_SCENE_WAVES_FLAT>},  // ... and no checker texture: the Next-Week final scene
#endif
    {F_ALL | F_SET, k_render_scene<F_ALL | F_SET, RT_SCENE_WAVES_FLAT>},  // no object needs the stack
    {F_ALL | F_STACK, k_render_scene<F_ALL | F_STACK, RT_SCENE_MIN_WAVES>},
    {F_ALL | F_SET | F_STACK, k_render_scene<F_ALL | F_SET | F_STACK, RT_SCENE_WAVES_SET>},
    {F_EVERY, k_render_scene<F_EVERY, RT_SCENE_MIN_WAVES>},
    {V_IS & ~F_STACK, k_render_scene<V_IS & ~F_STACK, RT_SCENE_WAVES_IS_FLAT>},
    {V_IS, k_render_scene<V_IS, 2>},
    {F_EVERY | F_IS, k_render_scene<F_EVERY | F_IS, 2>},
};
static const SceneVariant& variant_for(int features, bool importance) {
    features = (features & F_EVERY) | (importance ? F_IS : 0);
    const SceneVariant* last = nullptr;
    for (const SceneVariant& v : g_variants) {
        if (((v.features & F_IS) != 0) != importance) continue;
        last = &v;
        if ((features & ~v.features) == 0) return v;
    }
    return *last;
}

hipError_t launch_render_scene(const SceneRenderArgs& args, int blocks, hipStream_t stream) {
    hipLaunchKernelGGL(variant_for(args.scene.features, args.importance != 0).kernel, dim3((unsigned)blocks), dim3(256), 0, stream, args);
    return hipGetLastError();
}

hipError_t launch_first_hit_scene(const SceneFirstHitArgs& args, hipStream_t stream) {
    const uint32_t npix = (uint32_t)args.width * (uint32_t)args.height;
    hipLaunchKernelGGL(k_first_hit_scene, dim3((npix + 255u) / 256u), dim3(256), 0, stream, args);
    return hipGetLastError();
}

#ifdef RT_SCENE_PROBE
void scene_probe_dump() {
    unsigned long long h[16];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_probe_total), sizeof(h)) != hipSuccess) return;
    static const char* names[] = {"other", "sphere", "moving", "xy", "xz", "yz", "box", "list", "bvh", "translate", "rotate", "medium"};
    unsigned long long c[PS_COUNT];
    if (hipMemcpyFromSymbol(c, HIP_SYMBOL(g_probe_cycles), sizeof(c)) != hipSuccess) return;
    unsigned long long total = 0;
    for (int k = 0; k < PS_COUNT; ++k) total += c[k];
    if (total == 0) return;  // nothing rendered since the last dump
    if (PROBE_COUNTS) std::fprintf(stderr, "scene probe: sphere-set rays with more candidates than slots: %llu\n", h[15]);
    if (PROBE_COUNTS) std::fprintf(stderr, "scene probe: wave-traces %llu; stack loop: trips %llu (%.1f per trace), lane fill %.3f\n", h[14], h[12],
                 (double)h[12] / (double)(h[14] ? h[14] : 1), h[12] ? (double)h[13] / (64.0 * (double)h[12]) : 0.0);
    for (int k = 0; k < 12; ++k)
        if (h[k]) std::fprintf(stderr, "  stack entries %-9s %12llu  (%.2f per lane-trip)\n", names[k], h[k], (double)h[k] / (double)(h[13] ? h[13] : 1));
    unsigned long long z[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_probe_total), z, sizeof(z));
    static const char* sections[] = {"refill + camera", "top-level primitives", "top-level boxes", "top-level media", "leaf sets: spheres",
                                     "leaf sets: boxes", "stack walk", "hit record + material", "first block + sphere", "texture lookup + noise",
                                     "materials + sample end"};
    for (int k = 0; k < PS_COUNT; ++k)
        std::fprintf(stderr, "  wave cycles %-22s %14llu  %5.1f %%\n", sections[k], c[k], 100.0 * (double)c[k] / (double)(total ? total : 1));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_probe_cycles), z, sizeof(c));
    unsigned long long mn[2][6];
    if (hipMemcpyFromSymbol(mn, HIP_SYMBOL(g_probe_mine), sizeof(mn)) != hipSuccess) return;
    for (int k = 0; k < 2; ++k) {
        const unsigned long long visits = mn[k][0] + mn[k][1] + mn[k][2] + mn[k][3] + mn[k][4];
        if (visits)
            std::fprintf(stderr, "  leaf sets of %s: %llu wave visits (%.2f per trace), lanes per visit %.1f; visits by lanes 1-4 / 5-8 / 9-16 / 17-32 / 33-64: %.1f %.1f %.1f %.1f %.1f %%\n",
                         k == 0 ? "spheres" : "boxes", visits, (double)visits / (double)h[14], (double)mn[k][5] / (double)visits,
                         100.0 * mn[k][0] / visits, 100.0 * mn[k][1] / visits, 100.0 * mn[k][2] / visits, 100.0 * mn[k][3] / visits, 100.0 * mn[k][4] / visits);
    }
    unsigned long long zz[2][6] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_probe_mine), zz, sizeof(zz));
}
#endif

int scene_feature_bit_for_node(int kind) {
    switch (kind) {
    case RT_NODE_MOVING_SPHERE: return F_MOVING;
    case RT_NODE_XY_RECT:
    case RT_NODE_XZ_RECT:
    case RT_NODE_YZ_RECT:
    case RT_NODE_BOX: return F_RECT;
    case RT_NODE_TRANSLATE:
    case RT_NODE_ROTATE_Y: return F_XF;
    case RT_NODE_FLIP_FACE: return F_XF | F_FLIP;
    case RT_NODE_MEDIUM: return F_MEDIUM;
    case DEV_NODE_LEAF_SET: return F_SET;
    default: return 0;
    }
}
int scene_feature_bit_for_texture(int kind) {
    return kind == RT_TEX_CHECKER ? F_CHECKER : kind == RT_TEX_NOISE ? F_NOISE : kind == RT_TEX_IMAGE ? F_IMAGE : 0;
}
int scene_feature_bit_for_material(int kind) { return kind == RT_MAT_ISOTROPIC ? F_MEDIUM : 0; }
int scene_feature_bit_for_mixed_set() { return F_STACK; }  // a leaf set of spheres AND boxes: see trace()
int scene_feature_bit_for_top(int end_kind) {  // end_kind: the device kind of a world object below its instance chain
    const bool in_the_wave_walk = (end_kind >= RT_NODE_SPHERE && end_kind <= RT_NODE_YZ_RECT) || end_kind == DEV_NODE_BOX6 ||
                                  end_kind == RT_NODE_MEDIUM || end_kind == DEV_NODE_LEAF_SET;
    return in_the_wave_walk ? 0 : F_STACK;
}

int render_scene_blocks_per_cu(int features, bool importance) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, variant_for(features, importance).kernel, 256, 0) != hipSuccess || n < 1) n = 1;
    return n;
}

}  // namespace rtd
