// C ABI of the library (include/raytracing_amd.h): context, scene upload, render, first hit.
// Host code, plus the one-line kernel that turns fixed-point sums into doubles; the render kernels live in
// rt_kernels_*.hip.
//
// A scene is laid out for one of two kernel families at upload:
//   sphere list   root list of spheres with book-1 materials  -> rt_kernels_spheres.hip
//   scene graph   everything else the reference can build      -> rt_kernels_scene.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "raytracing_amd.h"
#include "rt_internal.h"
#include "rt_render_common.h"

using namespace rtd;

enum ScenePath { PATH_NONE = 0, PATH_SPHERE_LIST = 1, PATH_GRAPH = 2 };

struct rt_context {
    int device = -1;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int num_cus = 0;
    int blocks_per_cu_spheres = 0;
    int blocks_per_cu_spheres_timed = 0;
    int blocks_per_cu_scene = 0;
    bool graph_can_importance = false;  // the uploaded world has light shapes and no isotropic material
    // sphere-list launches with at least this many samples per pixel use the pixel-by-pixel pass (measured
    // crossover on config 2 with 16 equal runs per pixel: 3 % slower at 24 spp, 3 % faster at 32)
    int pixel_pass_min_samples = 32;
    int blocks_override = 0;
    std::string detail;
    bool exact_only = false;     // validation: bypass the float screen
    bool force_graph = false;    // validation: lay every scene out as a scene graph

    ScenePath path = PATH_NONE;
    // A book-2 world of spheres is laid out both ways: as a sphere list (spheres_timed) and as its graph.  The list's
    // screening image bounds every moving sphere over [sweep_lo, sweep_hi]; a camera whose shutter interval leaves
    // that range is rendered from the graph.
    bool spheres_timed = false;
    bool screen_unsafe = false;  // the sphere list holds values the float screen cannot take: double tests only
    double sweep_lo = 0.0, sweep_hi = 0.0;
    int sphere_set_max = 2048;  // above this many spheres the BVH walk of the graph kernels wins (RT_SPHERE_SET_MAX)
    // inside a graph, a BVH subtree of at least / at most this many static spheres and boxes is evaluated as a leaf set
    // (RT_SUBTREE_SET_MIN, RT_SUBTREE_SET_MAX; a minimum above the maximum turns sets off)
    int subtree_set_min = 64, subtree_set_max = 4096;
    int cluster_min = 64;  // smallest sphere set that gets a two-level screen (RT_CLUSTER_MIN)
    SphereListDev spheres{};
    SceneDev graph{};
    std::vector<void*> scene_allocs;  // device memory owned by the current scene

    // per-context device scratch, allocated once
    uint32_t* d_counter = nullptr;  // work-queue head
    uint64_t* d_stats = nullptr;    // 4 counters
    int64_t* d_accum = nullptr;     // fixed-point frame of rt_render (grown on demand)
    size_t accum_elems = 0;
    int32_t* d_ids = nullptr;
    size_t ids_elems = 0;
    // staging of rt_read_sums: the sums as doubles on the device, and a pinned host buffer for the copy
    double* d_conv = nullptr;
    double* h_conv = nullptr;
    size_t conv_elems = 0;
    unsigned char* noalloc_scratch = nullptr;  // this device's copy of `scratch` (rt_render_noalloc)
    // HIP event pairs around the render kernels launched since the last rt_context_take_stats
    std::vector<hipEvent_t> ev_pool;  // [2*i] = start, [2*i+1] = stop
    size_t ev_used = 0;               // pairs in use
};

static constexpr size_t MAX_TIMED_LAUNCHES = 256;


// Scratch of rt_render_noalloc: device memory the kernels need beyond the caller's buffers, held in the code object
// instead of coming from hipMalloc.  [0, 4) work-queue head, [64, 96) the four statistics counters,
// [NOALLOC_HEADER, ...) the fixed-point frame the samples are summed in before they leave as doubles.
// `scratch` is a host-side stand-in for the symbol: host code only ever uses the per-device address that
// hipGetSymbolAddress returned (every GPU holds its own copy of the array at its own address).
static constexpr size_t NOALLOC_HEADER = 256;
static constexpr size_t NOALLOC_SCRATCH_BYTES = NOALLOC_HEADER + (size_t)3 * RT_NOALLOC_MAX_PIXELS * sizeof(int64_t);
static __device__ __align__(256) unsigned char scratch[NOALLOC_SCRATCH_BYTES];
static constexpr int MAX_DEVICES = 64;
static unsigned char* g_scratch_of_device[MAX_DEVICES];  // filled by rt_context_create, once per device
static std::mutex g_scratch_mutex;

static thread_local std::string g_create_detail;

static int fail(rt_context* ctx, int code, const std::string& what) {
    if (ctx) ctx->detail = what;
    else g_create_detail = what;
    return code;
}
// the same without building a std::string: a message that fits the capacity reserved at creation allocates nothing
static int fail_c(rt_context* ctx, int code, const char* what) {
    if (ctx) ctx->detail.assign(what);
    return code;
}
static int hip_fail(rt_context* ctx, hipError_t e, const char* where) {
    return fail(ctx, RT_ERR_HIP, std::string(where) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(ctx, expr)                                        \
    do {                                                          \
        hipError_t e_ = (expr);                                   \
        if (e_ != hipSuccess) return hip_fail((ctx), e_, #expr);  \
    } while (0)

static void free_scene(rt_context* ctx) {
    for (void* p : ctx->scene_allocs) (void)hipFree(p);
    ctx->scene_allocs.clear();
    ctx->path = PATH_NONE;
}

// hipMalloc + hipMemcpy of a host vector; the allocation is owned by the scene.
template <class T>
static int to_device(rt_context* ctx, const std::vector<T>& host, const T** out) {
    void* d = nullptr;
    const size_t bytes = std::max<size_t>(1, host.size()) * sizeof(T);
    HIP_TRY(ctx, hipMalloc(&d, bytes));
    ctx->scene_allocs.push_back(d);
    if (!host.empty()) HIP_TRY(ctx, hipMemcpy(d, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = (const T*)d;
    return RT_OK;
}

// ------------------------------------------------------------------ scene validation
struct GraphCheck {
    const rt_scene_desc* sc;
    std::string error;
    int code = RT_OK;
    std::vector<int32_t> parent_xf, top_of;
    std::vector<uint8_t> visited;

    bool bad(int code_, const std::string& what) {
        if (code == RT_OK) {
            code = code_;
            error = what;
        }
        return false;
    }
    bool material_ok(int m) {
        if (m < 0 || m >= sc->n_materials || !sc->materials) return bad(RT_ERR_INVALID_ARGUMENT, "material index out of range");
        const rt_material& mt = sc->materials[m];
        switch (mt.kind) {
        case RT_MAT_LAMBERTIAN:
        case RT_MAT_DIFFUSE_LIGHT:
        case RT_MAT_ISOTROPIC: return texture_ok(mt.texture, 0);
        case RT_MAT_METAL:
        case RT_MAT_DIELECTRIC: return true;
        default: return bad(RT_ERR_INVALID_ARGUMENT, "unknown material kind " + std::to_string(mt.kind));
        }
    }
    bool texture_ok(int t, int depth) {
        if (t < 0 || t >= sc->n_textures || !sc->textures) return bad(RT_ERR_INVALID_ARGUMENT, "texture index out of range");
        const rt_texture& tx = sc->textures[t];
        switch (tx.kind) {
        case RT_TEX_SOLID: return true;
        case RT_TEX_CHECKER:
            if (depth >= 3) return bad(RT_ERR_LIMIT, "checker textures nested more than 3 deep");
            return texture_ok(tx.a, depth + 1) && texture_ok(tx.b, depth + 1);
        case RT_TEX_NOISE:
            if (tx.a < 0 || tx.a >= sc->n_perlins || !sc->perlins) return bad(RT_ERR_INVALID_ARGUMENT, "perlin index out of range");
            return true;
        case RT_TEX_IMAGE:
            if (tx.a >= sc->n_images || (tx.a >= 0 && !sc->images)) return bad(RT_ERR_INVALID_ARGUMENT, "image index out of range");
            return true;
        default: return bad(RT_ERR_INVALID_ARGUMENT, "unknown texture kind " + std::to_string(tx.kind));
        }
    }
    static bool is_prim(int kind) {
        return kind == RT_NODE_SPHERE || kind == RT_NODE_MOVING_SPHERE || kind == RT_NODE_XY_RECT ||
               kind == RT_NODE_XZ_RECT || kind == RT_NODE_YZ_RECT;
    }
    bool child_range_ok(const rt_node& n) {
        if (n.child0 < 0 || n.child1 < 0 || (int64_t)n.child0 + n.child1 > sc->n_children || (n.child1 > 0 && !sc->children))
            return bad(RT_ERR_INVALID_ARGUMENT, "child range outside children[]");
        return true;
    }
    // A constant_medium boundary: primitives, or a list/box of primitives, under translate / rotate_y levels.
    bool boundary_ok(int idx) {
        int depth = 0;
        while (true) {
            if (idx < 0 || idx >= sc->n_nodes) return bad(RT_ERR_INVALID_ARGUMENT, "node index out of range");
            const rt_node& n = sc->nodes[idx];
            if (n.kind == RT_NODE_TRANSLATE || n.kind == RT_NODE_ROTATE_Y) {
                if (++depth > XF_DEPTH_CAP) return bad(RT_ERR_LIMIT, "medium boundary under too many instance levels");
                idx = n.child0;
                continue;
            }
            if (is_prim(n.kind)) return true;
            if (n.kind == RT_NODE_LIST || n.kind == RT_NODE_BOX) {
                if (!child_range_ok(n)) return false;
                for (int i = 0; i < n.child1; ++i) {
                    const int c = sc->children[n.child0 + i];
                    if (c < 0 || c >= sc->n_nodes) return bad(RT_ERR_INVALID_ARGUMENT, "node index out of range");
                    if (!is_prim(sc->nodes[c].kind))
                        return bad(RT_ERR_UNSUPPORTED_SCENE, "constant_medium boundary: lists may only hold primitives");
                }
                return true;
            }
            return bad(RT_ERR_UNSUPPORTED_SCENE, "constant_medium boundary of node kind " + std::to_string(n.kind));
        }
    }
    // Returns the traversal stack occupancy needed while `idx` (already popped) is processed.
    int walk(int idx, int xf, int xf_depth, int top) {
        if (code != RT_OK) return 0;
        if (idx < 0 || idx >= sc->n_nodes) { bad(RT_ERR_INVALID_ARGUMENT, "node index out of range"); return 0; }
        if (visited[idx]) { bad(RT_ERR_UNSUPPORTED_SCENE, "a node is referenced twice: the graph must be a tree"); return 0; }
        visited[idx] = 1;
        parent_xf[idx] = xf;
        top_of[idx] = top;
        const rt_node& n = sc->nodes[idx];
        switch (n.kind) {
        case RT_NODE_SPHERE:
        case RT_NODE_MOVING_SPHERE:
        case RT_NODE_XY_RECT:
        case RT_NODE_XZ_RECT:
        case RT_NODE_YZ_RECT: material_ok(n.material); return 0;
        case RT_NODE_LIST:
        case RT_NODE_BOX: {
            if (!child_range_ok(n)) return 0;
            if (n.child1 > 65534) { bad(RT_ERR_LIMIT, "more than 65534 objects in one list"); return 0; }
            int need = 0;  // the list keeps one slot (its iterator) while a child is processed
            for (int i = 0; i < n.child1; ++i) {
                const int t = idx == sc->root ? i : top;
                need = std::max(need, 2 + walk(sc->children[n.child0 + i], xf, xf_depth, t));
            }
            return need;
        }
        case RT_NODE_BVH: {
            const int l = walk(n.child0, xf, xf_depth, top);
            const int r = walk(n.child1, xf, xf_depth, top);
            return std::max(2, std::max(1 + l, r));
        }
        case RT_NODE_TRANSLATE:
        case RT_NODE_ROTATE_Y:
        case RT_NODE_FLIP_FACE:  // a wrapper like the instances: the ray passes unchanged, the record's face flips
            if (xf_depth + 1 > XF_DEPTH_CAP) { bad(RT_ERR_LIMIT, "more than 4 nested translate / rotate_y / flip_face levels"); return 0; }
            return std::max(2, 1 + walk(n.child0, idx, xf_depth + 1, top));
        case RT_NODE_MEDIUM: {
            material_ok(n.material);
            boundary_ok(n.child0);
            // the boundary subtree is evaluated in place, never pushed; mark it as used
            std::vector<int> todo{n.child0};
            while (!todo.empty() && code == RT_OK) {
                const int b = todo.back();
                todo.pop_back();
                if (visited[b]) { bad(RT_ERR_UNSUPPORTED_SCENE, "a node is referenced twice: the graph must be a tree"); break; }
                visited[b] = 1;
                parent_xf[b] = xf;
                top_of[b] = top;
                const rt_node& bn = sc->nodes[b];
                if (bn.kind == RT_NODE_TRANSLATE || bn.kind == RT_NODE_ROTATE_Y) todo.push_back(bn.child0);
                if (bn.kind == RT_NODE_LIST || bn.kind == RT_NODE_BOX)
                    for (int i = 0; i < bn.child1; ++i) todo.push_back(sc->children[bn.child0 + i]);
            }
            return 0;
        }
        default: bad(RT_ERR_INVALID_ARGUMENT, "unknown node kind " + std::to_string(n.kind)); return 0;
        }
    }
};

static bool is_book1_sphere_list(const rt_scene_desc* sc) {
    const rt_node& root = sc->nodes[sc->root];
    if (root.child1 > 65535) return false;
    for (int i = 0; i < root.child1; ++i) {
        const rt_node& nd = sc->nodes[sc->children[root.child0 + i]];
        if (nd.kind != RT_NODE_SPHERE) return false;
        const rt_material& m = sc->materials[nd.material];
        if (m.kind == RT_MAT_LAMBERTIAN) {
            if (sc->textures[m.texture].kind != RT_TEX_SOLID) return false;
        } else if (m.kind != RT_MAT_METAL && m.kind != RT_MAT_DIELECTRIC) {
            return false;
        }
    }
    return true;
}

static bool is_reference_box(const rt_scene_desc* sc, const rt_node& n);

// Leaves of a world whose only primitives are spheres and moving spheres, nested in lists and BVH nodes in any way,
// with materials the sphere kernels shade (lambertian over solid / checker textures, metal, dielectric).  A closest
// hit does not depend on the order of the tests, so such a world can be rendered as the list of its leaves --
// provided no BVH node culls a leaf the list would reach: every node's box must enclose the boxes of the leaves below
// it, a moving sphere's taken over its own [time0, time1] as B2/moving_sphere.h:80-89 does.  (bvh_node's constructor,
// B2/bvh.h:50-95, builds exactly such boxes when it is given those times.)
// `prims` receives the node ids in depth-first order, `tops` the index of the top-level object holding each.
struct SphereSetWalk {
    const rt_scene_desc* sc;
    int max_spheres;
    std::vector<int>& prims;
    std::vector<int>& tops;
    bool graph_leaves;  // leaves of a leaf set inside a graph: static spheres of any material (shading goes by node id)
                        // and boxes whose sides are those box::box derives; moving spheres disqualify the subtree
    bool saw_bvh = false;
    std::vector<int> parent_of;  // per leaf: the bvh_node (or list) it hangs under
    int parent = -1;
    struct Box {
        double lo[3], hi[3];
    };
    static Box join(const Box& a, const Box& b) {  // surrounding_box, B2/aabb.h:44-54
        Box r;
        for (int k = 0; k < 3; ++k) {
            r.lo[k] = std::fmin(a.lo[k], b.lo[k]);
            r.hi[k] = std::fmax(a.hi[k], b.hi[k]);
        }
        return r;
    }
    static Box around(const double c[3], double radius) {  // B2/sphere.h:53-58
        Box r;
        for (int k = 0; k < 3; ++k) {
            r.lo[k] = c[k] - radius;
            r.hi[k] = c[k] + radius;
        }
        return r;
    }
    bool texture_fits(int t, int depth) const {
        const rt_texture& tx = sc->textures[t];
        if (tx.kind == RT_TEX_SOLID) return true;
        return tx.kind == RT_TEX_CHECKER && depth < 3 && texture_fits(tx.a, depth + 1) && texture_fits(tx.b, depth + 1);
    }
    bool walk(int idx, int top, Box& box) {
        const rt_node& n = sc->nodes[idx];
        if (n.kind == RT_NODE_SPHERE || n.kind == RT_NODE_MOVING_SPHERE) {
            const rt_material& m = sc->materials[n.material];
            if (graph_leaves) {
            } else if (m.kind == RT_MAT_LAMBERTIAN) {
                if (!texture_fits(m.texture, 0)) return false;
            } else if (m.kind != RT_MAT_METAL && m.kind != RT_MAT_DIELECTRIC) {
                return false;
            }
            if ((int)prims.size() >= max_spheres) return false;
            if (graph_leaves && n.kind != RT_NODE_SPHERE) return false;
            // a negative radius (the hollow-glass trick of book 1) turns sphere::bounding_box inside out: node boxes
            // built from it do not enclose the sphere, and the tree walk culls what a list would hit
            if (!((n.kind == RT_NODE_SPHERE ? n.p[3] : n.p[8]) > 0.0)) return false;
            if (n.kind == RT_NODE_SPHERE) {
                box = around(n.p, n.p[3]);
            } else {  // p = centre0, centre1, time0, time1, radius
                const double time0 = n.p[6], time1 = n.p[7];
                if (!(time0 < time1)) return false;
                double at0[3], at1[3];
                for (int k = 0; k < 3; ++k) {  // moving_sphere::center at both ends, B2/moving_sphere.h:27-29
                    at0[k] = n.p[k] + ((time0 - time0) / (time1 - time0)) * (n.p[3 + k] - n.p[k]);
                    at1[k] = n.p[k] + ((time1 - time0) / (time1 - time0)) * (n.p[3 + k] - n.p[k]);
                }
                box = join(around(at0, n.p[8]), around(at1, n.p[8]));
            }
            prims.push_back(idx);
            tops.push_back(top);
            parent_of.push_back(parent);
            return true;
        }
        if (graph_leaves && n.kind == RT_NODE_BOX) {  // box::bounding_box, B2/box.h:20-23
            if (!is_reference_box(sc, n) || (int)prims.size() >= max_spheres) return false;
            for (int k = 0; k < 3; ++k) {
                box.lo[k] = n.p[k];
                box.hi[k] = n.p[3 + k];
            }
            prims.push_back(idx);
            tops.push_back(top);
            parent_of.push_back(parent);
            return true;
        }
        if (n.kind == RT_NODE_BVH) {
            Box l, r;
            saw_bvh = true;
            const int above = parent;
            parent = idx;
            const bool both = walk(n.child0, top, l) && walk(n.child1, top, r);
            parent = above;
            if (!both) return false;
            box = join(l, r);
            for (int k = 0; k < 3; ++k)
                if (!(n.p[k] <= box.lo[k] && n.p[3 + k] >= box.hi[k])) return false;  // this node could cull a leaf
            for (int k = 0; k < 3; ++k) {
                box.lo[k] = n.p[k];
                box.hi[k] = n.p[3 + k];
            }
            return true;
        }
        if (n.kind == RT_NODE_LIST) {
            bool first = true;
            for (int i = 0; i < n.child1; ++i) {
                Box c;
                parent = idx;
                if (!walk(sc->children[n.child0 + i], idx == sc->root ? i : top, c)) return false;
                box = first ? c : join(box, c);
                first = false;
            }
            if (first) box = Box{{0, 0, 0}, {0, 0, 0}};
            return true;
        }
        return false;
    }
};
// Two leaves of the same shape in the same place.  Between the leaves of a tree the flat form settles a hit at exactly
// the same t in favour of the later leaf, as hittable_list::hit does and as bvh_node::hit does for the two children of
// one node; but a node further up tests its box against the closest hit so far first and rejects it when the ray
// enters the box exactly there (aabb::hit: t_max <= t_min, B2/aabb.h:24-41), so of two coincident leaves in different
// subtrees the reference keeps the EARLIER one.  For rays in general position such ties do not happen; a world that holds
// the same leaf twice makes them happen for every ray that hits it, so such a tree keeps its walk -- unless the two are
// the children of one node (the reference's own trees hold those: bvh_node's constructor makes the last object of an
// odd span both children of its node, B2/bvh.h:65-66) or neighbours in a list, where the later one wins in both forms.
static bool has_coincident_leaves(const rt_scene_desc* sc, const std::vector<int>& prims, const std::vector<int>& parent_of) {
    struct Key {
        std::array<double, 10> k;
        int parent;
        bool operator<(const Key& o) const { return k < o.k; }
    };
    std::vector<Key> keys;
    keys.reserve(prims.size());
    for (size_t i = 0; i < prims.size(); ++i) {
        const rt_node& n = sc->nodes[prims[i]];
        Key key{};
        key.k[0] = (double)n.kind;
        const int used = n.kind == RT_NODE_SPHERE ? 4 : (n.kind == RT_NODE_BOX ? 6 : 9);
        for (int c = 0; c < used; ++c) key.k[1 + c] = n.p[c];
        key.parent = parent_of[i];
        keys.push_back(key);
    }
    std::sort(keys.begin(), keys.end());
    for (size_t i = 0; i + 1 < keys.size(); ++i)
        if (keys[i].k == keys[i + 1].k && keys[i].parent != keys[i + 1].parent) return true;
    return false;
}

static bool collect_sphere_set(const rt_scene_desc* sc, int max_spheres, std::vector<int>& prims, std::vector<int>& tops) {
    SphereSetWalk w{sc, max_spheres, prims, tops, false};
    SphereSetWalk::Box box;
    if (!w.walk(sc->root, -1, box)) return false;
    return !(w.saw_bvh && has_coincident_leaves(sc, prims, w.parent_of));
}

static std::vector<MaterialDev> device_materials(const rt_scene_desc* sc) {
    std::vector<MaterialDev> mats((size_t)sc->n_materials);
    for (int i = 0; i < sc->n_materials; ++i) {
        const rt_material& m = sc->materials[i];
        MaterialDev md{};
        md.kind = m.kind;
        md.tex = m.texture;
        md.param = m.param;
        md.ax = m.albedo[0];
        md.ay = m.albedo[1];
        md.az = m.albedo[2];
        // A solid colour is carried by the material itself: no texture is looked up for it (tex = MAT_TEX_SOLID)
        if ((m.kind == RT_MAT_LAMBERTIAN || m.kind == RT_MAT_DIFFUSE_LIGHT || m.kind == RT_MAT_ISOTROPIC) && m.texture >= 0 &&
            sc->textures[m.texture].kind == RT_TEX_SOLID) {
            md.ax = sc->textures[m.texture].p[0];
            md.ay = sc->textures[m.texture].p[1];
            md.az = sc->textures[m.texture].p[2];
            md.tex = MAT_TEX_SOLID;
        }
        if (md.tex >= 0) {  // does texture::value of this material ever read u, v?  (an image, possibly behind checkers)
            std::vector<int> todo{md.tex};
            bool image = false;
            for (int level = 0; level < 5 && !todo.empty(); ++level) {
                std::vector<int> next;
                for (int t : todo) {
                    if (t < 0 || t >= sc->n_textures) continue;
                    const rt_texture& tx = sc->textures[t];
                    if (tx.kind == RT_TEX_IMAGE) image = true;
                    if (tx.kind == RT_TEX_CHECKER) {
                        next.push_back(tx.a);
                        next.push_back(tx.b);
                    }
                }
                todo.swap(next);
            }
            if (image) md.tex |= MAT_TEX_WANTS_UV;
        }
        // A dielectric has no albedo (its attenuation is white, B1/material.h:59); the slot carries 1.0 / ir, the
        // refraction ratio of a front-face hit (B1/material.h:60), so that no lane divides for it
        // and ay, az Schlick's r0 = ((1 - x) / (1 + x))^2 (B1/material.h:85-86) for the two ratios a ray can meet,
        // x = 1 / ir (front face) and x = ir
        if (m.kind == RT_MAT_DIELECTRIC) {
            md.ax = 1.0 / m.param;
            const double f = (1 - md.ax) / (1 + md.ax), b = (1 - m.param) / (1 + m.param);
            md.ay = f * f;
            md.az = b * b;
        }
        mats[i] = md;
    }
    return mats;
}

// Device image of the spheres `prims` (node ids, in list order) for the screening + exact passes.  With `timed`
// every sphere also gets a motion record, and the screening image holds, for a moving sphere, a sphere that bounds
// its sweep over its own [time0, time1]; sweep_lo / sweep_hi return the interval common to all moving spheres.
// The float screen squares coordinates and radii: beyond about 1e15 (or with non-finite values) its products leave the
// float range and the error bound no longer holds.  Such spheres are tested in double only.
static bool spheres_fit_float_screen(const rt_scene_desc* sc, const std::vector<int>& prims) {
    for (int id : prims) {
        const rt_node& nd = sc->nodes[id];
        const int used = nd.kind == RT_NODE_MOVING_SPHERE ? 6 : 4;
        for (int k = 0; k < used; ++k)
            if (!(std::fabs(nd.p[k]) <= 1e15)) return false;
        if (nd.kind == RT_NODE_MOVING_SPHERE && !(std::fabs(nd.p[8]) <= 1e15)) return false;
    }
    return true;
}

// Splits the objects `rest` (indices into a set) at the median of their centres along the longest axis of the
// centres' bounding box until at most CLUSTER_SIZE remain; left halves are whole multiples of CLUSTER_SIZE, so all but
// one cluster are full.  centre(i, k) = k-th coordinate of object i.  Members of a cluster are in ascending order.
template <class CentreFn>
static std::vector<std::vector<int>> split_into_clusters(std::vector<int> rest, CentreFn centre) {
    std::vector<std::vector<int>> clusters;
    std::vector<std::pair<size_t, size_t>> todo{{0, rest.size()}};  // ranges of `rest` still to split
    while (!todo.empty()) {
        const auto [lo, hi] = todo.back();
        todo.pop_back();
        if (hi - lo <= (size_t)CLUSTER_SIZE) {
            clusters.emplace_back(rest.begin() + (long)lo, rest.begin() + (long)hi);
            std::sort(clusters.back().begin(), clusters.back().end());
            continue;
        }
        double bmin[3] = {INFINITY, INFINITY, INFINITY}, bmax[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (size_t i = lo; i < hi; ++i)
            for (int k = 0; k < 3; ++k) {
                bmin[k] = std::fmin(bmin[k], centre(rest[i], k));
                bmax[k] = std::fmax(bmax[k], centre(rest[i], k));
            }
        int axis = 0;
        for (int k = 1; k < 3; ++k)
            if (bmax[k] - bmin[k] > bmax[axis] - bmin[axis]) axis = k;
        const size_t half = ((hi - lo) / 2 + CLUSTER_SIZE - 1) / CLUSTER_SIZE * CLUSTER_SIZE;
        std::nth_element(rest.begin() + (long)lo, rest.begin() + (long)(lo + half), rest.begin() + (long)hi,
                         [&](int a, int b) { return centre(a, axis) < centre(b, axis); });
        todo.push_back({lo + half, hi});
        todo.push_back({lo, lo + half});
    }
    return clusters;
}

// Two-level form of a set (SphereClustersDev) from its screening spheres `sph` (centre, radius >= 0, in set order) and
// their float images `screen`.  Spheres with more than four times the median radius are "wide"; the others are split
// into clusters by position.  A cluster's box encloses its members' screening
// spheres and is widened by 2^-20 (B + O), B = largest coordinate of any box, O = cover = eight times that: 8 eps
// for the float slab test (rt_screen_ops.h) and as much again for the reference's own rounding -- its discriminant,
// computed in double, can accept a ray that passes a sphere at up to ~3e-8 |o - c| from its surface.
// Leaves C.n_clusters == 0 (no two-level form) when the set is small or its coordinates are too large for the margin.
struct ScreenSphere {
    double c[3], r;
};
static int build_sphere_clusters(rt_context* ctx, const std::vector<ScreenSphere>& sph, const std::vector<SpherePair>& screen,
                                 SphereClustersDev& C) {
    C = SphereClustersDev{};
    const int n = (int)sph.size();
    if (n < ctx->cluster_min || n >= (int)NO_MEMBER) return RT_OK;
    double extent = 0.0;
    std::vector<double> radii((size_t)n);
    for (int i = 0; i < n; ++i) {
        radii[(size_t)i] = sph[(size_t)i].r;
        for (int k = 0; k < 3; ++k) extent = std::fmax(extent, std::fabs(sph[(size_t)i].c[k]) + sph[(size_t)i].r);
    }
    if (!(extent <= 1e6)) return RT_OK;
    std::nth_element(radii.begin(), radii.begin() + n / 2, radii.end());
    const double wide_from = 4.0 * radii[(size_t)n / 2];
    std::vector<int> wide, rest;
    for (int i = 0; i < n; ++i) (sph[(size_t)i].r > wide_from ? wide : rest).push_back(i);
    if (rest.empty()) return RT_OK;

    const std::vector<std::vector<int>> clusters =
        split_into_clusters(rest, [&](int i, int k) { return sph[(size_t)i].c[k]; });

    const int n_clusters = (int)clusters.size();
    if (n_clusters > 1024) return RT_OK;  // a queue item of the walk names a cluster in 10 bits
    const double cover = 8.0 * extent, margin = 0x1p-20 * (extent + cover);
    const int n_groups = (n_clusters + SCREEN_GROUP - 1) / SCREEN_GROUP;
    std::vector<BoxPair> boxes((size_t)n_groups * (SCREEN_GROUP / 2));
    std::memset(boxes.data(), 0, boxes.size() * sizeof(BoxPair));
    SpherePair never;  // a padding slot never yields a candidate: q = -huge
    for (int h = 0; h < 2; ++h) {
        never.cx[h] = never.cy[h] = never.cz[h] = 0.0f;
        never.k[h] = 3.0e38f;
    }
    std::vector<SpherePair> pairs((size_t)n_clusters * (CLUSTER_SIZE / 2), never);
    std::vector<uint16_t> members((size_t)n_clusters * CLUSTER_SIZE, NO_MEMBER);
    auto copy_half = [&](SpherePair& dst, int dst_half, int member) {
        const SpherePair& src = screen[(size_t)member / 2];
        dst.cx[dst_half] = src.cx[member & 1];
        dst.cy[dst_half] = src.cy[member & 1];
        dst.cz[dst_half] = src.cz[member & 1];
        dst.k[dst_half] = src.k[member & 1];
    };
    for (int c = 0; c < n_clusters; ++c) {
        BoxPair& bp = boxes[(size_t)c / 2];
        for (int k = 0; k < 3; ++k) {
            double lo = INFINITY, hi = -INFINITY;
            for (int m : clusters[(size_t)c]) {
                lo = std::fmin(lo, sph[(size_t)m].c[k] - sph[(size_t)m].r);
                hi = std::fmax(hi, sph[(size_t)m].c[k] + sph[(size_t)m].r);
            }
            const double mid = 0.5 * (lo + hi);
            const float mid_f = (float)mid;
            const double hw = (0.5 * (hi - lo) + std::fabs(mid - (double)mid_f) + margin) * (1.0 + 1e-12);
            float hw_f = (float)hw;
            if ((double)hw_f < hw) hw_f = std::nextafterf(hw_f, INFINITY);
            bp.mid[k][c & 1] = mid_f;
            bp.hw[k][c & 1] = hw_f;
        }
        for (size_t j = 0; j < clusters[(size_t)c].size(); ++j) {
            const int m = clusters[(size_t)c][j];
            copy_half(pairs[(size_t)c * (CLUSTER_SIZE / 2) + j / 2], (int)(j & 1), m);
            members[(size_t)c * CLUSTER_SIZE + j] = (uint16_t)m;
        }
    }
    const int n_wide = (int)wide.size();
    const int wide_groups = std::max(1, (n_wide + SCREEN_GROUP - 1) / SCREEN_GROUP);
    std::vector<SpherePair> wide_pairs((size_t)wide_groups * (SCREEN_GROUP / 2), never);
    std::vector<uint16_t> wide_members((size_t)wide_groups * SCREEN_GROUP, NO_MEMBER);
    for (int j = 0; j < n_wide; ++j) {
        copy_half(wide_pairs[(size_t)j / 2], j & 1, wide[(size_t)j]);
        wide_members[(size_t)j] = (uint16_t)wide[(size_t)j];
    }
    int rc;
    if ((rc = to_device(ctx, boxes, &C.boxes)) != RT_OK) return rc;
    if ((rc = to_device(ctx, pairs, &C.pairs)) != RT_OK) return rc;
    if ((rc = to_device(ctx, members, &C.members)) != RT_OK) return rc;
    if ((rc = to_device(ctx, wide_pairs, &C.wide_pairs)) != RT_OK) return rc;
    if ((rc = to_device(ctx, wide_members, &C.wide_members)) != RT_OK) return rc;
    C.n_wide = n_wide;
    C.cover = cover;
    C.n_clusters = n_clusters;
    return RT_OK;
}

static int build_sphere_set(rt_context* ctx, const rt_scene_desc* sc, const std::vector<int>& prims, bool timed, bool two_level,
                            SphereSetDev& S, double& sweep_lo, double& sweep_hi) {
    const int n = (int)prims.size();
    std::vector<ScreenSphere> sph((size_t)n);
    const int n_pairs = (n + 1) / 2;
    const int n_chunks = std::max(1, (n_pairs + 255) / 256);
    std::vector<SpherePair> screen((size_t)n_chunks * 256);  // whole chunks: the sparse walk reads 4 pairs per lane
    std::vector<SphereExact> exact((size_t)std::max(1, n));
    std::vector<SphereMotion> motion(timed ? (size_t)std::max(1, n) : 0);
    for (auto& s : screen) {  // padding never yields a candidate: q = -huge
        for (int h = 0; h < 2; ++h) {
            s.cx[h] = s.cy[h] = s.cz[h] = 0.0f;
            s.k[h] = 3.0e38f;
        }
    }
    sweep_lo = -INFINITY;
    sweep_hi = INFINITY;
    for (int i = 0; i < n; ++i) {
        const rt_node& nd = sc->nodes[prims[(size_t)i]];
        double cx = nd.p[0], cy = nd.p[1], cz = nd.p[2], r = nd.p[3];
        exact[i] = SphereExact{cx, cy, cz, r};
        if (timed) motion[i] = SphereMotion{0.0, 0.0, 0.0, 0, 0.0, 1.0};
        if (nd.kind == RT_NODE_MOVING_SPHERE) {  // p = centre0, centre1, time0, time1, radius
            const double dx = nd.p[3] - cx, dy = nd.p[4] - cy, dz = nd.p[5] - cz;
            exact[i].radius = nd.p[8];
            motion[i] = SphereMotion{dx, dy, dz, (nd.p[6] == 0.0 && nd.p[7] == 1.0) ? 1 : 2, nd.p[6], nd.p[7]};
            sweep_lo = std::max(sweep_lo, nd.p[6]);
            sweep_hi = std::min(sweep_hi, nd.p[7]);
            // every centre(t), t in [time0, time1], lies on the segment centre0..centre1 (up to a few ulps): bound it by
            // the sphere around the midpoint, widened by a relative 1e-6 -- far more than those ulps
            cx += 0.5 * dx;
            cy += 0.5 * dy;
            cz += 0.5 * dz;
            r = (std::fabs(nd.p[8]) + 0.5 * std::sqrt(dx * dx + dy * dy + dz * dz)) * (1.0 + 1e-6) + 1e-300;
        }
        sph[(size_t)i] = ScreenSphere{{cx, cy, cz}, std::fabs(r)};
        const double cc = cx * cx + cy * cy + cz * cz, rr = r * r;
        const double k = cc - rr - 0x1p-18 * (cc + rr);
        // round k toward -inf so the float image never shrinks the margin
        float kf = (float)k;
        if ((double)kf > k) kf = std::nextafterf(kf, -INFINITY);
        SpherePair& sp = screen[i / 2];
        sp.cx[i & 1] = (float)cx;
        sp.cy[i & 1] = (float)cy;
        sp.cz[i & 1] = (float)cz;
        sp.k[i & 1] = kf;
    }
    S = SphereSetDev{};
    int rc;
    if ((rc = to_device(ctx, screen, &S.screen)) != RT_OK) return rc;
    if ((rc = to_device(ctx, exact, &S.exact)) != RT_OK) return rc;
    if (timed && (rc = to_device(ctx, motion, &S.motion)) != RT_OK) return rc;
    S.n = n;
    S.n_chunks = n_chunks;
    if (two_level) return build_sphere_clusters(ctx, sph, screen, S.clusters);
    return RT_OK;
}

// Float image of the boxes `prims` (RT_NODE_BOX node ids) for the slab screen: centre and half-width per axis, the
// half-width widened by the margin derived in rt_box_screen.h for rays whose origin lies within `cover` of the
// coordinate origin; padded to whole screening groups.  cover must be >= 0 and, like the coordinates, <= 8e6.
static int build_box_set(rt_context* ctx, const rt_scene_desc* sc, const std::vector<int>& prims, double cover, BoxSetDev& B) {
    const int n = (int)prims.size();
    const int n_groups = std::max(1, (n + SCREEN_GROUP - 1) / SCREEN_GROUP);
    std::vector<BoxPair> screen((size_t)n_groups * (SCREEN_GROUP / 2));
    std::memset(screen.data(), 0, screen.size() * sizeof(BoxPair));
    double coord_max = 0.0;
    for (int id : prims)
        for (int k = 0; k < 6; ++k) coord_max = std::fmax(coord_max, std::fabs(sc->nodes[id].p[k]));
    const double margin = 0x1p-21 * (coord_max + cover);
    for (int i = 0; i < n; ++i) {
        const rt_node& nd = sc->nodes[prims[(size_t)i]];
        BoxPair& bp = screen[(size_t)i / 2];
        for (int k = 0; k < 3; ++k) {
            // the corners as given are box_min / box_max (B2/box.h:28-30); order them in case they are not
            const double lo = std::fmin(nd.p[k], nd.p[3 + k]), hi = std::fmax(nd.p[k], nd.p[3 + k]);
            const double mid = 0.5 * (lo + hi);
            const float mid_f = (float)mid;
            const double hw = (0.5 * (hi - lo) + std::fabs(mid - (double)mid_f) + margin) * (1.0 + 1e-12);
            float hw_f = (float)hw;
            if ((double)hw_f < hw) hw_f = std::nextafterf(hw_f, INFINITY);
            bp.mid[k][i & 1] = mid_f;
            bp.hw[k][i & 1] = hw_f;
        }
    }
    std::vector<BoxExact> exact((size_t)std::max(1, n));
    for (int i = 0; i < n; ++i) {
        const rt_node& nd = sc->nodes[prims[(size_t)i]];
        BoxExact e{};
        for (int k = 0; k < 6; ++k) e.corner[k] = nd.p[k];
        e.first_side = sc->children[nd.child0];  // a box of the reference's form: its sides are numbered consecutively
        exact[(size_t)i] = e;
    }
    B = BoxSetDev{};
    int rc;
    if ((rc = to_device(ctx, screen, &B.screen)) != RT_OK) return rc;
    if ((rc = to_device(ctx, exact, &B.exact)) != RT_OK) return rc;
    B.n = n;
    B.cover = cover;

    // Two-level form: clusters by position of the box centres; a cluster's box is the union of its members' boxes,
    // widened like theirs (the same bound on the same arithmetic); members keep their own images.
    if (n < ctx->cluster_min || n >= (int)NO_MEMBER) return RT_OK;
    std::vector<int> all((size_t)n);
    for (int i = 0; i < n; ++i) all[(size_t)i] = i;
    const std::vector<std::vector<int>> clusters = split_into_clusters(all, [&](int i, int k) {
        const rt_node& nd = sc->nodes[prims[(size_t)i]];
        return 0.5 * (nd.p[k] + nd.p[3 + k]);
    });
    const int n_clusters = (int)clusters.size();
    if (n_clusters > 1024) return RT_OK;  // a queue item of the walk names a cluster in 10 bits
    const int cluster_groups = (n_clusters + SCREEN_GROUP - 1) / SCREEN_GROUP;
    std::vector<BoxPair> cboxes((size_t)cluster_groups * (SCREEN_GROUP / 2));
    std::memset(cboxes.data(), 0, cboxes.size() * sizeof(BoxPair));
    BoxPair never;  // a padding slot: negative half-widths, never a candidate
    for (int k = 0; k < 3; ++k)
        for (int h = 0; h < 2; ++h) {
            never.mid[k][h] = 0.0f;
            never.hw[k][h] = -1.0f;
        }
    std::vector<BoxPair> pairs((size_t)n_clusters * (CLUSTER_SIZE / 2), never);
    std::vector<uint16_t> members((size_t)n_clusters * CLUSTER_SIZE, NO_MEMBER);
    for (int c = 0; c < n_clusters; ++c) {
        BoxPair& bp = cboxes[(size_t)c / 2];
        for (int k = 0; k < 3; ++k) {
            double lo = INFINITY, hi = -INFINITY;
            for (int m : clusters[(size_t)c]) {
                const rt_node& nd = sc->nodes[prims[(size_t)m]];
                lo = std::fmin(lo, std::fmin(nd.p[k], nd.p[3 + k]));
                hi = std::fmax(hi, std::fmax(nd.p[k], nd.p[3 + k]));
            }
            const double mid = 0.5 * (lo + hi);
            const float mid_f = (float)mid;
            const double hw = (0.5 * (hi - lo) + std::fabs(mid - (double)mid_f) + margin) * (1.0 + 1e-12);
            float hw_f = (float)hw;
            if ((double)hw_f < hw) hw_f = std::nextafterf(hw_f, INFINITY);
            bp.mid[k][c & 1] = mid_f;
            bp.hw[k][c & 1] = hw_f;
        }
        for (size_t j = 0; j < clusters[(size_t)c].size(); ++j) {
            const int m = clusters[(size_t)c][j];
            BoxPair& dst = pairs[(size_t)c * (CLUSTER_SIZE / 2) + j / 2];
            const BoxPair& src = screen[(size_t)m / 2];
            for (int k = 0; k < 3; ++k) {
                dst.mid[k][j & 1] = src.mid[k][m & 1];
                dst.hw[k][j & 1] = src.hw[k][m & 1];
            }
            members[(size_t)c * CLUSTER_SIZE + j] = (uint16_t)m;
        }
    }
    if ((rc = to_device(ctx, cboxes, &B.clusters.boxes)) != RT_OK) return rc;
    if ((rc = to_device(ctx, pairs, &B.clusters.pairs)) != RT_OK) return rc;
    if ((rc = to_device(ctx, members, &B.clusters.members)) != RT_OK) return rc;
    B.clusters.n_clusters = n_clusters;
    return RT_OK;
}

// Lays out a world of spheres for the sphere kernels.  tops == nullptr: a book-1 list; otherwise the book-2 layout
// (motion records, textures, top-level object of every sphere).
static int upload_sphere_list(rt_context* ctx, const rt_scene_desc* sc, const std::vector<int>& prims,
                              const std::vector<int>* tops) {
    const int n = (int)prims.size();
    const bool timed = tops != nullptr;
    std::vector<SphereShade> shade_of((size_t)std::max(1, n));
    for (int i = 0; i < n; ++i) {
        const rt_node& nd = sc->nodes[prims[(size_t)i]];
        const double radius = nd.kind == RT_NODE_MOVING_SPHERE ? nd.p[8] : nd.p[3];
        shade_of[(size_t)i] = SphereShade{1.0 / radius, sc->materials[nd.material].kind, 0};
    }
    std::vector<MaterialDev> by_index = device_materials(sc);
    // the sphere kernels read a solid albedo from the material itself; tex stays only where a checker must be evaluated
    for (int i = 0; i < sc->n_materials; ++i) {
        const rt_material& m = sc->materials[i];
        const bool checker = timed && m.kind == RT_MAT_LAMBERTIAN && m.texture >= 0 && m.texture < sc->n_textures &&
                             sc->textures[m.texture].kind == RT_TEX_CHECKER;
        if (!checker) by_index[(size_t)i].tex = -1;
        else by_index[(size_t)i].tex &= ~MAT_TEX_WANTS_UV;
    }
    std::vector<MaterialDev> mats((size_t)std::max(1, n));  // one per sphere
    for (int i = 0; i < n; ++i) mats[(size_t)i] = by_index[(size_t)sc->nodes[prims[(size_t)i]].material];

    SphereListDev& S = ctx->spheres;
    S = SphereListDev{};
    int rc;
    if ((rc = build_sphere_set(ctx, sc, prims, timed, true, S.set, ctx->sweep_lo, ctx->sweep_hi)) != RT_OK) return rc;
    if ((rc = to_device(ctx, shade_of, &S.shade_of)) != RT_OK) return rc;
    if ((rc = to_device(ctx, mats, &S.materials)) != RT_OK) return rc;
    if (timed) {
        std::vector<int32_t> top_of(tops->begin(), tops->end());
        if ((rc = to_device(ctx, top_of, &S.top_of)) != RT_OK) return rc;
        S.textures = ctx->graph.textures;  // uploaded with the graph of the same world
    }
    S.sky_mode = sc->sky_mode;
    for (int c = 0; c < 3; ++c) S.background[c] = sc->background[c];
    ctx->spheres_timed = timed;
    ctx->screen_unsafe = !spheres_fit_float_screen(sc, prims);
    ctx->path = PATH_SPHERE_LIST;
    return RT_OK;
}

// True if box node n has exactly the six sides box::box (B2/box.h:31-38) derives from its corners, in that order,
// stored as consecutively numbered nodes.  Values are compared bit for bit.
static bool is_reference_box(const rt_scene_desc* sc, const rt_node& n) {
    if (n.child1 != 6 || n.child0 < 0 || n.child0 + 6 > sc->n_children) return false;
    const double x0 = n.p[0], y0 = n.p[1], z0 = n.p[2], x1 = n.p[3], y1 = n.p[4], z1 = n.p[5];
    const int kinds[6] = {RT_NODE_XY_RECT, RT_NODE_XY_RECT, RT_NODE_XZ_RECT, RT_NODE_XZ_RECT, RT_NODE_YZ_RECT, RT_NODE_YZ_RECT};
    const double want[6][5] = {{x0, x1, y0, y1, z1}, {x0, x1, y0, y1, z0}, {x0, x1, z0, z1, y1},
                               {x0, x1, z0, z1, y0}, {y0, y1, z0, z1, x1}, {y0, y1, z0, z1, x0}};
    const int first = sc->children[n.child0];
    for (int i = 0; i < 6; ++i) {
        const int c = sc->children[n.child0 + i];
        if (c != first + i || c < 0 || c >= sc->n_nodes) return false;
        const rt_node& r = sc->nodes[c];
        if (r.kind != kinds[i] || std::memcmp(r.p, want[i], sizeof(want[i])) != 0) return false;
    }
    return true;
}

static int upload_graph(rt_context* ctx, const rt_scene_desc* sc, const GraphCheck& chk) {
    std::vector<SceneNode> nodes((size_t)sc->n_nodes);
    std::vector<double> extra;
    for (int i = 0; i < sc->n_nodes; ++i) {
        const rt_node& n = sc->nodes[i];
        SceneNode d{};
        d.kind = n.kind;
        d.a = d.b = d.c = -1;
        switch (n.kind) {
        case RT_NODE_SPHERE:
            d.a = n.material;
            for (int k = 0; k < 4; ++k) d.p[k] = n.p[k];
            break;
        case RT_NODE_MOVING_SPHERE:
            d.a = n.material;
            for (int k = 0; k < 6; ++k) d.p[k] = n.p[k];
            d.c = (int32_t)extra.size();
            extra.push_back(n.p[6]);
            extra.push_back(n.p[7]);
            extra.push_back(n.p[8]);
            break;
        case RT_NODE_XY_RECT:
        case RT_NODE_XZ_RECT:
        case RT_NODE_YZ_RECT:
            d.a = n.material;
            for (int k = 0; k < 5; ++k) d.p[k] = n.p[k];
            break;
        case RT_NODE_LIST:
        case RT_NODE_BOX:
            d.a = n.child0;
            d.b = n.child1;
            if (n.kind == RT_NODE_BOX && is_reference_box(sc, n)) {
                d.kind = DEV_NODE_BOX6;
                d.c = sc->children[n.child0];
                for (int k = 0; k < 6; ++k) d.p[k] = n.p[k];
            }
            break;
        case RT_NODE_BVH:
            d.a = n.child0;
            d.b = n.child1;
            for (int k = 0; k < 6; ++k) d.p[k] = n.p[k];
            break;
        case RT_NODE_TRANSLATE:
            d.a = n.child0;
            for (int k = 0; k < 3; ++k) d.p[k] = n.p[k];
            break;
        case RT_NODE_ROTATE_Y:
            d.a = n.child0;
            d.p[0] = n.p[0];
            d.p[1] = n.p[1];
            break;
        case RT_NODE_FLIP_FACE:
            d.a = n.child0;
            break;
        case RT_NODE_MEDIUM:
            d.a = n.material;
            d.b = n.child0;
            d.p[0] = n.p[0];
            break;
        default: break;  // unreachable nodes of unknown kind are never visited
        }
        nodes[i] = d;
    }
    // BVH subtrees of static spheres and boxes directly under a top-level object's instance chain become leaf sets
    std::vector<LeafSetDev> sets;
    std::vector<int32_t> set_nodes, set_order;
    // eight times the largest coordinate that appears in the world: room for instance offsets and for a camera
    // outside the objects; rays that start farther out skip the boxes' float screen
    double world_extent = 0.0;
    for (int i = 0; i < sc->n_nodes; ++i) {
        const rt_node& n = sc->nodes[i];
        int used = 0;
        switch (n.kind) {
        case RT_NODE_SPHERE: used = 4; break;
        case RT_NODE_MOVING_SPHERE: used = 6; break;
        case RT_NODE_XY_RECT:
        case RT_NODE_XZ_RECT:
        case RT_NODE_YZ_RECT: used = 5; break;
        case RT_NODE_BOX:
        case RT_NODE_BVH: used = 6; break;
        case RT_NODE_TRANSLATE: used = 3; break;
        default: break;
        }
        for (int k = 0; k < used; ++k)
            if (std::isfinite(n.p[k])) world_extent = std::fmax(world_extent, std::fabs(n.p[k]));
        if (n.kind == RT_NODE_SPHERE) world_extent = std::fmax(world_extent, std::fabs(n.p[0]) + std::fabs(n.p[3]));
    }
    const double cover = std::fmin(8.0 * world_extent, 8e6);
    {
        const rt_node& root = sc->nodes[sc->root];
        for (int i = 0; i < root.child1 && ctx->subtree_set_min <= ctx->subtree_set_max; ++i) {
            int idx = sc->children[root.child0 + i];
            while (sc->nodes[idx].kind == RT_NODE_TRANSLATE || sc->nodes[idx].kind == RT_NODE_ROTATE_Y) idx = sc->nodes[idx].child0;
            if (sc->nodes[idx].kind != RT_NODE_BVH) continue;
            std::vector<int> prims, tops;
            SphereSetWalk w{sc, ctx->subtree_set_max, prims, tops, true};
            SphereSetWalk::Box box;
            if (!w.walk(idx, i, box) || (int)prims.size() < ctx->subtree_set_min) continue;
            if (has_coincident_leaves(sc, prims, w.parent_of)) continue;
            std::vector<int> spheres, boxes, sphere_order, box_order;  // order = position in the tree's depth-first walk
            double coord_max = 0.0;
            for (size_t k = 0; k < prims.size(); ++k) {
                const int id = prims[k];
                const rt_node& leaf = sc->nodes[id];
                (leaf.kind == RT_NODE_BOX ? boxes : spheres).push_back(id);
                (leaf.kind == RT_NODE_BOX ? box_order : sphere_order).push_back((int)k);
                if (leaf.kind == RT_NODE_BOX)
                    for (int k = 0; k < 6; ++k) coord_max = std::fmax(coord_max, std::fabs(leaf.p[k]));
            }
            if (!(coord_max <= 8e6)) continue;  // the slab screen's bound assumes it (also rejects NaN)
            if (!spheres_fit_float_screen(sc, spheres)) continue;
            LeafSetDev sd{};
            double lo, hi;
            int rc = spheres.empty() ? RT_OK : build_sphere_set(ctx, sc, spheres, false, true, sd.spheres, lo, hi);
            if (rc == RT_OK && !boxes.empty()) rc = build_box_set(ctx, sc, boxes, cover, sd.boxes);
            if (rc != RT_OK) return rc;
            {   // the subtree in depth-first order with skip links (ReplayNode)
                std::vector<ReplayNode> replay;
                std::vector<int> member_of((size_t)sc->n_nodes, -1);
                for (size_t k = 0; k < spheres.size(); ++k) member_of[(size_t)spheres[k]] = (int)k;
                for (size_t k = 0; k < boxes.size(); ++k) member_of[(size_t)boxes[k]] = (int)(spheres.size() + k);
                struct Emit {
                    const rt_scene_desc* sc;
                    std::vector<ReplayNode>& out;
                    const std::vector<int>& member_of;
                    void node(int id) {
                        const rt_node& n = sc->nodes[id];
                        ReplayNode r{};
                        if (n.kind == RT_NODE_BVH) {
                            for (int c = 0; c < 6; ++c) r.box[c] = n.p[c];
                            r.member = -1;
                            const size_t at = out.size();
                            out.push_back(r);
                            node(n.child0);
                            node(n.child1);
                            out[at].skip = (int32_t)out.size();
                        } else {
                            r.member = member_of[(size_t)id];
                            r.skip = (int32_t)out.size() + 1;
                            out.push_back(r);
                        }
                    }
                } emit{sc, replay, member_of};
                emit.node(idx);
                rc = to_device(ctx, replay, &sd.replay);
                if (rc != RT_OK) return rc;
                sd.n_replay = (int32_t)replay.size();
            }
            nodes[(size_t)idx].kind = DEV_NODE_LEAF_SET;  // p keeps the subtree's box
            nodes[(size_t)idx].a = (int32_t)sets.size();
            nodes[(size_t)idx].b = (int32_t)set_nodes.size();
            sets.push_back(sd);
            set_nodes.insert(set_nodes.end(), spheres.begin(), spheres.end());
            set_nodes.insert(set_nodes.end(), boxes.begin(), boxes.end());
            set_order.insert(set_order.end(), sphere_order.begin(), sphere_order.end());
            set_order.insert(set_order.end(), box_order.begin(), box_order.end());
        }
    }
    std::vector<TopEntry> tops;
    {
        const rt_node& root = sc->nodes[sc->root];
        for (int i = 0; i < root.child1; ++i) {
            TopEntry te{};
            te.first = sc->children[root.child0 + i];
            int idx = te.first;
            while (te.chain_len < XF_DEPTH_CAP && (sc->nodes[idx].kind == RT_NODE_TRANSLATE || sc->nodes[idx].kind == RT_NODE_ROTATE_Y ||
                                                   sc->nodes[idx].kind == RT_NODE_FLIP_FACE)) {
                te.chain_kind[te.chain_len] = sc->nodes[idx].kind;
                for (int k = 0; k < 3; ++k) te.chain_p[te.chain_len][k] = nodes[(size_t)idx].p[k];
                te.chain[te.chain_len++] = idx;
                idx = sc->nodes[idx].child0;
            }
            te.node = idx;
            te.end = nodes[(size_t)idx];  // after the conversions above: boxes and leaf sets carry their device kinds
            tops.push_back(te);
        }
    }
    std::vector<int32_t> children(sc->children, sc->children + sc->n_children);
    const std::vector<MaterialDev> mats = device_materials(sc);
    std::vector<TextureDev> texs((size_t)sc->n_textures);
    for (int i = 0; i < sc->n_textures; ++i) {
        const rt_texture& t = sc->textures[i];
        TextureDev d{};
        d.kind = t.kind;
        d.a = t.a;
        d.b = t.b;
        for (int k = 0; k < 4; ++k) d.p[k] = t.p[k];
        texs[i] = d;
    }
    std::vector<rt_perlin> perlins(sc->perlins, sc->perlins + sc->n_perlins);
    std::vector<ImageDev> images((size_t)sc->n_images);
    for (int i = 0; i < sc->n_images; ++i) {
        const rt_image& im = sc->images[i];
        images[i] = ImageDev{im.width, im.height, nullptr};
        if (im.rgb && im.width > 0 && im.height > 0) {
            std::vector<uint8_t> bytes(im.rgb, im.rgb + (size_t)3 * im.width * im.height);
            const uint8_t* d = nullptr;
            int rc = to_device(ctx, bytes, &d);
            if (rc != RT_OK) return rc;
            images[i].rgb = d;
        }
    }

    SceneDev& G = ctx->graph;
    G = SceneDev{};
    int rc;
    if ((rc = to_device(ctx, nodes, &G.nodes)) != RT_OK) return rc;
    if ((rc = to_device(ctx, children, &G.children)) != RT_OK) return rc;
    if ((rc = to_device(ctx, chk.parent_xf, &G.parent_xf)) != RT_OK) return rc;
    if ((rc = to_device(ctx, chk.top_of, &G.top_of)) != RT_OK) return rc;
    if ((rc = to_device(ctx, extra, &G.extra)) != RT_OK) return rc;
    if ((rc = to_device(ctx, mats, &G.materials)) != RT_OK) return rc;
    {
        std::vector<MaterialDev> node_mats(nodes.size(), MaterialDev{RT_MAT_DIFFUSE_LIGHT, -1, 0, 0, 0, 0});
        for (size_t i = 0; i < nodes.size(); ++i) {
            const int kind = nodes[i].kind;
            const bool carries = (kind >= RT_NODE_SPHERE && kind <= RT_NODE_YZ_RECT) || kind == RT_NODE_MEDIUM;
            if (carries && nodes[i].a >= 0 && nodes[i].a < sc->n_materials) node_mats[i] = mats[(size_t)nodes[i].a];
        }
        if ((rc = to_device(ctx, node_mats, &G.node_materials)) != RT_OK) return rc;
    }
    if ((rc = to_device(ctx, texs, &G.textures)) != RT_OK) return rc;
    if ((rc = to_device(ctx, perlins, &G.perlins)) != RT_OK) return rc;
    if ((rc = to_device(ctx, images, &G.images)) != RT_OK) return rc;
    if ((rc = to_device(ctx, tops, &G.tops)) != RT_OK) return rc;
    G.n_tops = (int32_t)tops.size();
    if ((rc = to_device(ctx, sets, &G.sets)) != RT_OK) return rc;
    if ((rc = to_device(ctx, set_nodes, &G.set_nodes)) != RT_OK) return rc;
    if ((rc = to_device(ctx, set_order, &G.set_order)) != RT_OK) return rc;
    std::vector<int32_t> lights;
    bool isotropic = false;
    for (int i = 0; i < sc->n_materials; ++i) isotropic = isotropic || sc->materials[i].kind == RT_MAT_ISOTROPIC;
    if (sc->lights_plus1 != 0) {
        const rt_node& L = sc->nodes[sc->lights_plus1 - 1];
        lights.assign(sc->children + L.child0, sc->children + L.child0 + L.child1);
    }
    G.n_lights = (int32_t)lights.size();
    for (int i = 0; i < MAX_LIGHT_SHAPES; ++i) G.lights[i] = i < G.n_lights ? lights[(size_t)i] : -1;
    ctx->graph_can_importance = G.n_lights > 0 && !isotropic;
    G.root = sc->root;
    G.n_nodes = sc->n_nodes;
    G.sky_mode = sc->sky_mode;
    for (int c = 0; c < 3; ++c) G.background[c] = sc->background[c];
    // which kernel variant the world needs (a superset is always valid; RT_SCENE_VARIANT=all forces the general one)
    int features = 0;
    for (int i = 0; i < sc->n_nodes; ++i) features |= scene_feature_bit_for_node(sc->nodes[i].kind);
    if (!sets.empty()) features |= scene_feature_bit_for_node(DEV_NODE_LEAF_SET);
    for (int i = 0; i < sc->n_textures; ++i) features |= scene_feature_bit_for_texture(sc->textures[i].kind);
    for (int i = 0; i < sc->n_materials; ++i) features |= scene_feature_bit_for_material(sc->materials[i].kind);
    for (const TopEntry& te : tops) features |= scene_feature_bit_for_top(te.end.kind);
    for (const LeafSetDev& sd : sets)
        if (sd.spheres.n > 0 && sd.boxes.n > 0) features |= scene_feature_bit_for_mixed_set();
    if (const char* v = std::getenv("RT_SCENE_VARIANT")) {
        if (std::strcmp(v, "all") == 0) features = ~0;
    }
    G.features = features;
    ctx->blocks_per_cu_scene = render_scene_blocks_per_cu(features, false);
    ctx->path = PATH_GRAPH;
    return RT_OK;
}

extern "C" {

int rt_get_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

const char* rt_error_string(int code) {
    switch (code) {
    case RT_OK: return "ok";
    case RT_ERR_INVALID_ARGUMENT: return "invalid argument";
    case RT_ERR_UNSUPPORTED_SCENE: return "scene uses a construct no kernel covers";
    case RT_ERR_NO_DEVICE: return "no usable HIP device";
    case RT_ERR_HIP: return "HIP runtime error";
    case RT_ERR_NO_SCENE: return "no scene uploaded";
    case RT_ERR_OUT_OF_MEMORY: return "out of memory";
    case RT_ERR_LIMIT: return "size limit exceeded";
    case RT_ERR_COMM: return "RCCL error";
    default: return "unknown error code";
    }
}

const char* rt_last_error_detail(const rt_context* ctx) { return ctx ? ctx->detail.c_str() : g_create_detail.c_str(); }

#ifdef RT_TAIL_PROBE
static uint64_t* g_probe = nullptr;
#endif

int rt_context_create(int device_ordinal, void* stream, rt_context** out) {
    if (!out) return fail(nullptr, RT_ERR_INVALID_ARGUMENT, "out is null");
    *out = nullptr;
    int n = rt_get_device_count();
    if (n <= 0 || device_ordinal < 0 || device_ordinal >= n)
        return fail(nullptr, RT_ERR_NO_DEVICE, "device ordinal " + std::to_string(device_ordinal) + " of " + std::to_string(n));
    rt_context* ctx = new (std::nothrow) rt_context();
    if (!ctx) return fail(nullptr, RT_ERR_OUT_OF_MEMORY, "context");
    ctx->device = device_ordinal;
    hipError_t e = hipSetDevice(device_ordinal);
    hipDeviceProp_t prop;
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device_ordinal);
    if (e == hipSuccess) {
        ctx->num_cus = prop.multiProcessorCount;
        if (stream) {
            ctx->stream = (hipStream_t)stream;
        } else {
            e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
            ctx->own_stream = (e == hipSuccess);
        }
    }
    if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_counter, 256);
    if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_stats, 256);
    if (e == hipSuccess) e = hipMemsetAsync(ctx->d_stats, 0, 256, ctx->stream);
    if (e == hipSuccess && device_ordinal < MAX_DEVICES) {
        // the address of this device's copy of the scratch array, looked up once per device with that device current
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        if (!g_scratch_of_device[device_ordinal]) {
            void* ptr = nullptr;
            e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
            if (e == hipSuccess) g_scratch_of_device[device_ordinal] = (unsigned char*)ptr;
        }
        ctx->noalloc_scratch = g_scratch_of_device[device_ordinal];
    }
    ctx->detail.reserve(512);  // later messages that fit do not allocate
    if (e != hipSuccess) {
        int rc = hip_fail(nullptr, e, "rt_context_create");
        rt_context_destroy(ctx);
        return rc;
    }
    ctx->blocks_per_cu_spheres = render_spheres_blocks_per_cu(false);
    ctx->blocks_per_cu_spheres_timed = render_spheres_blocks_per_cu(true);
    if (const char* s = std::getenv("RT_SPHERE_SET_MAX")) ctx->sphere_set_max = std::min(65535, std::max(0, std::atoi(s)));
    if (const char* s = std::getenv("RT_CLUSTER_MIN")) ctx->cluster_min = std::max(1, std::atoi(s));
    if (const char* s = std::getenv("RT_SUBTREE_SET_MIN")) ctx->subtree_set_min = std::max(1, std::atoi(s));
    if (const char* s = std::getenv("RT_SUBTREE_SET_MAX")) ctx->subtree_set_max = std::min(65535, std::max(0, std::atoi(s)));
    if (const char* s = std::getenv("RT_PIXEL_PASS")) ctx->pixel_pass_min_samples = std::atoi(s) != 0 ? 0 : 0x7FFFFFFF;
    if (const char* s = std::getenv("RT_BLOCKS_PER_CU")) {
        int v = std::atoi(s);
        if (v >= 1 && v <= 8) ctx->blocks_override = v;
    }
    *out = ctx;
    return RT_OK;
}

void rt_context_destroy(rt_context* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    free_scene(ctx);
    (void)hipFree(ctx->d_counter);
    (void)hipFree(ctx->d_stats);
    (void)hipFree(ctx->d_accum);
    (void)hipFree(ctx->d_ids);
    (void)hipFree(ctx->d_conv);
    if (ctx->h_conv) (void)hipHostFree(ctx->h_conv);
    for (hipEvent_t ev : ctx->ev_pool) (void)hipEventDestroy(ev);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int rt_scene_upload(rt_context* ctx, const rt_scene_desc* sc) {
    if (!ctx) return RT_ERR_INVALID_ARGUMENT;
    if (!sc || sc->abi_version != RT_ABI_VERSION) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "scene null or abi_version mismatch");
    if (sc->n_nodes <= 0 || !sc->nodes || sc->root < 0 || sc->root >= sc->n_nodes)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "bad node array / root");
    if (sc->n_nodes > 65535) return fail(ctx, RT_ERR_LIMIT, "more than 65535 nodes");
    if (sc->nodes[sc->root].kind != RT_NODE_LIST) return fail(ctx, RT_ERR_UNSUPPORTED_SCENE, "root must be a hittable_list");
    if (sc->sky_mode != RT_SKY_GRADIENT && sc->sky_mode != RT_SKY_BACKGROUND)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "bad sky_mode");
    // every array is walked in full at upload, reachable from the root or not: counts and pointers must agree
    if (sc->n_children < 0 || sc->n_materials < 0 || sc->n_textures < 0 || sc->n_perlins < 0 || sc->n_images < 0)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "negative array count");
    if ((sc->n_children > 0 && !sc->children) || (sc->n_materials > 0 && !sc->materials) || (sc->n_textures > 0 && !sc->textures) ||
        (sc->n_perlins > 0 && !sc->perlins) || (sc->n_images > 0 && !sc->images))
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "an array with a non-zero count is null");
    // perlin::noise indexes ranvec[perm_x ^ perm_y ^ perm_z] (B2/perlin.h:41-43): entries are 0..255
    for (int i = 0; i < sc->n_perlins; ++i)
        for (int k = 0; k < 256; ++k) {
            const rt_perlin& pn = sc->perlins[i];
            if ((pn.perm_x[k] | pn.perm_y[k] | pn.perm_z[k]) & ~255)
                return fail(ctx, RT_ERR_INVALID_ARGUMENT, "perlin permutation entry outside 0..255");
        }
    for (int i = 0; i < sc->n_images; ++i)
        if (sc->images[i].width < 0 || sc->images[i].height < 0 || (int64_t)sc->images[i].width * sc->images[i].height > (int64_t)1 << 28)
            return fail(ctx, RT_ERR_INVALID_ARGUMENT, "bad image size");

    GraphCheck chk;
    chk.sc = sc;
    chk.parent_xf.assign((size_t)sc->n_nodes, -1);
    chk.top_of.assign((size_t)sc->n_nodes, -1);
    chk.visited.assign((size_t)sc->n_nodes, 0);
    const int stack_need = 1 + chk.walk(sc->root, -1, 0, -1);
    if (chk.code != RT_OK) return fail(ctx, chk.code, chk.error);
    if (sc->lights_plus1 != 0) {  // the shapes light is sampled towards (RT_RENDER_IMPORTANCE)
        const int li = sc->lights_plus1 - 1;
        if (li < 0 || li >= sc->n_nodes || sc->nodes[li].kind != RT_NODE_LIST || chk.visited[li])
            return fail(ctx, RT_ERR_INVALID_ARGUMENT, "lights_plus1 must name a list node outside the world graph");
        const rt_node& L = sc->nodes[li];
        if (!chk.child_range_ok(L)) return fail(ctx, chk.code, chk.error);
        if (L.child1 < 1 || L.child1 > MAX_LIGHT_SHAPES) return fail(ctx, RT_ERR_LIMIT, "1 to 16 light shapes");
        for (int i = 0; i < L.child1; ++i) {
            const int c = sc->children[L.child0 + i];
            if (c < 0 || c >= sc->n_nodes || chk.visited[c] ||
                (sc->nodes[c].kind != RT_NODE_XZ_RECT && sc->nodes[c].kind != RT_NODE_SPHERE))
                return fail(ctx, RT_ERR_UNSUPPORTED_SCENE, "light shapes must be xz_rect or sphere nodes of their own");
        }
    }
    if (stack_need > STACK_CAP)
        return fail(ctx, RT_ERR_LIMIT, "traversal needs " + std::to_string(stack_need) + " stack slots, limit " + std::to_string(STACK_CAP));

    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    free_scene(ctx);
    ctx->spheres_timed = false;
    int rc;
    try {  // host containers sized by the caller's counts: no exception leaves the C boundary
        std::vector<int> prims, tops;
        const bool list_kernels = !ctx->force_graph && sc->lights_plus1 == 0;
        if (list_kernels && is_book1_sphere_list(sc)) {
            const rt_node& root = sc->nodes[sc->root];
            prims.assign(sc->children + root.child0, sc->children + root.child0 + root.child1);
            rc = upload_sphere_list(ctx, sc, prims, nullptr);
        } else {
            rc = upload_graph(ctx, sc, chk);
            if (rc == RT_OK && list_kernels && collect_sphere_set(sc, ctx->sphere_set_max, prims, tops))
                rc = upload_sphere_list(ctx, sc, prims, &tops);
        }
    } catch (const std::bad_alloc&) {
        rc = fail(ctx, RT_ERR_OUT_OF_MEMORY, "host memory for the scene layout");
    } catch (const std::exception& e) {
        rc = fail(ctx, RT_ERR_INVALID_ARGUMENT, std::string("scene layout: ") + e.what());
    }
    if (rc != RT_OK) free_scene(ctx);
    return rc;
}

static void fill_camera(CameraDev& C, const rt_camera_desc* cam) {
    for (int c = 0; c < 3; ++c) {
        C.origin[c] = cam->origin[c];
        C.llc[c] = cam->lower_left_corner[c];
        C.horizontal[c] = cam->horizontal[c];
        C.vertical[c] = cam->vertical[c];
        C.u[c] = cam->u[c];
        C.v[c] = cam->v[c];
    }
    C.lens_radius = cam->lens_radius;
    C.time0 = cam->time0;
    C.time1 = cam->time1;
}

// Whether the sphere kernels render this camera: always for a book-1 list; for a book-2 sphere world only while
// the times the camera gives its rays stay inside the interval the moving spheres were bounded over.
static bool use_sphere_kernels(const rt_context* ctx, double t_first, double t_last) {
    if (ctx->path != PATH_SPHERE_LIST) return false;
    if (!ctx->spheres_timed) return true;
    return std::min(t_first, t_last) >= ctx->sweep_lo && std::max(t_first, t_last) <= ctx->sweep_hi;
}

static int check_render_args(rt_context* ctx, const rt_camera_desc* cam, const rt_render_params* p) {
    if (!ctx) return RT_ERR_INVALID_ARGUMENT;
    if (!cam || !p) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "null camera or params");
    if (ctx->path == PATH_NONE) return fail(ctx, RT_ERR_NO_SCENE, "rt_scene_upload has not succeeded on this context");
    if (p->width < 2 || p->height < 2) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "width and height must be >= 2 (the reference divides by W-1, H-1)");
    if ((int64_t)p->width * p->height > (int64_t)1 << 28) return fail(ctx, RT_ERR_LIMIT, "more than 2^28 pixels");
    if (p->sample_begin < 0 || p->sample_count < 0 || (int64_t)p->sample_begin + p->sample_count > (int64_t)1 << 30)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "bad sample range");
    return RT_OK;
}

// What make_item_plan is given for a call: the longest run of samples of one pixel that forms a work item of the
// frame-by-frame plan (eight; RT_ITEM_SAMPLES overrides), or, for the pixel-by-pixel plan, the number of runs a pixel's
// samples are cut into (RT_ITEM_LEVELS overrides).  Sixteen: a power of two, so that four whole pixels fill a wave's 64
// lanes, and the best or within 0.2 % of the best at every count measured on config 2 (62 samples per pixel: 9.20 ms
// against 9.64 with six runs of 8 and one of 14; 63: 9.31 against 9.66; 125: 17.7 against 18.0; 250: 34.4 against 34.9;
// 500: 67.8 against 68.0; 1000, 2000 and the bouncing spheres: level with 32 and 64; 4 runs: +8 %, 64 runs: +0.5 % at
// 500 samples and +25 % at 63).  Sums do not depend on the partition (fixed point).
static int item_cap_for(const rt_render_params* p, bool pixel_pass) {
    (void)p;
    if (!pixel_pass) {
        int cap = 8;
        if (const char* s = std::getenv("RT_ITEM_SAMPLES")) {
            int v = std::atoi(s);
            if (v >= 1) cap = v;
        }
        return cap;
    }
    int runs = 16;
    if (const char* s = std::getenv("RT_ITEM_LEVELS")) {
        int v = std::atoi(s);
        if (v >= 1 && v <= 64) runs = v;
    }
    return runs;
}

// Enqueues the render kernel for a checked call on the context's stream: the samples are ADDED to d_accum_fixed; the
// work-queue head `d_counter` is zeroed first; `d_stats` (4 counters) is added to.  `timed`: bracket the launch with an
// event pair of the context's pool (created on demand), for rt_context_take_stats.
static int enqueue_render(rt_context* ctx, const rt_camera_desc* cam, const rt_render_params* p, int64_t* d_accum_fixed,
                          uint32_t* d_counter, uint64_t* d_stats, bool timed) {

    const uint32_t npix = (uint32_t)p->width * (uint32_t)p->height;
    const bool spheres = use_sphere_kernels(ctx, cam->time0, cam->time1);
    // the pixel-by-pixel pass pays off from 32 samples per pixel (config 2, kernel alone: 24 samples 4.20 against 4.06 ms,
    // 32: 5.16 against 5.32, 48: 7.27 against 7.60, 63: 9.31 against 9.70; RT_PIXEL_PASS=0/1 overrides)
    bool pixel_pass = spheres && p->sample_count >= ctx->pixel_pass_min_samples;
    const int item_cap = item_cap_for(p, pixel_pass);
    const ItemPlan plan = make_item_plan(p->sample_count, item_cap, npix, pixel_pass);
    const int64_t items = (int64_t)item_count(plan, npix);
    // (room above the last item for the one batch every wave still adds to the queue head when it finds it dry)
    if (items > 0xFF000000ll) return fail(ctx, RT_ERR_LIMIT, "more than 2^32 work items; raise RT_ITEM_SAMPLES");
    const bool importance = (p->flags & RT_RENDER_IMPORTANCE) != 0;
    if (importance && (spheres || !ctx->graph_can_importance))
        return fail(ctx, RT_ERR_UNSUPPORTED_SCENE,
                    "RT_RENDER_IMPORTANCE needs a world uploaded with light shapes (lights_plus1) and without isotropic materials");
    int per_cu = spheres ? (ctx->spheres_timed ? ctx->blocks_per_cu_spheres_timed : ctx->blocks_per_cu_spheres)
                         : (importance ? render_scene_blocks_per_cu(ctx->graph.features, true) : ctx->blocks_per_cu_scene);
    if (ctx->blocks_override) per_cu = ctx->blocks_override;
    int blocks = ctx->num_cus * per_cu;
    const int wg_threads = spheres ? RT_SPHERES_WG : 256;
    const int64_t max_useful = (items + wg_threads - 1) / wg_threads;
    if (blocks > max_useful) blocks = (int)max_useful;
    if (blocks < 1) blocks = 1;

    HIP_TRY(ctx, hipMemsetAsync(d_counter, 0, sizeof(uint32_t), ctx->stream));
    // time the kernel with an event pair on its own stream (the first MAX_TIMED_LAUNCHES launches
    // after each rt_context_take_stats)
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (timed && ctx->ev_used < MAX_TIMED_LAUNCHES) {
        if (ctx->ev_pool.size() < 2 * (ctx->ev_used + 1)) {
            hipEvent_t a = nullptr, b = nullptr;
            HIP_TRY(ctx, hipEventCreate(&a));
            ctx->ev_pool.push_back(a);
            HIP_TRY(ctx, hipEventCreate(&b));
            ctx->ev_pool.push_back(b);
        }
        e0 = ctx->ev_pool[2 * ctx->ev_used];
        e1 = ctx->ev_pool[2 * ctx->ev_used + 1];
    }
    if (e0) HIP_TRY(ctx, hipEventRecord(e0, ctx->stream));
    if (spheres) {
        RenderArgs A{};
        A.scene = ctx->spheres;
        fill_camera(A.cam, cam);
        A.width = p->width;
        A.height = p->height;
        A.wdiv = make_width_divider(p->width);
        A.width_m1 = (double)(p->width - 1);
        A.height_m1 = (double)(p->height - 1);
        A.sample_begin = p->sample_begin;
        A.sample_count = p->sample_count;
        A.max_depth = p->max_depth;
        A.item_cap = item_cap;
        A.pixel_pass = pixel_pass ? 1 : 0;
        A.plan = plan;
        A.n_items = (uint32_t)items;
        A.key0 = (uint32_t)p->seed;
        A.key1 = (uint32_t)(p->seed >> 32);
        A.exact_only = (ctx->exact_only || ctx->screen_unsafe || (p->flags & RT_RENDER_EXACT_ONLY)) ? 1 : 0;
        A.accum = d_accum_fixed;
        A.item_counter = d_counter;
        A.stats = d_stats;
#ifdef RT_TAIL_PROBE
        {   // debug build: per-wave timestamps of the LAST launch, dumped by rt_context_take_stats
            static uint64_t* d_probe = nullptr;
            if (!d_probe) HIP_TRY(ctx, hipMalloc((void**)&d_probe, 16 * sizeof(uint64_t) * 65536));
            HIP_TRY(ctx, hipMemsetAsync(d_probe, 0, 16 * sizeof(uint64_t) * 65536, ctx->stream));
            A.probe = blocks * 4 <= 65536 ? d_probe : nullptr;
            g_probe = d_probe;
        }
#endif
        HIP_TRY(ctx, launch_render_spheres(A, ctx->spheres_timed, blocks, ctx->stream));
    } else {
        SceneRenderArgs A{};
        A.scene = ctx->graph;
        fill_camera(A.cam, cam);
        A.width = p->width;
        A.height = p->height;
        A.wdiv = make_width_divider(p->width);
        A.width_m1 = (double)(p->width - 1);
        A.height_m1 = (double)(p->height - 1);
        A.sample_begin = p->sample_begin;
        A.sample_count = p->sample_count;
        A.max_depth = p->max_depth;
        A.item_cap = item_cap;
        A.n_items = (uint32_t)items;
        A.key0 = (uint32_t)p->seed;
        A.key1 = (uint32_t)(p->seed >> 32);
        A.importance = importance ? 1 : 0;
        A.exact_only = (ctx->exact_only || (p->flags & RT_RENDER_EXACT_ONLY)) ? 1 : 0;
        A.accum = d_accum_fixed;
        A.item_counter = d_counter;
        A.stats = d_stats;
        HIP_TRY(ctx, launch_render_scene(A, blocks, ctx->stream));
    }
    if (e1) {
        HIP_TRY(ctx, hipEventRecord(e1, ctx->stream));
        ++ctx->ev_used;
    }
    return RT_OK;
}

// What enqueue_render refuses, without enqueuing anything.
static int check_launch(rt_context* ctx, const rt_camera_desc* cam, const rt_render_params* p) {
    if (p->sample_count == 0 || p->max_depth <= 0) return RT_OK;  // nothing is launched
    const uint32_t npix = (uint32_t)p->width * (uint32_t)p->height;
    const bool spheres = use_sphere_kernels(ctx, cam->time0, cam->time1);
    const bool pixel_pass = spheres && p->sample_count >= ctx->pixel_pass_min_samples;
    const ItemPlan plan = make_item_plan(p->sample_count, item_cap_for(p, pixel_pass), npix, pixel_pass);
    if ((int64_t)item_count(plan, npix) > 0xFF000000ll) return fail(ctx, RT_ERR_LIMIT, "more than 2^32 work items; raise RT_ITEM_SAMPLES");
    if ((p->flags & RT_RENDER_IMPORTANCE) != 0 && (spheres || !ctx->graph_can_importance))
        return fail(ctx, RT_ERR_UNSUPPORTED_SCENE,
                    "RT_RENDER_IMPORTANCE needs a world uploaded with light shapes (lights_plus1) and without isotropic materials");
    return RT_OK;
}

int rt_render_check(rt_context* ctx, const rt_camera_desc* cam, const rt_render_params* p) {
    const int rc = check_render_args(ctx, cam, p);
    return rc != RT_OK ? rc : check_launch(ctx, cam, p);
}

int rt_render_device(rt_context* ctx, const rt_camera_desc* cam, const rt_render_params* p, int64_t* d_accum_fixed) {
    int rc = check_render_args(ctx, cam, p);
    if (rc != RT_OK) return rc;
    if (!d_accum_fixed) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "null accumulation buffer");
    if (p->sample_count == 0 || p->max_depth <= 0) return RT_OK;  // depth <= 0: every sample is black
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return enqueue_render(ctx, cam, p, d_accum_fixed, ctx->d_counter, ctx->d_stats, true);
}

static void launch_sums_to_double(rt_context* ctx, const int64_t* d_fixed, double* d_out, size_t elems);

int rt_render_noalloc(rt_context* ctx, const rt_camera_desc* cam, const rt_render_params* p, double* d_accum_rgb_out) {
    int rc = check_render_args(ctx, cam, p);
    if (rc != RT_OK) return rc;
    if (!d_accum_rgb_out) return fail_c(ctx, RT_ERR_INVALID_ARGUMENT, "null output buffer");
    if (!ctx->noalloc_scratch) return fail_c(ctx, RT_ERR_NO_DEVICE, "this device has no scratch array (device ordinal beyond 63)");
    const int64_t npix = (int64_t)p->width * p->height;
    if (npix > (int64_t)RT_NOALLOC_MAX_PIXELS)
        return fail_c(ctx, RT_ERR_LIMIT, "rt_render_noalloc: width*height exceeds RT_NOALLOC_MAX_PIXELS (the scratch frame is fixed-size); use rt_render_device");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // Every call on this GPU shares the array and finds the last call's contents in it: the queue head, the counters
    // and the part of the frame this call sums in all start from zero.
    unsigned char* base = ctx->noalloc_scratch;
    const size_t elems = (size_t)3 * (size_t)npix;
    HIP_TRY(ctx, hipMemsetAsync(base, 0, NOALLOC_HEADER + elems * sizeof(int64_t), ctx->stream));
    int64_t* frame = (int64_t*)(base + NOALLOC_HEADER);
    if (p->sample_count > 0 && p->max_depth > 0) {
        rc = enqueue_render(ctx, cam, p, frame, (uint32_t*)base, (uint64_t*)(base + 64), false);
        if (rc != RT_OK) return rc;
    }
    launch_sums_to_double(ctx, frame, d_accum_rgb_out, elems);
    HIP_TRY(ctx, hipGetLastError());
    return RT_OK;
}

int rt_context_synchronize(rt_context* ctx) {
    if (!ctx) return RT_ERR_INVALID_ARGUMENT;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

// fixed-point sums -> doubles: the int64 -> double conversion rounds to nearest even like the host's, and the scale
// is a power of two, so the values are those a host loop over the copied integers would give
__global__ void __launch_bounds__(256) k_sums_to_double(const int64_t* __restrict__ fixed, double* __restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = (double)fixed[i] * (1.0 / (double)(1ull << RT_ACCUM_FRAC_BITS));
}

static void launch_sums_to_double(rt_context* ctx, const int64_t* d_fixed, double* d_out, size_t elems) {
    const unsigned blocks = (unsigned)std::max<size_t>(1, std::min<size_t>((elems + 255) / 256, (size_t)ctx->num_cus * 8));
    hipLaunchKernelGGL(k_sums_to_double, dim3(blocks), dim3(256), 0, ctx->stream, d_fixed, d_out, elems);
}

// Staging of rt_read_sums for frames of `elems` values.  The buffers are never in flight when a call returns
// (rt_read_sums ends with a stream wait), so growing them needs no synchronisation -- rt_render grows them while its
// kernel is running.
static int ensure_staging(rt_context* ctx, size_t elems) {
    if (ctx->conv_elems >= elems) return RT_OK;
    (void)hipFree(ctx->d_conv);
    if (ctx->h_conv) (void)hipHostFree(ctx->h_conv);
    ctx->d_conv = nullptr;
    ctx->h_conv = nullptr;
    ctx->conv_elems = 0;
    HIP_TRY(ctx, hipMalloc((void**)&ctx->d_conv, elems * sizeof(double)));
    HIP_TRY(ctx, hipHostMalloc((void**)&ctx->h_conv, elems * sizeof(double), hipHostMallocDefault));
    ctx->conv_elems = elems;
    return RT_OK;
}

int rt_read_sums(rt_context* ctx, const int64_t* d_accum_fixed, int width, int height, double* accum_rgb_out) {
    if (!ctx) return RT_ERR_INVALID_ARGUMENT;
    if (!d_accum_fixed || !accum_rgb_out) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "null buffer");
    if (width < 1 || height < 1 || (int64_t)width * height > (int64_t)1 << 28) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "bad image size");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t elems = (size_t)3 * width * height;
    const int rc = ensure_staging(ctx, elems);
    if (rc != RT_OK) return rc;
    launch_sums_to_double(ctx, d_accum_fixed, ctx->d_conv, elems);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_conv, ctx->d_conv, elems * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    std::memcpy(accum_rgb_out, ctx->h_conv, elems * sizeof(double));
    return RT_OK;
}

int rt_context_set_exact_only(rt_context* ctx, int on) {
    if (!ctx) return RT_ERR_INVALID_ARGUMENT;
    ctx->exact_only = on != 0;
    return RT_OK;
}

int rt_context_set_force_graph(rt_context* ctx, int on) {
    if (!ctx) return RT_ERR_INVALID_ARGUMENT;
    ctx->force_graph = on != 0;
    return RT_OK;
}

int rt_context_kernels_for(const rt_context* ctx, const rt_camera_desc* cam) {
    if (!ctx || !cam || ctx->path == PATH_NONE) return RT_KERNELS_NONE;
    return use_sphere_kernels(ctx, cam->time0, cam->time1) ? RT_KERNELS_SPHERE_LIST : RT_KERNELS_SCENE_GRAPH;
}

int rt_context_take_stats(rt_context* ctx, rt_stats* out) {
    if (!ctx || !out) return RT_ERR_INVALID_ARGUMENT;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
#ifdef RT_SCENE_PROBE
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    scene_probe_dump();
#endif
#ifdef RT_TAIL_PROBE
    if (g_probe) {
        if (const char* path = std::getenv("RT_TAIL_PROBE_FILE")) {
            std::vector<uint64_t> h(16 * 65536);
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            HIP_TRY(ctx, hipMemcpy(h.data(), g_probe, h.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
            if (FILE* f = std::fopen(path, "wb")) {
                std::fwrite(h.data(), sizeof(uint64_t), h.size(), f);
                std::fclose(f);
            }
        }
    }
#endif
    uint64_t h[4] = {0, 0, 0, 0};
    HIP_TRY(ctx, hipMemcpyAsync(h, ctx->d_stats, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_stats, 0, sizeof(h), ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    out->samples = h[0];
    out->rays = h[1];
    out->nan_samples = h[2];
    out->overflow_rays = h[3];
    double total = 0.0;
    for (size_t i = 0; i < ctx->ev_used; ++i) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, ctx->ev_pool[2 * i], ctx->ev_pool[2 * i + 1]) != hipSuccess) {
            (void)hipGetLastError();
            ms = 0.0f;
        }
        total += ms;
    }
    out->kernel_ms = total;
    out->kernel_launches = (uint64_t)ctx->ev_used;
    ctx->ev_used = 0;
    return RT_OK;
}

int rt_render(rt_context* ctx, const rt_camera_desc* cam, const rt_render_params* p, double* accum_rgb_out,
              rt_stats* stats) {
    int rc = check_render_args(ctx, cam, p);
    if (rc != RT_OK) return rc;
    if (!accum_rgb_out) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "null output");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t elems = (size_t)3 * p->width * p->height;
    if (ctx->accum_elems < elems) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->d_accum);
        ctx->d_accum = nullptr;
        ctx->accum_elems = 0;
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_accum, elems * sizeof(int64_t)));
        ctx->accum_elems = elems;
    }
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_accum, 0, elems * sizeof(int64_t), ctx->stream));
    rc = rt_render_device(ctx, cam, p, ctx->d_accum);
    if (rc != RT_OK) return rc;
    if ((rc = ensure_staging(ctx, elems)) != RT_OK) return rc;  // a first call allocates while the kernel runs
    rc = rt_read_sums(ctx, ctx->d_accum, p->width, p->height, accum_rgb_out);
    if (rc != RT_OK) return rc;
    if (stats) {
        rc = rt_context_take_stats(ctx, stats);
        if (rc != RT_OK) return rc;
    }
    return RT_OK;
}

int rt_first_hit(rt_context* ctx, const rt_camera_desc* cam, int width, int height, int32_t* ids_out) {
    if (!ctx) return RT_ERR_INVALID_ARGUMENT;
    if (!cam || !ids_out) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "null camera or output");
    if (ctx->path == PATH_NONE) return fail(ctx, RT_ERR_NO_SCENE, "rt_scene_upload has not succeeded on this context");
    if (width < 2 || height < 2 || (int64_t)width * height > (int64_t)1 << 28)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "bad image size");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t elems = (size_t)width * height;
    if (ctx->ids_elems < elems) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->d_ids);
        ctx->d_ids = nullptr;
        ctx->ids_elems = 0;
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_ids, elems * sizeof(int32_t)));
        ctx->ids_elems = elems;
    }
    if (use_sphere_kernels(ctx, cam->time0, cam->time0)) {
        FirstHitArgs A{};
        A.scene = ctx->spheres;
        fill_camera(A.cam, cam);
        A.width = width;
        A.height = height;
        A.exact_only = (ctx->exact_only || ctx->screen_unsafe) ? 1 : 0;
        A.ids = ctx->d_ids;
        HIP_TRY(ctx, launch_first_hit_spheres(A, ctx->spheres_timed, ctx->stream));
    } else {
        SceneFirstHitArgs A{};
        A.scene = ctx->graph;
        fill_camera(A.cam, cam);
        A.width = width;
        A.height = height;
        A.exact_only = ctx->exact_only ? 1 : 0;
        A.ids = ctx->d_ids;
        HIP_TRY(ctx, launch_first_hit_scene(A, ctx->stream));
    }
    HIP_TRY(ctx, hipMemcpyAsync(ids_out, ctx->d_ids, elems * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

}  // extern "C"
