// Internal interface between the C-ABI host layer (rt_api.hip) and the kernel files.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "raytracing_amd.h"

namespace rtd {

// ---- device layout of a flat list of spheres (book-1 worlds: B1/hittable_list.h + B1/sphere.h)
//
// Two images of every sphere live in HBM (a 487-sphere world is 7.8 KB + 15.6 KB, so both
// stay in the scalar / vector caches for the whole render):
//  - SpherePair, 32 B, float: two consecutive spheres interleaved component by component, read
//    with wave-uniform scalar loads by the screening pass.  Each member is one 64-bit SGPR pair,
//    i.e. one operand of a packed FP32 instruction that works on both spheres at once;
//  - SphereExact, 32 B, double: gathered per lane by the exact pass for the few candidates.
struct SpherePair {
    float cx[2], cy[2], cz[2];
    float k[2];  // |c|^2 - r^2 - margin(c, r), rounded down from double; see screen pass
};
struct SphereExact {
    double cx, cy, cz, radius;
};
// Worlds of book 2 that hold nothing but spheres (static or moving, under any nesting of BVH nodes) run on the same
// kernels: every sphere also has a SphereMotion record, and the screening image holds, for a moving sphere, the
// sphere that bounds its sweep over [time0, time1].
struct TextureDev;
struct SphereMotion {
    double vx, vy, vz;     // center1 - center0 (B2/moving_sphere.h:28; the SphereExact centre is center0), from the upload
    int64_t moving;        // 0: static sphere (the other members are unused), 1: moving over [0, 1], 2: over [time0, time1]
    double time0, time1;   // read only when moving == 2
};
struct MaterialDev {
    int32_t kind;  // RT_MAT_*
    int32_t tex;   // texture index of lambertian / diffuse_light / isotropic (general scenes); MAT_TEX_SOLID: the colour is in ax, ay, az; else -1
    double ax, ay, az;  // albedo (a solid texture's colour, metal); dielectric: ax = 1.0 / ir, ay and az = Schlick's r0 for the ratios 1 / ir and ir
    double param;       // fuzz / ir
};

constexpr int32_t MAT_TEX_SOLID = -2;
constexpr int32_t MAT_TEX_WANTS_UV = 1 << 30;  // set in a texture index whose texture (or a checker's branch) is an image: the only reader of u, v

constexpr int SCREEN_GROUP = 32;  // spheres per screening mask word (16 pairs)
constexpr int CAND_CAP = 32;      // candidate slots per lane held in LDS
#ifndef RT_CLUSTER_SIZE
#define RT_CLUSTER_SIZE 16
#endif
constexpr int CLUSTER_SIZE = RT_CLUSTER_SIZE;  // spheres per cluster of the two-level screen (even, at most 32)
constexpr uint16_t NO_MEMBER = 0xFFFF;         // padding slot of a cluster

// Screening image of axis-aligned boxes: two boxes interleaved component by component, like SpherePair.
struct BoxPair {
    float mid[3][2];  // centre per axis
    float hw[3][2];   // half-width per axis, widened by the screen's error margin (rt_screen_ops.h), rounded up
};
// Two-level form of a large set.  The spheres are partitioned by position into clusters of at most CLUSTER_SIZE; the
// wave screens the clusters' bounding boxes (slab test, scalar operands), then every lane screens only the members of
// the clusters its own ray meets, reading their pair images with vector loads.  Spheres much larger than the others
// ("wide": they would bloat any cluster) stay in a one-level image of their own.
struct SphereClustersDev {
    int32_t n_clusters;  // 0: the set has no two-level form
    int32_t n_wide;
    double cover;        // the box margins hold for rays whose origin coordinates are at most this in magnitude (held
                         // as a double: it is compared with doubles, and a converted float would sit in a vector register)
    const BoxPair* boxes;       // per cluster, padded to whole groups of SCREEN_GROUP
    const SpherePair* pairs;    // CLUSTER_SIZE / 2 per cluster; padding slots never yield a candidate
    const uint16_t* members;    // CLUSTER_SIZE per cluster: index in the set, NO_MEMBER for padding
    const SpherePair* wide_pairs;   // padded to whole groups of SCREEN_GROUP / 2 pairs
    const uint16_t* wide_members;   // n_wide indices in the set
};

struct SphereSetDev {
    int32_t n;         // spheres in list order (= object ids)
    int32_t n_chunks;       // 256-pair chunks of the screen image (padded with never-hit pairs to a whole chunk)
    const SpherePair* screen;
    const SphereExact* exact;
    const SphereMotion* motion;  // book-2 sphere worlds only, else null
    SphereClustersDev clusters;
};
// Two-level form of a large box set, as for spheres: spatial clusters of at most CLUSTER_SIZE boxes, the clusters'
// bounding boxes for the wave-wide first level, per cluster the images and indices of its members.
struct BoxClustersDev {
    int32_t n_clusters;  // 0: the set has no two-level form
    int32_t pad;
    const BoxPair* boxes;     // per cluster, padded to whole groups of SCREEN_GROUP
    const BoxPair* pairs;     // CLUSTER_SIZE / 2 per cluster; a padding slot has a negative half-width: never a candidate
    const uint16_t* members;  // CLUSTER_SIZE per cluster: index in the set, NO_MEMBER for padding
};
// What the double-precision test of box j needs, in one 64-byte record (a candidate's index leads straight to it:
// no walk from the set's node list to the node).
struct BoxExact {
    double corner[6];    // box_min, box_max (B2/box.h:28-30)
    int32_t first_side;  // node id of the first of the six sides
    int32_t pad[3];
};
struct BoxSetDev {
    int32_t n;
    int32_t pad0;
    double cover;           // the margin holds for rays whose origin coordinates are at most this in magnitude
    const BoxPair* screen;  // padded to whole groups of SCREEN_GROUP boxes
    const BoxExact* exact;  // per box, in set order
    BoxClustersDev clusters;
};
// The leaves of a DEV_NODE_LEAF_SET node: spheres, then boxes.
// The subtree itself, for rays the float screens cannot hold (a NaN, an infinity, values whose products overflow: the
// reference's own comparisons then come out false and order decides everything).  Its nodes in depth-first order; a
// bvh_node whose box test fails is left by its skip link, so no stack is needed.
struct ReplayNode {
    double box[6];   // bvh_node: its box (B2/bvh.h:98)
    int32_t skip;    // bvh_node: index of the node after its subtree
    int32_t member;  // leaf: its index in the set (spheres first, then boxes); -1 for a bvh_node
};
struct LeafSetDev {
    SphereSetDev spheres;
    BoxSetDev boxes;
    const ReplayNode* replay;
    int32_t n_replay;
    int32_t pad;
};
// What shading a hit on sphere j needs beyond its SphereExact record, in one 16-byte load.
struct SphereShade {
    double inv_radius;  // 1.0 / radius: the factor of (p - center) / radius, B1/sphere.h:46, B1/vec3.h:96-98
    int32_t material_kind;  // RT_MAT_* of its material, known one load earlier than the material itself
    int32_t pad;
};
struct SphereListDev {
    SphereSetDev set;
    const SphereShade* shade_of;
    const MaterialDev* materials;  // per sphere (a copy of its material): a hit's record and material come back together
    int32_t sky_mode;
    double background[3];
    // book-2 sphere worlds only (null for a book-1 list)
    const TextureDev* textures;        // solid and checker textures of the lambertian materials
    const int32_t* top_of;              // per sphere: index within the root list of the top-level object holding it
};

struct CameraDev {
    double origin[3], llc[3], horizontal[3], vertical[3], u[3], v[3];
    double lens_radius, time0, time1;
};

// pixel -> (column, row) without an integer division per sample: n / width = (n * mul) >> shift, exact for n < 2^28
// (mul = ceil(2^shift / width), shift = 28 + ceil(log2 width): the error term n e / (width 2^shift), e < width, stays
// below 1 / width)
struct WidthDivider {
    uint32_t mul, shift;
};
inline WidthDivider make_width_divider(int32_t width) {
    uint32_t l = 0;
    while ((1u << l) < (uint32_t)width) ++l;
    WidthDivider d;
    d.shift = 28u + l;
    d.mul = (uint32_t)((((uint64_t)1 << d.shift) + (uint64_t)width - 1u) / (uint64_t)width);
    return d;
}

// n / d for every 32-bit n (Granlund and Montgomery's round-up form): t = high word of n * mul,
// q = (t + ((n - t) >> sh1)) >> sh2 -- five vector instructions where the division itself takes about twenty-five.
struct U32Divider {
    uint32_t mul, sh1, sh2;
};
#if defined(__HIPCC__)
__host__ __device__
#endif
inline U32Divider make_u32_divider(uint32_t d) {
    if (d == 0u) d = 1u;
    uint32_t l = 0;
    while (l < 32u && ((uint64_t)1 << l) < (uint64_t)d) ++l;
    U32Divider v;
    v.mul = (uint32_t)(((((uint64_t)1 << l) - (uint64_t)d) << 32) / (uint64_t)d + 1u);
    v.sh1 = l < 1u ? l : 1u;
    v.sh2 = l < 1u ? 0u : l - 1u;
    return v;
}
#if defined(__HIPCC__)
__host__ __device__
#endif
inline uint32_t div_u32(uint32_t n, const U32Divider& v) {
    const uint32_t t = (uint32_t)(((uint64_t)n * (uint64_t)v.mul) >> 32);
    return (t + ((n - t) >> v.sh1)) >> v.sh2;
}

// How the samples of a frame are cut into work items (make_item_plan, rt_render_common.h).
struct ItemPlan {
    int32_t cap, K, R, levels;
    U32Divider by_levels, by_npix;  // item number -> pixel (pixel-by-pixel plan) or level (frame-by-frame plan)
    uint32_t npix;
    uint32_t bulk_items;  // pixel-by-pixel plan: npix * levels (every item); else 0
    uint32_t batch;       // items a wave takes from the queue at a time, at most 64
    int32_t exclusive;    // 1: every item of a pixel goes to the same wave (whole pixels per batch)
};

// Threads per workgroup of the sphere-list render kernels (a multiple of 64).  Their waves share nothing -- each has its
// own slices of the workgroup's LDS -- so one wave per workgroup: a wave that ends gives its registers and LDS back by
// itself instead of waiting for the slowest of four, and the next launch on another stream (frames in flight) finds
// room sooner (config 2 69.35 -> 69.05 ms, a rank's 63 samples with three frames in flight 9.45 -> 9.39 ms; 128: no gain).
#ifndef RT_SPHERES_WG
#define RT_SPHERES_WG 64
#endif

struct RenderArgs {
    SphereListDev scene;
    CameraDev cam;
    int32_t width, height;
    WidthDivider wdiv;
    double width_m1, height_m1;  // (double)(width - 1), (double)(height - 1): the jitter's divisors, B1/RayTracing.cpp:120-121
    int32_t sample_begin, sample_count;
    int32_t max_depth;
    int32_t item_cap;       // longest run of samples of one pixel that forms a work item (see ItemPlan)
    int32_t pixel_pass;     // 1: runs are numbered pixel by pixel and summed in the accumulation cache (ItemPlan)
    ItemPlan plan;          // make_item_plan(sample_count, item_cap, width * height, pixel_pass), worked out by the host
    uint32_t n_items;       // number of work items
    uint32_t key0, key1;    // Philox key
    int32_t exact_only;     // 1: skip the float screen, test every sphere in double (validation)
    int64_t* accum;         // [3*width*height] fixed point, added to with atomics
    uint32_t* item_counter; // work-queue head, zeroed before launch
    uint64_t* stats;        // [4]: samples, rays, non-finite samples, overflow rays
#ifdef RT_TAIL_PROBE
    uint64_t* probe;  // debug build only: 8 words per wave (timestamps and iteration counts)
#endif
};

struct FirstHitArgs {
    SphereListDev scene;
    CameraDev cam;
    int32_t width, height;
    int32_t exact_only;
    int32_t* ids;
};

// ---- device layout of a general scene graph (book 2/3 worlds)
//
// One SceneNode per rt_node, same index, so that ids (first-hit gate, Philox keys of media) need no
// translation.  64 bytes: four 16-byte vector loads per visit.  The graph, materials, textures, Perlin
// tables and images of the largest reference scene total about 0.5 MB: resident in every XCD's L2.
// Device-only node kind: a box whose six sides are those the reference's constructor derives from its corners and are
// numbered consecutively; the traversal tests the sides from the corners in one step.
constexpr int DEV_NODE_BOX6 = 64;
// Device-only node kind: a BVH subtree whose leaves are all static spheres or boxes (of the DEV_NODE_BOX6 kind) and
// whose node boxes enclose them, directly under a top-level object's translate / rotate_y chain.  The traversal
// tests the subtree's own box, then finds the closest leaf with a wave-wide float screen (the sphere-list kernels'
// for the spheres, a slab test for the boxes) and the reference's double-precision test of the few candidates,
// instead of walking the tree.
constexpr int DEV_NODE_LEAF_SET = 65;
struct SceneNode {
    int32_t kind;  // RT_NODE_* or DEV_NODE_BOX6
    int32_t a;     // sphere/rect/medium: material; LIST/BOX: first child slot; BVH: left; instance/medium: child
    int32_t b;     // LIST/BOX: child count; BVH: right; medium: boundary node
    int32_t c;     // moving sphere: index of {time0, time1, radius} in extra[]; DEV_NODE_BOX6: node id of the first side
                   // DEV_NODE_LEAF_SET: a = index in sets[], b = first slot in set_nodes[], p = the subtree's box
    double p[6];   // sphere: c, r | moving sphere: c0, c1 | rect: a0 a1 b0 b1 k | BVH: box | box: corners | translate: offset
                   // | rotate_y: sin, cos | medium: neg_inv_density
};
// One entry per object of the world list, in list order: what the wave-level walk needs to know about it in one
// scalar load -- the instance chain above it (outermost first) and a copy of the node below the chain.
struct TopEntry {
    int32_t node;       // the node below the chain (the object itself when chain_len == 0)
    int32_t chain_len;  // translate / rotate_y / flip_face nodes above it, at most XF_DEPTH_CAP
    int32_t chain[4];
    int32_t first;      // the object's own node id (= chain[0] when there is a chain)
    int32_t pad;
    SceneNode end;      // copy of nodes[node]
    int32_t chain_kind[4];   // RT_NODE_TRANSLATE / ROTATE_Y / FLIP_FACE per chain level
    double chain_p[4][3];    // translate: offset; rotate_y: sin, cos
};
struct TextureDev {
    int32_t kind, a, b, pad;  // RT_TEX_*; checker: even, odd; noise: perlin index; image: image index or -1
    double p[4];              // solid: colour; noise: scale
};
struct ImageDev {
    int32_t width, height;
    const uint8_t* rgb;
};
constexpr int MAX_LIGHT_SHAPES = 16;
struct SceneDev {
    int32_t root;
    int32_t n_nodes;
    const SceneNode* nodes;
    const int32_t* children;
    const int32_t* parent_xf;  // per node: enclosing translate/rotate_y node, or -1
    const int32_t* top_of;     // per node: index within the root list of the top-level object containing it
    const TopEntry* tops;      // per object of the root list
    int32_t n_tops;
    const double* extra;
    const MaterialDev* materials;
    const MaterialDev* node_materials;  // materials[nodes[i].a] per node i: the hit's material without waiting for its node
    const TextureDev* textures;
    const rt_perlin* perlins;
    const ImageDev* images;
    const LeafSetDev* sets;    // one per DEV_NODE_LEAF_SET node: the screening images of its leaves
    const int32_t* set_nodes;  // node ids of the leaves of all sets: per set its spheres, then its boxes
    const int32_t* set_order;  // same slots: the leaf's position in the depth-first walk of its subtree (tie rule)
    int32_t sky_mode;
    int32_t n_lights;                  // light shapes sampled under RT_RENDER_IMPORTANCE
    int32_t lights[MAX_LIGHT_SHAPES];  // their node ids (xz_rect or sphere, world space)
    int32_t features;  // F_* bits of rt_kernels_scene.hip the world needs
    double background[3];
};
constexpr int STACK_CAP = 40;   // traversal stack slots per lane (uint32 each, in LDS); a leaf set, evaluated while the
                                // stack is empty, keeps its candidate lists in the same memory
constexpr int XF_DEPTH_CAP = 4;  // nested translate / rotate_y levels

struct SceneRenderArgs {
    SceneDev scene;
    CameraDev cam;
    int32_t width, height;
    WidthDivider wdiv;
    double width_m1, height_m1;
    int32_t sample_begin, sample_count;
    int32_t max_depth;
    int32_t item_cap;
    uint32_t n_items;
    uint32_t key0, key1;
    int32_t importance;  // 1: ray_color of book 3 (RT_RENDER_IMPORTANCE)
    int32_t exact_only;  // 1: leaf sets skip their float screens and test every leaf in double (validation)
    int64_t* accum;
    uint32_t* item_counter;
    uint64_t* stats;
};
struct SceneFirstHitArgs {
    SceneDev scene;
    CameraDev cam;
    int32_t width, height;
    int32_t exact_only;
    int32_t* ids;
};
hipError_t launch_render_scene(const SceneRenderArgs& args, int blocks, hipStream_t stream);
hipError_t launch_first_hit_scene(const SceneFirstHitArgs& args, hipStream_t stream);
int render_scene_blocks_per_cu(int features, bool importance);
#ifdef RT_SCENE_PROBE
void scene_probe_dump();  // debug build: prints and clears the traversal counters
#endif
int scene_feature_bit_for_node(int node_kind);  // F_* bit a node / texture / material kind needs, 0 if none
int scene_feature_bit_for_texture(int texture_kind);
int scene_feature_bit_for_material(int material_kind);
int scene_feature_bit_for_mixed_set();
int scene_feature_bit_for_top(int end_kind);  // the stack-walk bit if a world object of this (device) kind needs it

// Kernel launchers (rt_kernels_spheres.hip).  `blocks` = number of 256-thread workgroups.
// `timed` selects the variant for book-2 sphere worlds (shutter time, moving spheres, checker textures).
hipError_t launch_render_spheres(const RenderArgs& args, bool timed, int blocks, hipStream_t stream);
hipError_t launch_first_hit_spheres(const FirstHitArgs& args, bool timed, hipStream_t stream);
// Resident 256-thread workgroups per CU of the render kernel, as the runtime reports it.
int render_spheres_blocks_per_cu(bool timed);

}  // namespace rtd
