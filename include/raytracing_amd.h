/*
 * raytracing_amd.h -- C ABI of the MI355X (gfx950) path-tracing library.
 *
 * The reference program (huris/RayTracing) has no plugin or FFI layer: each book is one
 * translation unit whose main() builds a `hittable_list`, a `camera`, and then runs the
 * sample loop -> ray_color() -> hittable::hit / material::scatter.  This header puts a C
 * boundary at exactly that seam.  The caller (a reference-shaped C++ scene builder, or a
 * ctypes/cgo/JNI stub) hands over
 *   - the scene as plain arrays that mirror the reference's object graph one-to-one,
 *   - the camera as the vectors the reference's camera constructor computes,
 *   - the loop bounds of the reference's render loop,
 * and receives the per-pixel colour SUM the reference hands to write_color().
 *
 * Everything crossing the boundary is a POD struct, a plain pointer or a size.  All
 * geometry is IEEE double, like the reference ("Ray Tracing In a Weekend/code/vec3.h":63).
 *
 * Reference citations use B1 = "Ray Tracing In a Weekend/code", B2 = "Ray Tracing The Next
 * Week/code" (B3, "The Rest Of Your Life/code", is a byte-identical copy of B2).
 */
#ifndef RAYTRACING_AMD_H
#define RAYTRACING_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 1

/* ------------------------------------------------------------------ error codes */
enum {
    RT_OK = 0,
    RT_ERR_INVALID_ARGUMENT = 1,   /* null pointer, bad size, bad index in a scene array   */
    RT_ERR_UNSUPPORTED_SCENE = 2,  /* scene graph uses a construct no kernel covers         */
    RT_ERR_NO_DEVICE = 3,          /* no gfx950 device visible / bad device ordinal        */
    RT_ERR_HIP = 4,                /* a HIP runtime call failed (see rt_last_error_detail) */
    RT_ERR_NO_SCENE = 5,           /* render / first-hit called before rt_scene_upload     */
    RT_ERR_OUT_OF_MEMORY = 6,
    RT_ERR_LIMIT = 7,              /* a documented size limit was exceeded                 */
    RT_ERR_COMM = 8                /* RCCL could not be loaded or a collective call failed  */
};

/* ------------------------------------------------------------------ scene graph
 * One rt_node per reference `hittable` object.  `kind` names the reference class; `p[]`
 * holds that class's public data members in declaration order (documented per kind).   */
enum {
    RT_NODE_SPHERE = 1,        /* B1/sphere.h:9-21      p[0..2]=center p[3]=radius (may be <0)        */
    RT_NODE_MOVING_SPHERE = 2, /* B2/moving_sphere.h:8-25 p[0..2]=center0 p[3..5]=center1
                                                         p[6]=time0 p[7]=time1 p[8]=radius.  Its hits, and a
                                                         medium's, carry u = v = 0 (the reference leaves the
                                                         record's u,v indeterminate there)             */
    RT_NODE_XY_RECT = 3,       /* B2/aarect.h:7-26      p[0]=x0 p[1]=x1 p[2]=y0 p[3]=y1 p[4]=k        */
    RT_NODE_XZ_RECT = 4,       /* B2/aarect.h:49-68     p[0]=x0 p[1]=x1 p[2]=z0 p[3]=z1 p[4]=k        */
    RT_NODE_YZ_RECT = 5,       /* B2/aarect.h:91-110    p[0]=y0 p[1]=y1 p[2]=z0 p[3]=z1 p[4]=k        */
    RT_NODE_BOX = 6,           /* B2/box.h:9-25         p[0..2]=box_min p[3..5]=box_max;
                                                         children[child0 .. child0+child1) = 6 sides  */
    RT_NODE_LIST = 7,          /* B1/hittable_list.h:12-24  children[child0 .. child0+child1)          */
    RT_NODE_BVH = 8,           /* B2/bvh.h:10-27        child0=left child1=right p[0..2]=box.min
                                                         p[3..5]=box.max                               */
    RT_NODE_TRANSLATE = 9,     /* B2/hittable.h:31-41   child0=ptr p[0..2]=offset                     */
    RT_NODE_ROTATE_Y = 10,     /* B2/hittable.h:62-79   child0=ptr p[0]=sin_theta p[1]=cos_theta
                                                         p[2]=hasbox p[3..5]=bbox.min p[6..8]=bbox.max */
    RT_NODE_FLIP_FACE = 12,    /* flip_face of "The Rest Of Your Life" (public book text; the reference repository
                                  holds no book-3 code): child0=ptr; hit() of the child with front_face negated  */
    RT_NODE_MEDIUM = 11        /* B2/constant_medium.h:10-28 child0=boundary p[0]=neg_inv_density,
                                                         material = the isotropic phase function       */
};

typedef struct rt_node {
    int32_t kind;
    int32_t material; /* index into materials[], -1 for containers                          */
    int32_t child0;
    int32_t child1;
    double p[12];
} rt_node;

enum {
    RT_MAT_LAMBERTIAN = 1,    /* B1/material.h:15-34, B2/material.h:20-40  texture = albedo  */
    RT_MAT_METAL = 2,         /* B1/material.h:36-52   albedo[3], param = fuzz (already <=1)  */
    RT_MAT_DIELECTRIC = 3,    /* B1/material.h:54-89   param = ir                              */
    RT_MAT_DIFFUSE_LIGHT = 4, /* B2/material.h:97-114  texture = emit                          */
    RT_MAT_ISOTROPIC = 5      /* B2/material.h:116-131 texture = albedo                        */
};

typedef struct rt_material {
    int32_t kind;
    int32_t texture; /* index into textures[] or -1 */
    double albedo[3];
    double param;
} rt_material;

enum {
    RT_TEX_SOLID = 1,   /* B2/texture.h:15-28  p[0..2] = color_value                           */
    RT_TEX_CHECKER = 2, /* B2/texture.h:30-46  a = even texture, b = odd texture                */
    RT_TEX_NOISE = 3,   /* B2/texture.h:48-60  a = index into perlins[], p[0] = scale           */
    RT_TEX_IMAGE = 4    /* B2/texture.h:62-110 a = index into images[], -1 = no data (cyan)     */
};

typedef struct rt_texture {
    int32_t kind;
    int32_t a;
    int32_t b;
    int32_t reserved;
    double p[4];
} rt_texture;

/* B2/perlin.h:63-68: the four tables a `perlin` object rolls in its constructor. */
typedef struct rt_perlin {
    double ranvec[256][3];
    int32_t perm_x[256];
    int32_t perm_y[256];
    int32_t perm_z[256];
} rt_perlin;

/* Decoded RGB8 image, rows top to bottom, as stbi_load(…, 3) returns it (B2/texture.h:71). */
typedef struct rt_image {
    int32_t width;
    int32_t height;
    const uint8_t* rgb;
} rt_image;

/* What a ray that leaves the world returns.  Lights (RT_MAT_DIFFUSE_LIGHT) emit under either mode. */
enum {
    RT_SKY_GRADIENT = 0,  /* B1/RayTracing.cpp:28-30: white -> (0.5,0.7,1.0) by unit dir y    */
    RT_SKY_BACKGROUND = 1 /* B2/RayTracing.cpp:25-26: constant `background` colour             */
};

typedef struct rt_scene_desc {
    int32_t abi_version; /* RT_ABI_VERSION */
    int32_t root;        /* index of the world node (a RT_NODE_LIST in every reference scene) */
    int32_t n_nodes;
    int32_t n_children;
    int32_t n_materials;
    int32_t n_textures;
    int32_t n_perlins;
    int32_t n_images;
    const rt_node* nodes;
    const int32_t* children; /* child index lists of LIST / BOX nodes */
    const rt_material* materials;
    const rt_texture* textures;
    const rt_perlin* perlins;
    const rt_image* images;
    int32_t sky_mode;
    int32_t lights_plus1; /* 0 = none; else 1 + index of a RT_NODE_LIST (not reachable from root) whose children
                             are the shapes light is sampled towards under RT_RENDER_IMPORTANCE: RT_NODE_XZ_RECT
                             and RT_NODE_SPHERE, in world space (book 3's `lights` argument of ray_color)      */
    double background[3];
} rt_scene_desc;

/* The private members of the reference camera after its constructor has run
 * (B1/camera.h:41-47, B2/camera.h:47-54).  rt_camera_init() computes them on the host
 * exactly as the constructor does.                                                      */
typedef struct rt_camera_desc {
    double origin[3];
    double lower_left_corner[3];
    double horizontal[3];
    double vertical[3];
    double u[3];
    double v[3];
    double w[3];
    double lens_radius;
    double time0;
    double time1;
} rt_camera_desc;

/* Loop bounds of the reference render loop (B1/RayTracing.cpp:82-86,115-127).
 * Samples of a pixel are numbered globally 0..spp-1; a call renders the half-open range
 * [sample_begin, sample_begin + sample_count) of every pixel, so that a frame split over
 * several GPUs (or several calls) draws exactly the random numbers of the unsplit frame. */
typedef struct rt_render_params {
    int32_t width;
    int32_t height;
    int32_t sample_begin;
    int32_t sample_count;
    int32_t max_depth;
    int32_t flags; /* RT_RENDER_* */
    uint64_t seed; /* Philox key */
} rt_render_params;

enum {
    RT_RENDER_DEFAULT = 0,
    RT_RENDER_EXACT_ONLY = 1, /* validation: skip the float screening pass, test every object in double */
    RT_RENDER_IMPORTANCE = 2  /* ray_color of "The Rest Of Your Life" (final listing of the public book text; no
                                 reference code exists for it, so parity for this mode is unpinned): lambertian
                                 surfaces sample a 50/50 mixture of the cosine density and the density towards
                                 the scene's light shapes, metal and dielectric are specular, lights emit from
                                 their front face only.  Needs lights_plus1 != 0; scene-graph kernels only      */
};

/* Colour sums are accumulated in 64-bit fixed point so that the sum over samples does not
 * depend on the order in which lanes, waves or GPUs add them: value = fixed * 2^-RT_ACCUM_FRAC_BITS. */
#define RT_ACCUM_FRAC_BITS 36
/* Range of the format.  A sample enters the sum only if every channel is below RT_ACCUM_SAMPLE_LIMIT in magnitude
 * (2^26: its fixed-point image fits 63 bits with room to spare); any other sample -- NaN and infinities included -- is
 * dropped and counted in rt_stats.nan_samples.  A pixel's SUM over all samples must stay below 2^27 in magnitude (the
 * 64-bit sum wraps beyond; with the reference's lights of at most 15 that is 9 million samples per pixel). */
#define RT_ACCUM_SAMPLE_LIMIT 67108864.0

typedef struct rt_stats {
    uint64_t samples;       /* camera rays started                                   */
    uint64_t rays;          /* world.hit() calls (ray segments), as the reference counts them */
    uint64_t nan_samples;   /* samples dropped: NaN, Inf or beyond RT_ACCUM_SAMPLE_LIMIT */
    uint64_t overflow_rays; /* rays that took the slow exact path (candidate list full) */
    uint64_t kernel_launches; /* render kernel launches timed into kernel_ms (at most 256 per take) */
    double kernel_ms;       /* summed device time of those launches, HIP events on the kernel's stream */
} rt_stats;

typedef struct rt_context rt_context;

/* ------------------------------------------------------------------ entry points */

/* Number of HIP devices visible to this process (0 if none; never fails). */
int rt_get_device_count(void);

/* Create a context bound to one device.  One process per GPU is the intended use; a
 * context is used from one host thread at a time.  `stream` may be NULL (the library then
 * creates its own stream) or an existing hipStream_t on that device passed as void*.    */
int rt_context_create(int device_ordinal, void* stream, rt_context** out);
void rt_context_destroy(rt_context* ctx);

/* Copy the scene to the device (replaces any earlier scene of this context).  The caller
 * keeps ownership of every host array; nothing is referenced after the call returns.
 * Stands in for the world handed to ray_color (B1/RayTracing.cpp:90, B2/RayTracing.cpp:271-351). */
int rt_scene_upload(rt_context* ctx, const rt_scene_desc* scene);

/* The reference sample loop + ray_color (B1/RayTracing.cpp:11-31,115-127;
 * B2/RayTracing.cpp:17-36,365-377).  Adds the colour of samples
 * [sample_begin, sample_begin+sample_count) of every pixel and writes the sums, as doubles,
 * to the HOST array accum_rgb_out[3*width*height] (row 0 = top image row, i.e. the
 * reference's j = height-1; channel order r,g,b).  Synchronous.  stats may be NULL.     */
int rt_render(rt_context* ctx, const rt_camera_desc* cam, const rt_render_params* params,
              double* accum_rgb_out, rt_stats* stats);

/* Same work, but the result stays on the device as 64-bit fixed point (see
 * RT_ACCUM_FRAC_BITS) in the caller's DEVICE buffer d_accum_fixed[3*width*height]
 * (e.g. a torch int64 tensor), which the call ADDS to: zero it before the first call of a
 * frame.  The kernel is enqueued on the context's stream and the call returns without
 * waiting; rt_context_synchronize() waits.  Calls that add to the same buffer must be
 * ordered (one stream, or synchronised): from 32 samples per pixel on a pixel's
 * sums reach memory by plain loads and stores, not atomics.  This is the entry point a multi-GPU caller
 * uses before summing the buffers of all ranks with an all-reduce.                      */
int rt_render_device(rt_context* ctx, const rt_camera_desc* cam, const rt_render_params* params,
                     int64_t* d_accum_fixed);
int rt_context_synchronize(rt_context* ctx);
/* Whether rt_render_device would accept these arguments on this context (RT_OK, or the code and detail it would
 * return), without enqueuing anything: every refusal rt_render_device can make short of a HIP runtime error.  A
 * caller that must not start a frame it cannot finish everywhere (rt_group_render_async: its peers would wait in the
 * all-reduce) asks first.  The reference has no counterpart: its loop bounds are literals (B1/RayTracing.cpp:82-86). */
int rt_render_check(rt_context* ctx, const rt_camera_desc* cam, const rt_render_params* params);
/* The hand-over of the sums to write_color (B1/RayTracing.cpp:125: `write_color(std::cout, pixel_color, ...)`) for a
 * frame that rt_render_device left on the device: waits for the context's stream, converts the fixed-point sums
 * d_accum_fixed[3*width*height] (DEVICE memory) to doubles on the device and copies them to the HOST array
 * accum_rgb_out through a pinned staging buffer the context keeps.  rt_render = zero + rt_render_device + this. */
int rt_read_sums(rt_context* ctx, const int64_t* d_accum_fixed, int width, int height, double* accum_rgb_out);
/* rt_render's work without a single allocation.  It has no counterpart in the reference; it exists for callers that
 * cannot call hipMalloc: code under HIP graph capture, tight serving loops, harnesses that forbid allocation.  The
 * samples of the call are summed in a scratch frame that lives in the library's code object (a file-scope device
 * array of 256 + 24 * RT_NOALLOC_MAX_PIXELS bytes, about 24 MiB, which every GPU the library runs on holds until the
 * process ends) and leave as DOUBLE sums in the caller's DEVICE buffer d_accum_rgb_out[3*width*height] (row 0 = top,
 * r,g,b; overwritten, not added to).  Enqueued on the context's stream: rt_context_synchronize waits.  The call makes
 * no hipMalloc, hipFree, event creation or other allocating call, and no host allocation when it succeeds; it
 * returns RT_ERR_LIMIT when width*height exceeds RT_NOALLOC_MAX_PIXELS.  All contexts of one GPU share that GPU's
 * array: calls that overlap in time on one GPU race (calls on one stream, or on one context, never overlap).  Its
 * launches are not timed and its samples and rays are not reported by rt_context_take_stats. */
#define RT_NOALLOC_MAX_PIXELS (1 << 20)
int rt_render_noalloc(rt_context* ctx, const rt_camera_desc* cam, const rt_render_params* params,
                      double* d_accum_rgb_out);
/* Validation switch: when on, kernels skip their float screening pass and test every object
 * in double.  Results must be identical either way; only the speed differs. */
int rt_context_set_exact_only(rt_context* ctx, int on);
/* Validation switch, read by the next rt_scene_upload: when on, a world of spheres is laid out and
 * rendered as a general scene graph instead of by the dedicated sphere-list kernels.  Results must be
 * identical either way. */
int rt_context_set_force_graph(rt_context* ctx, int on);
/* Which kernels would render the uploaded world for this camera: RT_KERNELS_SPHERE_LIST (a flat list of
 * spheres, B1/hittable_list.h; or a book-2 world whose only primitives are spheres and moving spheres --
 * B2/moving_sphere.h -- under any nesting of B2/bvh.h nodes, walked as the list of its leaves while the camera's
 * shutter interval lies inside every moving sphere's [time0, time1]), RT_KERNELS_SCENE_GRAPH (everything else),
 * or RT_KERNELS_NONE (no world uploaded, or null arguments).  Results do not depend on the answer. */
#define RT_KERNELS_NONE 0
#define RT_KERNELS_SPHERE_LIST 1
#define RT_KERNELS_SCENE_GRAPH 2
int rt_context_kernels_for(const rt_context* ctx, const rt_camera_desc* cam);
/* Statistics of everything rendered since the last call of this function (synchronises). */
int rt_context_take_stats(rt_context* ctx, rt_stats* out);

/* Closest-hit object of the ray through each pixel CENTRE, s=(i+0.5)/(width-1),
 * t=(j+0.5)/(height-1), lens sample (0,0), shutter time time0: the index, within the root
 * list, of the object whose hit the reference's hittable_list::hit would keep
 * (B1/hittable_list.h:26-39, t_min = 0.001, later object wins exact ties), or -1 for a
 * miss.  ids_out is a HOST array of width*height int32, row 0 = top row.  Computed in
 * double with the reference's operation order; meant to be compared bit for bit.        */
int rt_first_hit(rt_context* ctx, const rt_camera_desc* cam, int width, int height,
                 int32_t* ids_out);

/* ------------------------------------------------------------------ several GPUs
 * The reference is single-threaded; its only word on parallelism is the author's recipe "copy the code N times with
 * different seeds onto N cores, then average the images" ("Ray Tracing In a Weekend/README.md":2538-2542).  A group
 * is that recipe made exact: the world is replicated on every GPU, member r of n renders the samples
 * rt_split_samples(sample_count, r, n) of every pixel (global sample indices, so the random numbers are those of the
 * unsplit frame), and the members' fixed-point sums are added by ONE RCCL all-reduce (ncclInt64, ncclSum) per frame.
 * Integer sums make the n-GPU frame equal the one-GPU frame bit for bit.  The group owns one RCCL communicator per
 * local member, one HIP stream, rendering context and accumulation buffer per member and frame slot; RCCL is loaded
 * (dlopen of librccl.so.1) when the first group is created.  A group is used from one host thread at a time.     */
typedef struct rt_group rt_group;

/* Member `index` of `n` renders samples [*begin, *begin + *count) of a frame of `sample_count` samples per pixel:
 * the first sample_count % n members get one sample more (500 over 8: 63,63,63,63,62,62,62,62).  Host arithmetic
 * only; bad arguments give *count = 0. */
void rt_split_samples(int sample_count, int index, int n, int* begin, int* count);

/* One process driving n GPUs (device_ordinals[n], all different): the communicators come from ncclCommInitAll.
 * frame_slots (1..8) = frames that may be in flight at once, each on its own stream and buffers. */
int rt_group_create(const int* device_ordinals, int n, int frame_slots, rt_group** out);

/* One process per GPU: rank 0 makes an id (ncclGetUniqueId), hands its bytes to the other ranks by any means, and
 * every rank calls rt_group_create_rank with the same id (ncclCommInitRank; collective over the ranks). */
#define RT_COMM_ID_BYTES 128
typedef struct rt_comm_id {
    char bytes[RT_COMM_ID_BYTES];
} rt_comm_id;
int rt_comm_id_create(rt_comm_id* out);
int rt_group_create_rank(int device_ordinal, int rank, int n_ranks, const rt_comm_id* id, int frame_slots,
                         rt_group** out);
void rt_group_destroy(rt_group* g);

int rt_group_size(const rt_group* g);        /* members over all processes (the n of rt_split_samples)          */
int rt_group_local_count(const rt_group* g); /* members driven by this process                                  */
int rt_group_frame_slots(const rt_group* g);
/* The rendering context of local member `local_index` on frame slot `slot` (validation switches, statistics);
 * owned by the group. */
rt_context* rt_group_context(rt_group* g, int local_index, int slot);

/* rt_scene_upload on every local context. */
int rt_group_scene_upload(rt_group* g, const rt_scene_desc* scene);

/* One frame of the reference render loop (as rt_render) over all members: returns when the summed frame is on the
 * host.  params->sample_begin / sample_count describe the WHOLE frame; every process of the group must call it with
 * the same arguments.  stats (may be NULL) sums the local members.  Uses frame slot 0. */
int rt_group_render(rt_group* g, const rt_camera_desc* cam, const rt_render_params* params, double* accum_rgb_out,
                    rt_stats* stats);
/* The same frame without waiting: on slot `slot`, every local member zeroes its buffer, renders its share and joins
 * the all-reduce, all enqueued on the slot's streams.  rt_group_wait blocks until the slot's summed frame is
 * complete on every local member; rt_group_read then hands it over as rt_read_sums does (from local member 0);
 * rt_group_device_sums is the buffer itself (DEVICE memory of that member's GPU, valid until the slot's next frame
 * or rt_group_destroy).  A slot must be waited for before it is given another frame.
 *
 * Failures and the collective.  Everything that can be refused (arguments, no world, limits: rt_render_check) is
 * checked for every local member, and the slot's buffers are grown, BEFORE anything is enqueued: such an error means
 * that nothing was started, and since all processes make the same call on the same world it is returned on all of
 * them.  If a member's enqueue fails after that (a HIP runtime error), the call still joins the all-reduce with every
 * member -- no peer is left waiting in it -- and returns the error; the failed member's share is missing from the
 * sums, and a failure word that travels through the all-reduce behind the sums makes rt_group_wait return RT_ERR_COMM
 * for that frame on EVERY rank.  The group stays usable.  Only an error of the all-reduce itself (RT_ERR_COMM from
 * this call) leaves the communicator unusable: destroy the group.
 * The slots of a member share one communicator; the all-reduces of successive frames are ordered across the slots'
 * streams by an event each records and the next one waits for, in the order the frames were submitted (the same on
 * every rank). */
int rt_group_render_async(rt_group* g, const rt_camera_desc* cam, const rt_render_params* params, int slot);
int rt_group_wait(rt_group* g, int slot);
/* Allocates the accumulation buffers of every slot for frames of width x height now, so that rt_group_render_async
 * only enqueues work (it grows them itself otherwise, on a frame's first use of a slot, before anything is started). */
int rt_group_reserve(rt_group* g, int width, int height);
/* Validation switch: the next rt_group_render_async treats local member `local_index` as if its kernel launch had
 * failed with a HIP error (exercises the path above on a healthy machine). */
int rt_group_set_fail_next(rt_group* g, int local_index);
int rt_group_read(rt_group* g, int slot, double* accum_rgb_out);
const int64_t* rt_group_device_sums(rt_group* g, int local_index, int slot);
/* Statistics of the local members since the last call (sums over members and slots; synchronises them). */
int rt_group_take_stats(rt_group* g, rt_stats* out);
const char* rt_group_last_error_detail(const rt_group* g); /* g == NULL: of the last failed creation */

/* Host helpers that restate small host-side pieces of the reference. */

/* camera::camera (B1/camera.h:8-32, B2/camera.h:8-36). */
void rt_camera_init(rt_camera_desc* out, const double lookfrom[3], const double lookat[3],
                    const double vup[3], double vfov_degrees, double aspect_ratio, double aperture,
                    double focus_dist, double time0, double time1);

/* write_color (B1/color.h:8-23) for a whole image: sums -> 8-bit gamma-2 rgb[3*w*h].
 * A NaN sum maps to 0 (the reference prints INT_MIN there). */
void rt_write_color(const double* accum_rgb, int width, int height, int samples_per_pixel,
                    uint8_t* rgb_out);

/* The reference's P3 output stream (B1/RayTracing.cpp:113,125): "P3\nW H\n255\n" then one
 * "r g b\n" line per pixel.  Returns RT_OK or RT_ERR_INVALID_ARGUMENT (file not writable). */
int rt_write_ppm(const char* path, const uint8_t* rgb, int width, int height);

const char* rt_error_string(int code);
/* Human-readable detail of the last failure on this context (or of context creation when
 * ctx is NULL).  The pointer stays valid until the next failing call.                   */
const char* rt_last_error_detail(const rt_context* ctx);

#ifdef __cplusplus
}
#endif
#endif /* RAYTRACING_AMD_H */
