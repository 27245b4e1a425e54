// C ABI of the library (include/raytracing_amd.h): context, scene upload, render, first hit.
// Host code only; the kernels live in rt_kernels_*.hip.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "raytracing_amd.h"
#include "rt_internal.h"

using namespace rtd;

struct rt_context {
    int device = -1;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int num_cus = 0;
    int blocks_per_cu = 0;
    std::string detail;
    bool exact_only = false;  // validation: bypass the float screen

    // scene (sphere list)
    bool have_scene = false;
    SphereListDev scene{};
    void* d_screen = nullptr;
    void* d_exact = nullptr;
    void* d_material_of = nullptr;
    void* d_materials = nullptr;

    // per-context device scratch, allocated once
    uint32_t* d_counter = nullptr;  // work-queue head
    uint64_t* d_stats = nullptr;    // 4 counters
    int64_t* d_accum = nullptr;     // fixed-point frame of rt_render (grown on demand)
    size_t accum_elems = 0;
    int32_t* d_ids = nullptr;
    size_t ids_elems = 0;
    // HIP event pairs around the render kernels launched since the last rt_context_take_stats
    std::vector<hipEvent_t> ev_pool;  // [2*i] = start, [2*i+1] = stop
    size_t ev_used = 0;               // pairs in use
};

static constexpr size_t MAX_TIMED_LAUNCHES = 256;

static thread_local std::string g_create_detail;

static int fail(rt_context* ctx, int code, const std::string& what) {
    if (ctx) ctx->detail = what;
    else g_create_detail = what;
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
    default: return "unknown error code";
    }
}

const char* rt_last_error_detail(const rt_context* ctx) { return ctx ? ctx->detail.c_str() : g_create_detail.c_str(); }

static void free_scene(rt_context* ctx) {
    (void)hipFree(ctx->d_screen);
    (void)hipFree(ctx->d_exact);
    (void)hipFree(ctx->d_material_of);
    (void)hipFree(ctx->d_materials);
    ctx->d_screen = ctx->d_exact = ctx->d_material_of = ctx->d_materials = nullptr;
    ctx->have_scene = false;
}

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
    if (e != hipSuccess) {
        int rc = hip_fail(nullptr, e, "rt_context_create");
        rt_context_destroy(ctx);
        return rc;
    }
    ctx->blocks_per_cu = render_spheres_blocks_per_cu();
    if (const char* s = std::getenv("RT_BLOCKS_PER_CU")) {
        int v = std::atoi(s);
        if (v >= 1 && v <= 8) ctx->blocks_per_cu = v;
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
    for (hipEvent_t ev : ctx->ev_pool) (void)hipEventDestroy(ev);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

// Validate the graph and, for a root list made of spheres only with book-1 materials, build
// the two device images of the list.
int rt_scene_upload(rt_context* ctx, const rt_scene_desc* sc) {
    if (!ctx) return RT_ERR_INVALID_ARGUMENT;
    if (!sc || sc->abi_version != RT_ABI_VERSION) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "scene null or abi_version mismatch");
    if (sc->n_nodes <= 0 || !sc->nodes || sc->root < 0 || sc->root >= sc->n_nodes)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "bad node array / root");
    const rt_node& root = sc->nodes[sc->root];
    if (root.kind != RT_NODE_LIST) return fail(ctx, RT_ERR_UNSUPPORTED_SCENE, "root must be a hittable_list");
    if (root.child1 < 0 || root.child0 < 0 || (int64_t)root.child0 + root.child1 > sc->n_children ||
        (root.child1 > 0 && !sc->children))
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "root child range outside children[]");
    const int n = root.child1;
    if (n > 65535) return fail(ctx, RT_ERR_LIMIT, "more than 65535 objects in a flat list");
    if (sc->sky_mode != RT_SKY_GRADIENT && sc->sky_mode != RT_SKY_BACKGROUND)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "bad sky_mode");

    const int n_pairs = (n + 1) / 2;
    std::vector<SpherePair> screen((size_t)std::max(1, n_pairs));
    std::vector<SphereExact> exact((size_t)std::max(1, n));
    std::vector<int32_t> material_of((size_t)std::max(1, n));
    for (auto& s : screen) {  // the pad of an odd list never yields a candidate: q = -huge
        for (int h = 0; h < 2; ++h) {
            s.cx[h] = s.cy[h] = s.cz[h] = 0.0f;
            s.k[h] = 3.0e38f;
        }
    }
    for (int i = 0; i < n; ++i) {
        const int ni = sc->children[root.child0 + i];
        if (ni < 0 || ni >= sc->n_nodes) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "child index out of range");
        const rt_node& nd = sc->nodes[ni];
        if (nd.kind != RT_NODE_SPHERE)
            return fail(ctx, RT_ERR_UNSUPPORTED_SCENE, "object " + std::to_string(i) + ": only spheres are supported in this build");
        if (nd.material < 0 || nd.material >= sc->n_materials || !sc->materials)
            return fail(ctx, RT_ERR_INVALID_ARGUMENT, "material index out of range");
        const double cx = nd.p[0], cy = nd.p[1], cz = nd.p[2], r = nd.p[3];
        exact[i] = SphereExact{cx, cy, cz, r};
        material_of[i] = nd.material;
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
    std::vector<MaterialDev> mats((size_t)std::max(1, sc->n_materials));
    for (int i = 0; i < sc->n_materials; ++i) {
        const rt_material& m = sc->materials[i];
        MaterialDev md{};
        md.kind = m.kind;
        md.param = m.param;
        switch (m.kind) {
        case RT_MAT_LAMBERTIAN: {
            if (m.texture < 0 || m.texture >= sc->n_textures || !sc->textures)
                return fail(ctx, RT_ERR_INVALID_ARGUMENT, "lambertian without a texture");
            const rt_texture& t = sc->textures[m.texture];
            if (t.kind != RT_TEX_SOLID)
                return fail(ctx, RT_ERR_UNSUPPORTED_SCENE, "only solid_color textures are supported in this build");
            md.ax = t.p[0];
            md.ay = t.p[1];
            md.az = t.p[2];
            break;
        }
        case RT_MAT_METAL:
            md.ax = m.albedo[0];
            md.ay = m.albedo[1];
            md.az = m.albedo[2];
            break;
        case RT_MAT_DIELECTRIC: break;
        default:
            return fail(ctx, RT_ERR_UNSUPPORTED_SCENE, "material kind " + std::to_string(m.kind) + " is not supported in this build");
        }
        mats[i] = md;
    }

    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    free_scene(ctx);
    HIP_TRY(ctx, hipMalloc(&ctx->d_screen, screen.size() * sizeof(SpherePair)));
    HIP_TRY(ctx, hipMalloc(&ctx->d_exact, exact.size() * sizeof(SphereExact)));
    HIP_TRY(ctx, hipMalloc(&ctx->d_material_of, material_of.size() * sizeof(int32_t)));
    HIP_TRY(ctx, hipMalloc(&ctx->d_materials, mats.size() * sizeof(MaterialDev)));
    HIP_TRY(ctx, hipMemcpy(ctx->d_screen, screen.data(), screen.size() * sizeof(SpherePair), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(ctx->d_exact, exact.data(), exact.size() * sizeof(SphereExact), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(ctx->d_material_of, material_of.data(), material_of.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(ctx->d_materials, mats.data(), mats.size() * sizeof(MaterialDev), hipMemcpyHostToDevice));

    SphereListDev& S = ctx->scene;
    S.n = n;
    S.n_full_groups = n / SCREEN_GROUP;
    S.n_tail_pairs = (n % SCREEN_GROUP + 1) / 2;
    S.screen = (const SpherePair*)ctx->d_screen;
    S.exact = (const SphereExact*)ctx->d_exact;
    S.material_of = (const int32_t*)ctx->d_material_of;
    S.materials = (const MaterialDev*)ctx->d_materials;
    S.sky_mode = sc->sky_mode;
    for (int c = 0; c < 3; ++c) S.background[c] = sc->background[c];
    ctx->have_scene = true;
    return RT_OK;
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

static int check_render_args(rt_context* ctx, const rt_camera_desc* cam, const rt_render_params* p) {
    if (!ctx) return RT_ERR_INVALID_ARGUMENT;
    if (!cam || !p) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "null camera or params");
    if (!ctx->have_scene) return fail(ctx, RT_ERR_NO_SCENE, "rt_scene_upload has not succeeded on this context");
    if (p->width < 2 || p->height < 2) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "width and height must be >= 2 (the reference divides by W-1, H-1)");
    if ((int64_t)p->width * p->height > (int64_t)1 << 28) return fail(ctx, RT_ERR_LIMIT, "more than 2^28 pixels");
    if (p->sample_begin < 0 || p->sample_count < 0 || (int64_t)p->sample_begin + p->sample_count > (int64_t)1 << 30)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "bad sample range");
    return RT_OK;
}

static int chunk_for(const rt_render_params* p) {
    int chunk = 8;
    if (const char* s = std::getenv("RT_ITEM_SAMPLES")) {
        int v = std::atoi(s);
        if (v >= 1) chunk = v;
    }
    if (chunk > p->sample_count) chunk = p->sample_count;
    return chunk < 1 ? 1 : chunk;
}

int rt_render_device(rt_context* ctx, const rt_camera_desc* cam, const rt_render_params* p, int64_t* d_accum_fixed) {
    int rc = check_render_args(ctx, cam, p);
    if (rc != RT_OK) return rc;
    if (!d_accum_fixed) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "null accumulation buffer");
    if (p->sample_count == 0 || p->max_depth <= 0) return RT_OK;  // depth <= 0: every sample is black
    HIP_TRY(ctx, hipSetDevice(ctx->device));

    RenderArgs A{};
    A.scene = ctx->scene;
    fill_camera(A.cam, cam);
    A.width = p->width;
    A.height = p->height;
    A.sample_begin = p->sample_begin;
    A.sample_count = p->sample_count;
    A.max_depth = p->max_depth;
    A.chunk = chunk_for(p);
    const int64_t chunks = ((int64_t)p->sample_count + A.chunk - 1) / A.chunk;
    const int64_t items = chunks * p->width * p->height;
    if (items > 0xFFFF0000ll) return fail(ctx, RT_ERR_LIMIT, "more than 2^32 work items; raise RT_ITEM_SAMPLES");
    A.n_items = (uint32_t)items;
    A.key0 = (uint32_t)p->seed;
    A.key1 = (uint32_t)(p->seed >> 32);
    A.exact_only = (ctx->exact_only || (p->flags & RT_RENDER_EXACT_ONLY)) ? 1 : 0;
    A.accum = d_accum_fixed;
    A.item_counter = ctx->d_counter;
    A.stats = ctx->d_stats;

    int blocks = ctx->num_cus * ctx->blocks_per_cu;
    const int64_t max_useful = (items + 255) / 256;
    if (blocks > max_useful) blocks = (int)max_useful;
    if (blocks < 1) blocks = 1;

    HIP_TRY(ctx, hipMemsetAsync(ctx->d_counter, 0, sizeof(uint32_t), ctx->stream));
    // time the kernel with an event pair on its own stream (the first MAX_TIMED_LAUNCHES launches
    // after each rt_context_take_stats)
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (ctx->ev_used < MAX_TIMED_LAUNCHES) {
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
    HIP_TRY(ctx, launch_render_spheres(A, blocks, ctx->stream));
    if (e1) {
        HIP_TRY(ctx, hipEventRecord(e1, ctx->stream));
        ++ctx->ev_used;
    }
    return RT_OK;
}

int rt_context_synchronize(rt_context* ctx) {
    if (!ctx) return RT_ERR_INVALID_ARGUMENT;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_context_set_exact_only(rt_context* ctx, int on) {
    if (!ctx) return RT_ERR_INVALID_ARGUMENT;
    ctx->exact_only = on != 0;
    return RT_OK;
}

int rt_context_take_stats(rt_context* ctx, rt_stats* out) {
    if (!ctx || !out) return RT_ERR_INVALID_ARGUMENT;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
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
    std::vector<int64_t> fixed(elems);
    HIP_TRY(ctx, hipMemcpyAsync(fixed.data(), ctx->d_accum, elems * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const double inv = 1.0 / (double)(1ull << RT_ACCUM_FRAC_BITS);
    for (size_t k = 0; k < elems; ++k) accum_rgb_out[k] = (double)fixed[k] * inv;
    if (stats) {
        rc = rt_context_take_stats(ctx, stats);
        if (rc != RT_OK) return rc;
    }
    return RT_OK;
}

int rt_first_hit(rt_context* ctx, const rt_camera_desc* cam, int width, int height, int32_t* ids_out) {
    if (!ctx) return RT_ERR_INVALID_ARGUMENT;
    if (!cam || !ids_out) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "null camera or output");
    if (!ctx->have_scene) return fail(ctx, RT_ERR_NO_SCENE, "rt_scene_upload has not succeeded on this context");
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
    FirstHitArgs A{};
    A.scene = ctx->scene;
    fill_camera(A.cam, cam);
    A.width = width;
    A.height = height;
    A.exact_only = ctx->exact_only ? 1 : 0;
    A.ids = ctx->d_ids;
    HIP_TRY(ctx, launch_first_hit_spheres(A, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ids_out, ctx->d_ids, elems * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

}  // extern "C"
