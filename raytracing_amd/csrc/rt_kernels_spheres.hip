// gfx950 kernels for worlds that are a flat list of spheres (book 1: configs 1 and 2), and, as the TIMED variant,
// for worlds of book 2 whose only primitives are spheres and moving spheres, however they are nested in BVH nodes
// (config 3): a closest hit does not depend on the order the objects are tested in (B2/bvh.h:97-106 and
// B1/hittable_list.h:26-39 both keep the smallest t in [t_min, closest so far]), so the tree is walked as the list
// of its leaves.  The TIMED variant adds B2/camera.h:44 (shutter time), B2/moving_sphere.h:27-29 and 49-77 (centre at
// the ray's time) and B2/texture.h:37-56 (checker texture).
//
// Replaces, for every pixel and sample, the reference's render loop body
//   B1/RayTracing.cpp:119-124  jitter -> camera::get_ray -> ray_color
//   B1/RayTracing.cpp:11-31    ray_color (recursion unrolled into a per-lane loop)
//   B1/hittable_list.h:26-39   closest hit over all objects
//   B1/sphere.h:23-50          sphere::hit
//   B1/material.h:19-77        lambertian / metal / dielectric scatter
//
// Structure of one wave (64 lanes, one path per lane):
//   refill   lanes whose path ended take the next sample of their work item (a run of
//            `chunk` samples of one pixel) or a new item from a device-wide queue, and
//            build the camera ray.  Colour sums stay in registers as 64-bit fixed point
//            and are added to HBM once per item, so the result does not depend on which
//            lane or wave (or GPU) rendered which sample.
//   screen   ALL lanes walk ALL spheres together: the sphere image is read with
//            wave-uniform scalar loads (SGPR operands, no vector memory traffic) and each
//            lane evaluates a conservative float discriminant in 9 VALU ops per sphere,
//            shifting the sign bit into a 32-sphere mask word.
//   exact    each lane re-tests only its own candidates in double with the reference's
//            operation order, in list order and with the reference's closest-so-far
//            rule, so the winner and its t are those of the reference bit for bit.
//   shade    the reference's scatter functions in double; random numbers from Philox.
//
// The float screen can only add candidates, never drop a true hit (bound in screen_prepare),
// so the rendered numbers are those of an all-double renderer.
#define RT_PHILOX_OPAQUE_KEY 1  // see philox4x32_10
#define RT_ISSUE_BOUND 1        // see rt_device_common.h
#include "rt_device_common.h"
#include "rt_internal.h"
#include "rt_render_common.h"
#include "rt_sphere_screen.h"

namespace rtd {

// ---------------------------------------------------------------- render kernel
#ifndef RT_MIN_WAVES_PER_SIMD
#define RT_MIN_WAVES_PER_SIMD 4
#endif
// The book-2 variant also fits 4 waves/SIMD (6 spilled registers outside the loop; measured 3 % faster than at 3
// waves) now that the checker texture evaluates no sin() inline.
#ifndef RT_ARGS_FROM_MEMORY
#define RT_ARGS_FROM_MEMORY 1
#endif
#ifndef RT_TIMED_WAVE_BLOCKS
#define RT_TIMED_WAVE_BLOCKS 0
#endif
#ifndef RT_MIN_WAVES_TIMED
#define RT_MIN_WAVES_TIMED 4
#endif
#ifndef RT_MERGED_UNIT_VECTOR
#define RT_MERGED_UNIT_VECTOR 1
#endif
// texture::value of a lambertian's albedo for the textures a sphere world may use: solid_color and checker_texture
// (B2/texture.h:23-25, 46-52), checkers nested at most three deep.
__device__ __forceinline__ d3 sphere_world_texture(const SphereListDev& S, int ti, d3 p) {
    for (int level = 0; level < 4; ++level) {
        const TextureDev t = S.textures[ti];
        if (t.kind != RT_TEX_CHECKER) return d3{t.p[0], t.p[1], t.p[2]};
        ti = checker_sines_negative(p) ? t.b : t.a;
    }
    return d3{0, 0, 0};
}

template <bool TIMED>
__global__ void __launch_bounds__(RT_SPHERES_WG, TIMED ? RT_MIN_WAVES_TIMED : RT_MIN_WAVES_PER_SIMD) k_render_spheres(const RenderArgs A_in) {
    constexpr int WG_WAVES = RT_SPHERES_WG / 64;
    const RenderArgs& A = A_in;
    // (16-byte aligned, slices of whole 8-byte words: spread_exact_tests runs 64-bit LDS operations on them)
    __shared__ __align__(16) uint16_t s_list[WG_WAVES][CAND_CAP * 64];
    __shared__ __align__(16) uint32_t s_rows[WG_WAVES][ROW_WORDS * 64];
    const int lane = threadIdx.x & 63;
    uint16_t* list = s_list[threadIdx.x >> 6];
    uint32_t* rows = s_rows[threadIdx.x >> 6];
    // The camera (21 doubles) is needed once per sample; read from the kernel arguments it would sit in 42 scalar
    // registers for the whole kernel.  One copy per wave in LDS, read back where it is used.
    __shared__ CameraDev s_cam[WG_WAVES];
    {
        const double* src = (const double*)&A.cam;
        double* dst = (double*)&s_cam[threadIdx.x >> 6];
        if (lane < (int)(sizeof(CameraDev) / sizeof(double))) dst[lane] = src[lane];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    const CameraDev& cam = s_cam[threadIdx.x >> 6];
    // reciprocals of the jitter's two divisors (shared_divisor), one copy per wave like the camera
    __shared__ double s_frame_rcp[WG_WAVES][2];
    {
        double* rcp = s_frame_rcp[threadIdx.x >> 6];
        if (lane < 2) rcp[lane] = shared_divisor(lane == 0 ? A.width_m1 : A.height_m1).r;
        lds_wave_sync();
    }
    const double* frame_rcp = s_frame_rcp[threadIdx.x >> 6];
    const SphereListDev& S = A.scene;
    __shared__ AccumCache s_acc[WG_WAVES];
    constexpr bool WAVE_BLOCKS = (!TIMED || RT_TIMED_WAVE_BLOCKS != 0) && (RT_COOP_SHADE != 0);  // Philox blocks of the rejection loops computed wave-wide
    accum_cache_init(&s_acc[threadIdx.x >> 6], lane);

    // per-lane path state
    bool active = false;
    uint32_t bounce = 0;
    d3 o{0, 0, 0}, d{0, 0, 1}, T{1, 1, 1};
    double tm = 0.0;  // ray::time(), the same for every segment of a path (TIMED only)
    LaneWork w{};
    WaveWork wq{};
#if !RT_ARGS_FROM_MEMORY
    const ItemPlan plan = A.plan;
    const WorkQueues Q{A.item_counter, A.n_items, A.accum, &s_acc[threadIdx.x >> 6], plan.batch, plan.exclusive != 0};
#endif
    // statistics of the wave (counted with population counts of lane masks: scalar registers, not one per lane)
    uint32_t n_samples = 0, n_rays = 0;
    uint32_t n_bad = 0, n_overflow = 0;  // per lane: rare events counted where they happen
#ifdef RT_TAIL_PROBE
    const uint64_t pr_t0 = wall_clock64();
    uint64_t pr_tq = 0, pr_iters = 0, pr_iters_q = 0, pr_lanes_q = 0, pr_lanes = 0;
#endif

    ProbeClock pc;
    pc.start();
#if RT_ARGS_FROM_MEMORY
    // The arguments fill ~70 scalar registers; held across the loop they are spilled to lanes of a vector register and
    // every reload is a vector instruction.  Read through a pointer the compiler cannot see through, each field is a
    // scalar load where the trip uses it and nothing but the pointer lives from one trip to the next.
    typedef __attribute__((address_space(4))) const RenderArgs KernArgs;
    KernArgs* args_in_memory = (KernArgs*)__builtin_amdgcn_kernarg_segment_ptr();
#endif
    for (;;) {
#if RT_ARGS_FROM_MEMORY
        asm volatile("" : "+s"(args_in_memory));
        const RenderArgs& A = *(const RenderArgs*)args_in_memory;  // (the loads stay scalar: the address space is inferred)
        const SphereListDev& S = A.scene;
        // (the item plan and the queue's description likewise: read where refill_work uses them)
        const ItemPlan& plan = A.plan;
        const WorkQueues Q{A.item_counter, A.n_items, A.accum, &s_acc[threadIdx.x >> 6], plan.batch, plan.exclusive != 0};
#endif
        // ---- refill: finish items, take new ones, start the next sample
        if (!refill_work(Q, plan, A.sample_begin, lane, active, w, wq)) break;
        {   // the lanes that start a sample
            const bool start = !active && w.has_item;
            if constexpr (WAVE_BLOCKS) {
                const uint32_t row = div_by_width(w.pixel, A.wdiv.mul, A.wdiv.shift);
                const int i = (int)(w.pixel - row * (uint32_t)A.width);
                const int j = A.height - 1 - (int)row;  // row 0 = top = j = H-1
                camera_ray_for_lanes<TIMED, RT_COOP_CAMERA != 0>(start, cam, A.width_m1, A.height_m1, i, j, w.pixel, (uint32_t)w.s_cur, A.key0,
                                                                 A.key1, lane, nullptr, o, d, tm, frame_rcp);
            } else if (start) {  // (the book-2 variant is at its register limit: every lane computes its own blocks, in the branch)
                const uint32_t row = div_by_width(w.pixel, A.wdiv.mul, A.wdiv.shift);
                const int i = (int)(w.pixel - row * (uint32_t)A.width);
                const int j = A.height - 1 - (int)row;
                camera_ray_for_lanes<TIMED, false>(true, cam, A.width_m1, A.height_m1, i, j, w.pixel, (uint32_t)w.s_cur, A.key0, A.key1, lane,
                                                   nullptr, o, d, tm);
            }
            n_samples += (uint32_t)__popcll(ballot(start));
            if (start) {
                T = d3{1, 1, 1};
                bounce = 0;
                active = true;
            }
        }
        // ---- closest hit (all lanes together)
        const bool on = active;
        pc.mark(SEC_REFILL);
#ifdef RT_TAIL_PROBE
        {
            const uint64_t na = (uint64_t)__popcll(ballot(on));
            ++pr_iters;
            pr_lanes += na;
            if (wq.dry) {
                if (pr_tq == 0) pr_tq = wall_clock64();
                ++pr_iters_q;
                pr_lanes_q += na;
            }
        }
#endif
        const SetHit h = closest_hit<TIMED>(S.set, o, d, tm, __builtin_inf(), on, A.exact_only != 0, list, rows, lane, n_overflow, pc);

        // ---- shade
        const bool hit = on && h.id >= 0;
        const uint32_t site = 1u + bounce;
        // Every material's first draw is block 0 of the segment's site.  Book-1 kernel: the block and the unit-sphere
        // rejection loop of lambertian and metal run at wave level, the loop's further blocks computed by the whole wave
        // for the lanes still in it (only two words per lane stay live across it).
        u32x4 block0{0, 0, 0, 0};
        d3 rs{0, 0, 0};
        if constexpr (WAVE_BLOCKS) {
            int mat_kind = RT_MAT_DIELECTRIC;
            if (hit) mat_kind = S.shade_of[h.id].material_kind;
            const bool wants_sphere = hit && (mat_kind == RT_MAT_LAMBERTIAN || mat_kind == RT_MAT_METAL);
            block0 = philox4x32_10(w.pixel, (uint32_t)w.s_cur, site, 0u, A.key0, A.key1);
            rs = random_in_unit_sphere_for_lanes(wants_sphere, block0, w.pixel, (uint32_t)w.s_cur, site, A.key0, A.key1, lane, nullptr);
        }
#ifdef RT_PROBE_SPLIT
        pc.mark(SEC_SCREEN_SPARSE);
#endif
        n_rays += (uint32_t)__popcll(ballot(on));
        if (on) {
            bool done = false;
            d3 L_end{0, 0, 0};
            SphereExact e{0, 0, 0, 1};
            MaterialDev m{RT_MAT_LAMBERTIAN, -1, 0, 0, 0, 0};
            double inv_radius = 1.0;
            if (hit) {
                e = S.set.exact[h.id];
                inv_radius = S.shade_of[h.id].inv_radius;
                m = S.materials[h.id];
            }
            // unit_vector(r.direction()) is wanted by the sky, by metal and by dielectric: one copy of the square
            // root and division for all of them instead of one per divergent branch
            // -- and the lambertian's unit_vector(random_in_unit_sphere()) is the same function of another vector, wanted
            // by exactly the lanes that do not want the direction's: one evaluation on each lane's own argument.
            // (Where the sampler's vector is known by now: the variant that draws it inside the material's branch keeps two.)
            constexpr bool MERGED = WAVE_BLOCKS && (RT_MERGED_UNIT_VECTOR != 0);
            d3 unit_direction{0, 0, 0};
            constexpr bool SHORT = !TIMED || RT_TIMED_SHARED_DIVISOR != 0;  // roots and quotients without rescaling (rt_device_common.h)
            if constexpr (MERGED) unit_direction = unit_vector_shared((hit && m.kind == RT_MAT_LAMBERTIAN) ? rs : d);
            else if (!hit || m.kind != RT_MAT_LAMBERTIAN) unit_direction = SHORT ? unit_vector_shared(d) : unit_vector(d);
            if (!hit) {
                if (S.sky_mode == RT_SKY_GRADIENT) {  // B1/RayTracing.cpp:28-30
                    const double t = 0.5 * (unit_direction.y + 1.0);
                    L_end = (1.0 - t) * d3{1.0, 1.0, 1.0} + t * d3{0.5, 0.7, 1.0};
                } else {
                    L_end = d3{S.background[0], S.background[1], S.background[2]};
                }
                done = true;
            } else {
                // rest of sphere::hit, B1/sphere.h:43-47
                const d3 p = o + h.t * d;
                const d3 outward_normal = inv_radius * (p - center_at<TIMED>(S.set, h.id, e, tm));  // (p - center) / radius
                const bool front_face = dot(d, outward_normal) < 0;
                const d3 normal = front_face ? outward_normal : -outward_normal;
                d3 attenuation{m.ax, m.ay, m.az};
                if constexpr (TIMED) {  // m.tex >= 0 only for a lambertian whose albedo is a checker texture
                    if (m.tex >= 0) attenuation = sphere_world_texture(S, m.tex, p);
                }
                d3 nd{0, 0, 0};
                if constexpr (!WAVE_BLOCKS) {
                    block0 = philox4x32_10(w.pixel, (uint32_t)w.s_cur, site, 0u, A.key0, A.key1);
                    rs = random_in_unit_sphere(m.kind == RT_MAT_LAMBERTIAN || m.kind == RT_MAT_METAL, block0, w.pixel, (uint32_t)w.s_cur, site,
                                               A.key0, A.key1);
                }
                if (m.kind == RT_MAT_LAMBERTIAN) {  // B1/material.h:19-30
                    nd = normal + (MERGED ? unit_direction : (SHORT ? unit_vector_shared(rs) : unit_vector(rs)));  // (MERGED: a lambertian lane's holds unit_vector(rs))
                    const double s = 1e-8;
                    if ((fabs(nd.x) < s) && (fabs(nd.y) < s) && (fabs(nd.z) < s)) nd = normal;
                } else if (m.kind == RT_MAT_METAL) {  // B1/material.h:40-47
                    const d3 reflected = reflect(unit_direction, normal);
                    nd = reflected + m.param * rs;
                    if (!(dot(nd, normal) > 0)) done = true;  // absorbed: black
                } else {  // dielectric, B1/material.h:58-77
                    attenuation = d3{1.0, 1.0, 1.0};
                    const double refraction_ratio = front_face ? m.ax : m.param;  // m.ax = 1.0 / ir (device_materials)
                    const double cos_theta = fmin(dot(-unit_direction, normal), 1.0);
                    const double sin_theta = SHORT ? sqrt_shared(1.0 - cos_theta * cos_theta) : sqrt(1.0 - cos_theta * cos_theta);
                    bool do_reflect = refraction_ratio * sin_theta > 1.0;
                    if (!do_reflect) do_reflect = reflectance_from_r0(cos_theta, front_face ? m.ay : m.az) > u01(block0.x);  // r0: device_materials
                    nd = do_reflect ? reflect(unit_direction, normal) : refract(unit_direction, normal, refraction_ratio);
                }
                if (!done) {
                    T = T * attenuation;
                    o = p;
                    d = nd;
                    ++bounce;
                    if ((int)bounce >= A.max_depth) done = true;  // B1/RayTracing.cpp:15: the next call (depth 0) returns black
                }
            }
            if (done) {
                const d3 c = T * L_end;
                add_sample(c, w.fx, w.fy, w.fz, n_bad);
                active = false;
                ++w.s_cur;
            }
        }
        pc.mark(SEC_SHADE);
    }

    {
        const WorkQueues Q{A_in.item_counter, A_in.n_items, A_in.accum, &s_acc[threadIdx.x >> 6], A_in.plan.batch, A_in.plan.exclusive != 0};
        accum_cache_flush(Q, lane);
    }
#ifdef RT_TAIL_PROBE
    if (lane == 0 && A.probe) {
        uint64_t* w = A.probe + 16ull * (blockIdx.x * (uint32_t)WG_WAVES + (threadIdx.x >> 6));
        for (int k = 0; k < 8; ++k) w[8 + k] = pc.sec[k];
        w[0] = pr_t0;
        w[1] = pr_tq;
        w[2] = wall_clock64();
        w[3] = pr_iters;
        w[4] = pr_iters_q;
        w[5] = pr_lanes;
        w[6] = pr_lanes_q;
        w[7] = 1;
    }
#endif
    // ---- statistics: one atomic per counter per wave
    uint64_t v0 = n_samples, v1 = n_rays, v2 = n_bad, v3 = n_overflow;  // v0, v1: already the wave's totals
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        v2 += __shfl_xor(v2, off);
        v3 += __shfl_xor(v3, off);
    }
    if (lane == 0) {
        atomicAdd((unsigned long long*)&A.stats[0], (unsigned long long)v0);
        atomicAdd((unsigned long long*)&A.stats[1], (unsigned long long)v1);
        atomicAdd((unsigned long long*)&A.stats[2], (unsigned long long)v2);
        atomicAdd((unsigned long long*)&A.stats[3], (unsigned long long)v3);
    }
}

// ---------------------------------------------------------------- first-hit kernel
// One lane per pixel centre; definition in raytracing_amd.h (rt_first_hit).
template <bool TIMED>
__global__ void __launch_bounds__(256) k_first_hit_spheres(const FirstHitArgs A) {
    __shared__ __align__(16) uint16_t s_list[4][CAND_CAP * 64];
    __shared__ __align__(16) uint32_t s_rows[4][ROW_WORDS * 64];
    const int lane = threadIdx.x & 63;
    uint16_t* list = s_list[threadIdx.x >> 6];
    uint32_t* rows = s_rows[threadIdx.x >> 6];
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
    const d3 o = origin + offset;
    const d3 d = llc + s * hor + t * ver - origin - offset;
    uint32_t ovf = 0;
    ProbeClock clk;
    const SetHit h = closest_hit<TIMED>(A.scene.set, o, d, C.time0, __builtin_inf(), on, A.exact_only != 0, list, rows, lane, ovf, clk);
    if (on) A.ids[pixel] = (TIMED && h.id >= 0) ? A.scene.top_of[h.id] : h.id;
}

hipError_t launch_render_spheres(const RenderArgs& args, bool timed, int blocks, hipStream_t stream) {
    if (timed) hipLaunchKernelGGL(k_render_spheres<true>, dim3((unsigned)blocks), dim3(RT_SPHERES_WG), 0, stream, args);
    else hipLaunchKernelGGL(k_render_spheres<false>, dim3((unsigned)blocks), dim3(RT_SPHERES_WG), 0, stream, args);
    return hipGetLastError();
}

hipError_t launch_first_hit_spheres(const FirstHitArgs& args, bool timed, hipStream_t stream) {
    const uint32_t npix = (uint32_t)args.width * (uint32_t)args.height;
    if (timed) hipLaunchKernelGGL(k_first_hit_spheres<true>, dim3((npix + 255u) / 256u), dim3(256), 0, stream, args);
    else hipLaunchKernelGGL(k_first_hit_spheres<false>, dim3((npix + 255u) / 256u), dim3(256), 0, stream, args);
    return hipGetLastError();
}

int render_spheres_blocks_per_cu(bool timed) {
    int n = 0;
    const hipError_t e = timed ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_render_spheres<true>, RT_SPHERES_WG, 0)
                               : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_render_spheres<false>, RT_SPHERES_WG, 0);
    if (e != hipSuccess || n < 1) n = 2;
    return n;
}

}  // namespace rtd
