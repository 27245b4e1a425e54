// Device functions shared by the render kernels: camera ray, rejection samplers, Schlick term,
// fixed-point accumulation.  Double arithmetic in the reference's operation order.
#pragma once

#include "rt_device_common.h"
#include "rt_internal.h"

namespace rtd {

// ---------------------------------------------------------------- camera
// camera::get_ray (B1/camera.h:34-39) for pixel column i, reference row j, sample `sample`.
__device__ __forceinline__ void camera_ray(const CameraDev& C, int W, int H, int i, int j, uint32_t pixel,
                                           uint32_t sample, uint32_t k0, uint32_t k1, d3& o, d3& d) {
    u32x4 w = philox4x32_10(pixel, sample, 0u, 0u, k0, k1);
    const double s = (i + u01(w.x)) / (W - 1);  // B1/RayTracing.cpp:120-121
    const double t = (j + u01(w.y)) / (H - 1);
    double px = u_range(w.z, -1, 1), py = u_range(w.w, -1, 1);
    for (uint32_t k = 1; px * px + py * py + 0.0 * 0.0 >= 1; ++k) {  // random_in_unit_disk
        w = philox4x32_10(pixel, sample, 0u, k, k0, k1);
        px = u_range(w.x, -1, 1);
        py = u_range(w.y, -1, 1);
    }
    const double rdx = C.lens_radius * px, rdy = C.lens_radius * py;
    const d3 cu{C.u[0], C.u[1], C.u[2]}, cv{C.v[0], C.v[1], C.v[2]};
    const d3 offset = rdx * cu + rdy * cv;
    const d3 origin{C.origin[0], C.origin[1], C.origin[2]};
    const d3 llc{C.llc[0], C.llc[1], C.llc[2]};
    const d3 hor{C.horizontal[0], C.horizontal[1], C.horizontal[2]};
    const d3 ver{C.vertical[0], C.vertical[1], C.vertical[2]};
    o = origin + offset;
    d = llc + s * hor + t * ver - origin - offset;
}

// random_in_unit_sphere, B1/vec3.h:114-120, for the lanes with need == true.
__device__ __forceinline__ d3 random_in_unit_sphere(bool need, uint32_t pixel, uint32_t sample, uint32_t site,
                                                    uint32_t k0, uint32_t k1) {
    d3 p{0, 0, 0};
    for (uint32_t k = 0; need; ++k) {
        const u32x4 w = philox4x32_10(pixel, sample, site, k, k0, k1);
        p = d3{u_range(w.x, -1, 1), u_range(w.y, -1, 1), u_range(w.z, -1, 1)};
        if (!(len2(p) >= 1)) need = false;
    }
    return p;
}

__device__ __forceinline__ double reflectance(double cosine, double ref_idx) {  // B1/material.h:83-88
    double r0 = (1 - ref_idx) / (1 + ref_idx);
    r0 = r0 * r0;
    const double x = 1 - cosine;
    const double x2 = x * x;
    return r0 + (1 - r0) * (x2 * x2 * x);  // pow(x, 5) by multiplication (<= 2 ulp from libm pow)
}

__device__ __forceinline__ void atomic_add_i64(int64_t* p, int64_t v) {
    __hip_atomic_fetch_add((unsigned long long*)p, (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}


// Work items.  The samples [0, N) of a pixel are cut into `levels` runs: K runs of `cap` samples, then the
// remaining R = N - K*cap (cap <= R < 2*cap when N >= cap) halved repeatedly: R - R/2, R/2 - R/4, ..., 1.
// Item m is run (m / npix) of pixel (m % npix), so the queue hands out every pixel's long runs first and
// one-sample runs last: few atomics per pixel, and a fine-grained tail at the end of the launch.
struct ItemPlan {
    int32_t cap, K, R, levels;
};
__host__ __device__ inline ItemPlan make_item_plan(int32_t n_samples, int32_t cap) {
    ItemPlan p;
    p.cap = cap < 1 ? 1 : cap;
    p.K = n_samples / p.cap - 1;
    if (p.K < 0) p.K = 0;
    p.R = n_samples - p.K * p.cap;
    int bits = 0;
    for (int32_t r = p.R; r > 0; r >>= 1) ++bits;
    p.levels = p.K + bits;
    return p;
}
// samples [first, first + count) of run `level`, relative to the first sample of the call
__device__ __forceinline__ void item_run(const ItemPlan& p, int level, int& first, int& count) {
    if (level < p.K) {
        first = level * p.cap;
        count = p.cap;
    } else {
        const int g = level - p.K;
        first = p.K * p.cap + p.R - (p.R >> g);
        count = (p.R >> g) - (p.R >> (g + 1));
    }
}

// Converts one sample colour to 2^-RT_ACCUM_FRAC_BITS fixed point and adds it; a non-finite colour is
// dropped and counted (the reference would carry the NaN into its image).
__device__ __forceinline__ void add_sample(d3 c, int64_t& fx, int64_t& fy, int64_t& fz, uint32_t& n_bad) {
    if (isfinite(c.x) && isfinite(c.y) && isfinite(c.z)) {
        fx += __double2ll_rn(c.x * ACCUM_SCALE);
        fy += __double2ll_rn(c.y * ACCUM_SCALE);
        fz += __double2ll_rn(c.z * ACCUM_SCALE);
    } else {
        ++n_bad;
    }
}

}  // namespace rtd
