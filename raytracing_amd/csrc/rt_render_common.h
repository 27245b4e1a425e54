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
        if (k & 1u) {  // a new block every second attempt
            w = philox4x32_10(pixel, sample, 0u, (k + 1u) >> 1, k0, k1);
            px = u_range(w.x, -1, 1);
            py = u_range(w.y, -1, 1);
        } else {
            px = u_range(w.z, -1, 1);
            py = u_range(w.w, -1, 1);
        }
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
    uint32_t sx = 0, sy = 0, sz = 0;  // words 3 of the last three blocks: the fourth attempt of each group
    for (uint32_t k = 0; need; ++k) {
        uint32_t x, y, z;
        if ((k & 3u) == 3u) {
            x = sx;
            y = sy;
            z = sz;
        } else {
            const u32x4 w = philox4x32_10(pixel, sample, site, 3u * (k >> 2) + (k & 3u), k0, k1);
            x = w.x;
            y = w.y;
            z = w.z;
            sx = (k & 3u) == 0u ? w.w : sx;
            sy = (k & 3u) == 1u ? w.w : sy;
            sz = (k & 3u) == 2u ? w.w : sz;
        }
        p = d3{u_range(x, -1, 1), u_range(y, -1, 1), u_range(z, -1, 1)};
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

// ---------------------------------------------------------------- work source of a persistent wave
//
// Items of the plan above are handed out in batches of 64 from one device-wide atomic counter.  Once the
// queue is dry, a lane that still holds unstarted samples of its item gives them, one at a time, to
// idle lanes of its own wave (register exchange, no memory traffic): pixels differ tenfold in cost, and
// without this the end of a launch waits for one lane to walk through the rest of a heavy item.
// Polling a device-wide give-back queue instead was tried and is far slower: thousands of waves
// reading one counter every trip serialise on that address.
struct WorkQueues {
    uint32_t* counter;  // head of the item queue, zeroed before launch
    uint32_t n_items;
    int64_t* accum;
};
struct LaneWork {
    bool has_item;
    uint32_t pixel;
    int32_t s_cur, s_end;  // next sample to start / end of the run
    int64_t fx, fy, fz;    // fixed-point sums of the finished samples of the item
};
struct WaveWork {
    uint32_t batch_next, batch_end;
    bool dry;  // queue exhausted (wave-uniform)
};

// One refill step for the wave; `active` = lane has a path in flight.  Returns false when the wave is done.
// Afterwards a lane with has_item && !active must start sample s_cur.
__device__ __forceinline__ bool refill_work(const WorkQueues& Q, const ItemPlan& plan, uint32_t npix, int32_t sample_begin,
                                            int lane, bool active, LaneWork& w, WaveWork& q) {
    if (!active && w.has_item && w.s_cur == w.s_end) {
        atomic_add_i64(Q.accum + 3ull * w.pixel + 0, w.fx);
        atomic_add_i64(Q.accum + 3ull * w.pixel + 1, w.fy);
        atomic_add_i64(Q.accum + 3ull * w.pixel + 2, w.fz);
        w.has_item = false;
    }
    bool need = !active && !w.has_item;
    while (!q.dry && ballot(need) != 0ull) {
        if (q.batch_next == q.batch_end) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(Q.counter, 64u);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (base >= Q.n_items) {
                q.dry = true;
                break;
            }
            q.batch_next = base;
            q.batch_end = min(base + 64u, Q.n_items);
        }
        const uint64_t m = ballot(need);
        const uint32_t rank = rank_below(m);
        const uint32_t avail = q.batch_end - q.batch_next;
        if (need && rank < avail) {
            const uint32_t item = q.batch_next + rank;
            w.pixel = item % npix;
            int first, count;
            item_run(plan, (int)(item / npix), first, count);
            w.s_cur = sample_begin + first;
            w.s_end = w.s_cur + count;
            w.fx = w.fy = w.fz = 0;
            w.has_item = true;
            need = false;
        }
        q.batch_next += min((uint32_t)__popcll(m), avail);
    }
    if (!q.dry) return true;

    // ---- queue dry: hand unstarted samples (from the end of a run) to idle lanes of this wave
    for (;;) {
        const bool can_give = w.has_item && w.s_end - w.s_cur > 1;
        const uint64_t givers = ballot(can_give), takers = ballot(need);
        if (givers == 0ull || takers == 0ull) break;
        const int g = __builtin_ctzll(givers), t = __builtin_ctzll(takers);
        if (lane == g) --w.s_end;
        const uint32_t px = (uint32_t)__builtin_amdgcn_readlane((int)w.pixel, g);
        const int32_t smp = __builtin_amdgcn_readlane(w.s_end, g);
        if (lane == t) {
            w.pixel = px;
            w.s_cur = smp;
            w.s_end = smp + 1;
            w.fx = w.fy = w.fz = 0;
            w.has_item = true;
            need = false;
        }
    }
    return ballot(active || w.has_item) != 0ull;
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
