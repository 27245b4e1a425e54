// Device functions shared by the render kernels: camera ray, rejection samplers, Schlick term,
// fixed-point accumulation.  Double arithmetic in the reference's operation order.
#pragma once

#include "rt_device_common.h"
#include "rt_internal.h"

namespace rtd {

// ---------------------------------------------------------------- camera
// camera::get_ray (B1/camera.h:34-39) for pixel column i, reference row j, sample `sample`.
// wm1, hm1 = (double)(W - 1), (double)(H - 1).
__device__ __forceinline__ void camera_ray(const CameraDev& C, double wm1, double hm1, int i, int j, uint32_t pixel,
                                           uint32_t sample, uint32_t k0, uint32_t k1, d3& o, d3& d) {
    u32x4 w = philox4x32_10(pixel, sample, 0u, 0u, k0, k1);
    const double s = (i + u01(w.x)) / wm1;  // B1/RayTracing.cpp:120-121
    const double t = (j + u01(w.y)) / hm1;
    double px = u_signed(w.z), py = u_signed(w.w);
    for (uint32_t k = 1; px * px + py * py + 0.0 * 0.0 >= 1; ++k) {  // random_in_unit_disk
        if (k & 1u) {  // a new block every second attempt
            w = philox4x32_10(pixel, sample, 0u, (k + 1u) >> 1, k0, k1);
            px = u_signed(w.x);
            py = u_signed(w.y);
        } else {
            px = u_signed(w.z);
            py = u_signed(w.w);
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

// Whether the point p = (u_signed(x), u_signed(y), u_signed(z)) of three Philox words passes the test
// of the unit-sphere rejection loop, !(p.length_squared() >= 1) (B1/vec3.h:117), without the doubles.  A coordinate is
// exactly m 2^-30 with m = (word >> 1) - 2^30, so |p|^2 = S 2^-60 with the integer S = mx^2 + my^2 + mz^2 < 2^62.  The
// reference adds the three rounded squares in double: near 1 its sum is within 2^-51 of S 2^-60 (three roundings of
// values below 1, two of sums below 2), so S decides unless it lies within 2^10 of 2^60 -- then the doubles are evaluated.
// Eight vector instructions instead of twenty per attempt; the accepted point is converted once, after the loop.
// (Sphere-list kernels, bound by vector issue; the scene-graph kernels wait on loads and are short of registers: the
// final scene's variant loses 8 % to it.)
#ifndef RT_INTEGER_SPHERE_TEST
#define RT_INTEGER_SPHERE_TEST RT_ISSUE_BOUND
#endif
__device__ __forceinline__ bool inside_unit_sphere(uint32_t x, uint32_t y, uint32_t z) {
#if RT_INTEGER_SPHERE_TEST
    const int32_t mx = (int32_t)(x >> 1) - (1 << 30), my = (int32_t)(y >> 1) - (1 << 30), mz = (int32_t)(z >> 1) - (1 << 30);
    const uint64_t S = (uint64_t)((int64_t)mx * mx) + (uint64_t)((int64_t)my * my) + (uint64_t)((int64_t)mz * mz);
    if (S < (1ull << 60) - 1024ull) return true;
    if (S > (1ull << 60) + 1024ull) return false;
#endif
    const d3 p{u_signed(x), u_signed(y), u_signed(z)};
    return !(len2(p) >= 1);
}

// One lane's state across the attempts of the unit-sphere rejection loop: with the integer test the words of the attempt
// that passed, converted once at the end; otherwise the point itself, as the reference computes it attempt by attempt.
struct SpherePoint {
    bool wanted = false;
#if RT_INTEGER_SPHERE_TEST
    uint32_t kx = 0, ky = 0, kz = 0;
    __device__ __forceinline__ bool attempt(uint32_t x, uint32_t y, uint32_t z) {
        if (!inside_unit_sphere(x, y, z)) return false;
        kx = x;
        ky = y;
        kz = z;
        return true;
    }
    __device__ __forceinline__ d3 value() const {
        return wanted ? d3{u_signed(kx), u_signed(ky), u_signed(kz)} : d3{0, 0, 0};
    }
#else
    d3 p{0, 0, 0};
    __device__ __forceinline__ bool attempt(uint32_t x, uint32_t y, uint32_t z) {
        p = d3{u_signed(x), u_signed(y), u_signed(z)};
        return !(len2(p) >= 1);
    }
    __device__ __forceinline__ d3 value() const { return p; }
#endif
};

// random_in_unit_sphere, B1/vec3.h:114-120, for the lanes with need == true.  block0 = Philox block 0 of the site,
// computed by the caller (the dielectric's reflectance draw of the same site reads the same block: one evaluation
// serves the lanes of either kind).
__device__ __forceinline__ d3 random_in_unit_sphere(bool need, u32x4 block0, uint32_t pixel, uint32_t sample, uint32_t site,
                                                    uint32_t k0, uint32_t k1) {
    SpherePoint p;
    p.wanted = need;
    uint32_t sx = 0, sy = 0, sz = 0;  // words 3 of the last three blocks: the fourth attempt of each group
    for (uint32_t k = 0; need; ++k) {
        uint32_t x, y, z;
        if ((k & 3u) == 3u) {
            x = sx;
            y = sy;
            z = sz;
        } else {
            const u32x4 w = k == 0u ? block0 : philox4x32_10(pixel, sample, site, 3u * (k >> 2) + (k & 3u), k0, k1);
            x = w.x;
            y = w.y;
            z = w.z;
            sx = (k & 3u) == 0u ? w.w : sx;
            sy = (k & 3u) == 1u ? w.w : sy;
            sz = (k & 3u) == 2u ? w.w : sz;
        }
        if (p.attempt(x, y, z)) need = false;
    }
    return p.value();
}

// ---------------------------------------------------------------- Philox blocks computed by the wave for its lanes
//
// A rejection loop runs as long as its unluckiest lane, and in every round the whole wave issues the ~60 vector
// instructions of a Philox block for the few lanes still in the loop.  Here the lanes that want blocks are served by
// ALL lanes instead: NB lanes per wanting lane compute its next NB blocks side by side, in one evaluation, and the
// words travel back by lane permutation.  (The rounds that a loop of one block per round would take are mostly known
// to be needed: after a failed first attempt of the unit-sphere sampler, 48 % of the lanes need the second block too.)
// Same blocks, same words, same draws as computing them one by one: results do not change.
//
// want: this lane wants blocks first .. first+NB-1 of counter (c0, c1, c2, .); alt_last != 0 replaces the index of the
// last one.  All 64 lanes must call it.  When more than 63/NB lanes want blocks, every lane computes its own (NB
// evaluations).
__device__ __forceinline__ uint32_t lane_pull(int src_lane, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)v);
}
__device__ __forceinline__ uint32_t lane_push(int dst_lane, uint32_t v) {  // what was pushed to this lane (0 if nothing)
    return (uint32_t)__builtin_amdgcn_ds_permute(dst_lane << 2, (int)v);
}
template <int NB>
__device__ __forceinline__ void philox_blocks_for_lanes(bool want, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t first,
                                                        uint32_t alt_last, uint32_t k0, uint32_t k1, int lane, uint8_t* /*owners*/,
                                                        u32x4 (&out)[NB]) {
    const uint64_t m = ballot(want);
    const int n = __popcll(m);
    if (n * NB <= 63) {
        // The wanting lane of rank r pushes its counter to lanes NB r .. NB r + NB - 1 (one forward permutation per word
        // and destination; the other lanes push to lane 63, which serves nobody), those lanes compute one block each, and
        // the owner pulls the words back: two permutation latencies around one Philox evaluation.
        const int base = (int)rank_below(m) * NB;
        uint32_t b0 = 0, b1 = 0, b2 = 0, b3 = 0;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int dst = want ? base + j : 63;
            const uint32_t idx = (j == NB - 1 && alt_last != 0u) ? alt_last : first + (uint32_t)j;
            const uint32_t r0 = lane_push(dst, c0), r1 = lane_push(dst, c1), r2 = lane_push(dst, c2), r3 = lane_push(dst, idx);
            const bool mine = lane - (lane / NB) * NB == j;  // this lane is destination j of its slot
            b0 = mine ? r0 : b0;
            b1 = mine ? r1 : b1;
            b2 = mine ? r2 : b2;
            b3 = mine ? r3 : b3;
        }
        const u32x4 r = philox4x32_10(b0, b1, b2, b3, k0, k1);
#pragma unroll
        for (int j = 0; j < NB; ++j)
            out[j] = u32x4{lane_pull(base + j, r.x), lane_pull(base + j, r.y), lane_pull(base + j, r.z), lane_pull(base + j, r.w)};
    } else {
#pragma unroll
        for (int j = 0; j < NB; ++j)
            out[j] = philox4x32_10(c0, c1, c2, (j == NB - 1 && alt_last != 0u) ? alt_last : first + (uint32_t)j, k0, k1);
    }
}

// camera::get_ray (B1/camera.h:34-39, B2/camera.h:38-45) for the lanes with `start` set, the Philox blocks coming from
// philox_blocks_for_lanes: blocks 0 and 1 (jitter, lens attempts 0 to 2) in one evaluation -- blocks 0 and TRY_TIME when
// the camera has a shutter interval -- and two more per further evaluation for the lanes still rejecting lens points.
// Same draws as camera_ray.  All 64 lanes must call it.
#ifndef RT_COOP_CAMERA
#define RT_COOP_CAMERA 1
#endif
#ifndef RT_COOP_SHADE
#define RT_COOP_SHADE 1
#endif
template <bool TIMED, bool COOP = (RT_COOP_CAMERA != 0)>
__device__ __forceinline__ void camera_ray_for_lanes(bool start, const CameraDev& C, double wm1, double hm1, int i, int j, uint32_t pixel,
                                                     uint32_t sample, uint32_t k0, uint32_t k1, int lane, uint8_t* owners, d3& o,
                                                     d3& d, double& tm, const double* frame_rcp = nullptr) {
    if constexpr (!COOP) {
        if (start) {
            camera_ray(C, wm1, hm1, i, j, pixel, sample, k0, k1, o, d);
            if constexpr (TIMED) {
                tm = C.time0;
                if (C.time0 != C.time1) tm = u_range(philox4x32_10(pixel, sample, 0u, TRY_TIME, k0, k1).x, C.time0, C.time1);
            }
        }
        return;
    }
    u32x4 b[2];
    philox_blocks_for_lanes<2>(start, pixel, sample, 0u, 0u, TIMED ? TRY_TIME : 0u, k0, k1, lane, owners, b);
    // B1/RayTracing.cpp:120-121.  frame_rcp = the reciprocals of shared_divisor(wm1) and shared_divisor(hm1), worked out
    // once per wave: the numerators are 0 or within [2^-31, 2^28], the denominators within [1, 2^28].
    double s, t;
    if (frame_rcp != nullptr && RT_SHARED_DIVISOR != 0) {
        s = quotient_of_tame(i + u01(b[0].x), wm1, frame_rcp[0]);
        t = quotient_of_tame(j + u01(b[0].y), hm1, frame_rcp[1]);
    } else {
        s = (i + u01(b[0].x)) / wm1;
        t = (j + u01(b[0].y)) / hm1;
    }
    double px = u_signed(b[0].z), py = u_signed(b[0].w);
    bool need = start && px * px + py * py + 0.0 * 0.0 >= 1;  // random_in_unit_disk
    auto attempt = [&](uint32_t x, uint32_t y) {
        if (need) {
            px = u_signed(x);
            py = u_signed(y);
            need = px * px + py * py + 0.0 * 0.0 >= 1;
        }
    };
    uint32_t next = 2u;  // next block of lens attempts to fetch
    if constexpr (TIMED) {
        if (start) tm = C.time0 != C.time1 ? u_range(b[1].x, C.time0, C.time1) : C.time0;  // B2/camera.h:44
        next = 1u;
    } else {
        attempt(b[1].x, b[1].y);
        attempt(b[1].z, b[1].w);
    }
    while (ballot(need) != 0ull) {
        philox_blocks_for_lanes<2>(need, pixel, sample, 0u, next, 0u, k0, k1, lane, owners, b);
        attempt(b[0].x, b[0].y);
        attempt(b[0].z, b[0].w);
        attempt(b[1].x, b[1].y);
        attempt(b[1].z, b[1].w);
        next += 2u;
    }
    if (start) {
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
}

// The later attempts of the unit-sphere rejection loop, when at most four lanes of the wave are still in it (after four
// attempts 5 % of the lanes that wanted a point are: two or three of a wave).  Each of them is served by sixteen lanes
// that compute the blocks AND test the attempts, laid out in the order of the attempts: lanes 4i, 4i+1, 4i+2 of the
// sixteen hold blocks 3(g+i), 3(g+i)+1, 3(g+i)+2 and test their words 0, 1, 2 (attempts 4(g+i) .. 4(g+i)+2), lane 4i+3
// gathers the three words 3 of its neighbours (attempt 4(g+i)+3).  One ballot then holds, per owner, sixteen attempts in
// order: its lowest set bit names the lane whose words the reference's loop would have accepted.  One Philox evaluation
// and one test for sixteen attempts, where the owner's own loop tests four per evaluation one after the other (and the
// chance of needing another round falls from 5 % to 10^-5).  Same blocks, same words, same order: same point.
// Returns false, with nothing done, when more than four lanes need attempts.  All 64 lanes must call it.
#ifndef RT_SPHERE_ATTEMPTS_BY_HELPERS
#define RT_SPHERE_ATTEMPTS_BY_HELPERS 1
#endif
#if RT_INTEGER_SPHERE_TEST
__device__ __forceinline__ bool sphere_attempts_by_helpers(bool& need, SpherePoint& p, uint32_t pixel, uint32_t sample, uint32_t site,
                                                           uint32_t g, uint32_t k0, uint32_t k1, int lane) {
    const uint64_t m = ballot(need);
    const int n = __popcll(m);
    if (n > 4) return false;
    // the owners' lanes, in rank order (wave-uniform: scalar arithmetic); ranks without an owner name the first one
    const int l0 = __builtin_ctzll(m);
    const uint64_t m1 = m & (m - 1ull);
    const int l1 = m1 ? __builtin_ctzll(m1) : l0;
    const uint64_t m2 = m1 & (m1 - 1ull);
    const int l2 = m2 ? __builtin_ctzll(m2) : l0;
    const uint64_t m3 = m2 & (m2 - 1ull);
    const int l3 = m3 ? __builtin_ctzll(m3) : l0;
    const int q = lane >> 4;  // the rank this lane serves
    const int owner = q == 0 ? l0 : q == 1 ? l1 : q == 2 ? l2 : l3;
    const uint32_t c0 = lane_pull(owner, pixel), c1 = lane_pull(owner, sample), c2 = lane_pull(owner, site);
    const int j = lane & 15, role = j & 3;
    const uint32_t index = 3u * (g + (uint32_t)(j >> 2)) + (uint32_t)(role < 3 ? role : 2);
    const u32x4 b = philox4x32_10(c0, c1, c2, index, k0, k1);
    const uint32_t w3 = lane_pull(lane - 3, b.w), w2 = lane_pull(lane - 2, b.w), w1 = lane_pull(lane - 1, b.w);
    const uint32_t x = role == 3 ? w3 : b.x, y = role == 3 ? w2 : b.y, z = role == 3 ? w1 : b.z;
    const uint64_t passed = ballot(q < n && inside_unit_sphere(x, y, z));
    // an owner looks at the sixteen attempts made for it
    const int base = (int)rank_below(m) << 4;
    const uint32_t mine = need ? (uint32_t)(passed >> base) & 0xFFFFu : 0u;
    const int from = mine != 0u ? base + __builtin_ctz(mine) : lane;
    const uint32_t px = lane_pull(from, x), py = lane_pull(from, y), pz = lane_pull(from, z);
    if (mine != 0u) {
        need = false;
        p.kx = px;
        p.ky = py;
        p.kz = pz;
    }
    return true;
}
#endif

// random_in_unit_sphere (B1/vec3.h:114-120) for the lanes with need == true, the blocks beyond block0 coming from
// philox_blocks_for_lanes: blocks 1 and 2 (attempts 1 to 3) in one evaluation, then three per evaluation (four attempts).
// Same draws as random_in_unit_sphere below.  All 64 lanes must call it.
template <bool COOP = (RT_COOP_SHADE != 0)>
__device__ __forceinline__ d3 random_in_unit_sphere_for_lanes(bool need, u32x4 block0, uint32_t pixel, uint32_t sample,
                                                              uint32_t site, uint32_t k0, uint32_t k1, int lane, uint8_t* owners) {
    if constexpr (!COOP) return random_in_unit_sphere(need, block0, pixel, sample, site, k0, k1);
    SpherePoint p;
    p.wanted = need;
    auto attempt = [&](uint32_t x, uint32_t y, uint32_t z) {
        if (need && p.attempt(x, y, z)) need = false;
    };
    auto point = [&]() { return p.value(); };
    attempt(block0.x, block0.y, block0.z);
    if (ballot(need) == 0ull) return point();
    {
        u32x4 b[2];
        philox_blocks_for_lanes<2>(need, pixel, sample, site, 1u, 0u, k0, k1, lane, owners, b);
        attempt(b[0].x, b[0].y, b[0].z);
        if (ballot(need) != 0ull) attempt(b[1].x, b[1].y, b[1].z);
        if (ballot(need) != 0ull) attempt(block0.w, b[0].w, b[1].w);
    }
    for (uint32_t g = 1; ballot(need) != 0ull; ++g) {
#if RT_INTEGER_SPHERE_TEST && RT_SPHERE_ATTEMPTS_BY_HELPERS
        if (sphere_attempts_by_helpers(need, p, pixel, sample, site, g, k0, k1, lane)) {
            g += 3u;  // groups g .. g+3 are done
            continue;
        }
#endif
        u32x4 b[3];
        philox_blocks_for_lanes<3>(need, pixel, sample, site, 3u * g, 0u, k0, k1, lane, owners, b);
        attempt(b[0].x, b[0].y, b[0].z);
        if (ballot(need) != 0ull) attempt(b[1].x, b[1].y, b[1].z);
        if (ballot(need) != 0ull) attempt(b[2].x, b[2].y, b[2].z);
        if (ballot(need) != 0ull) attempt(b[0].w, b[1].w, b[2].w);
    }
    return point();
}

// reflectance for an r0 = ((1 - ref_idx) / (1 + ref_idx))^2 (B1/material.h:85-86) worked out before: B1/material.h:87
__device__ __forceinline__ double reflectance_from_r0(double cosine, double r0) {
    const double x = 1 - cosine;
    const double x2 = x * x;
    return r0 + (1 - r0) * (x2 * x2 * x);  // pow(x, 5) by multiplication (<= 2 ulp from libm pow)
}
__device__ __forceinline__ double reflectance(double cosine, double ref_idx) {  // B1/material.h:83-88
    double r0 = (1 - ref_idx) / (1 + ref_idx);
    r0 = r0 * r0;
    const double x = 1 - cosine;
    const double x2 = x * x;
    return r0 + (1 - r0) * (x2 * x2 * x);  // pow(x, 5) by multiplication (<= 2 ulp from libm pow)
}

// checker_texture::value (B2/texture.h:46-52) needs only the SIGN of sin(10 x) sin(10 y) sin(10 z).  The sign of
// sin(x) is that of (-1)^k r with k = rint(x / pi), r = x - k pi in [-pi/2, pi/2]; r comes from one fused
// multiply-add against the leading 53 bits of pi, whose neglected tail moves it by less than |k| 1.3e-16 <= 4.2e-9
// for |x| < 1e8.  When |r| > 1e-8 its sign is therefore certain (a correctly signed libm sin, as the reference's and
// the device's are, gives the same); closer to a multiple of pi -- about one argument in 10^8 -- and for huge or
// non-finite arguments sin() itself is evaluated.  Returns -1, 0 or +1 (0: the product is +-0 or NaN, never "< 0").
__device__ __noinline__ int sin_sign_by_libm(double x) {  // out of line: one copy, and its registers are not the caller's
    const double s = sin(x);
    return s < 0 ? -1 : (s > 0 ? 1 : 0);
}
__device__ __forceinline__ int sin_sign(double x) {
    const double k = rint(x * 0.318309886183790671537767526745);
    const double r = fma(-k, 3.141592653589793116, x);
    if (fabs(x) < 1e8 && fabs(r) > 1e-8) return (((long long)k & 1ll) != 0) == (r < 0) ? 1 : -1;
    return sin_sign_by_libm(x);
}
__device__ __forceinline__ bool checker_sines_negative(d3 p) {
    return sin_sign(10 * p.x) * sin_sign(10 * p.y) * sin_sign(10 * p.z) < 0;
}

__device__ __forceinline__ void atomic_add_i64(int64_t* p, int64_t v) {
    __hip_atomic_fetch_add((unsigned long long*)p, (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The same for a word that only this wave adds to during the launch: a plain load, add and store.  An agent-scope
// atomic has to be coherent across the eight XCDs, whose L2 caches are not: it is carried out at the memory side, one
// 64-byte request each (measured: WRITE_SIZE = 64 B x atomics), and a workgroup-scope atomic still costs most of that.
// Plain stores stay in this XCD's L2 as dirty bytes and reach memory when the line is evicted or the kernel ends;
// neighbouring words of the line that other waves own are written with their own byte masks.  Two additions to the
// same word by this wave (a pixel that lost its cache slot and claimed one again) are ordered by the wavefront fences
// of lds_wave_sync between them.
__device__ __forceinline__ void add_i64_own(int64_t* p, int64_t v) { *p += v; }

// Work items.  The samples [0, N) of a pixel are cut into runs ("levels"), one work item each.
//
// Frame-by-frame plan (pixel_pass = false): K runs of `cap` samples, then the remaining R = N - K*cap
// (cap <= R < 2*cap when N >= cap) halved repeatedly: R - R/2, R/2 - R/4, ..., 1.  Items are numbered run by run over
// the whole frame (one-sample runs last), so that the queue ends in fine grains spread over all pixels, whatever single
// pixels cost.  Every item adds its sums to memory itself (three 64-bit atomics).
//
// Pixel-by-pixel plan (pixel_pass = true; `cap` then carries the NUMBER of runs per pixel, L): the N samples of a pixel in
// L runs as equal as they can be -- the first N mod L runs hold floor(N / L) + 1 samples, the others floor(N / L) --
// numbered pixel by pixel (all runs of pixel 0, then of pixel 1, ...).  A wave takes whole pixels from the queue (batches
// are multiples of the runs per pixel while they fit 64 lanes; with L a power of two a batch fills all 64), its lanes trace
// similar rays, and the sums of a pixel's runs meet in the wave's accumulation cache (below) and reach memory once per
// pixel instead of once per run.  Equal runs matter: with K runs of 8 and one of the remaining 8..15 samples, the long run
// was the last to finish of every pixel (63 samples per pixel: 9.66 -> 9.40 ms with runs of 7 and 8).  (Ending this queue in
// fine grains as well -- one-sample items, or halved runs of every n-th pixel, after the long runs -- was measured and
// changes nothing: the waves balance their own ends, see refill_work.)
// In that plan ItemPlan::K = number of long runs, cap = their length, R = the length of the others.
// (struct ItemPlan: rt_internal.h, beside the kernel arguments that carry it)
__host__ __device__ inline ItemPlan make_item_plan(int32_t n_samples, int32_t cap, uint32_t npix, bool pixel_pass) {
    ItemPlan p;
    p.npix = npix;
    if (pixel_pass) {
        int32_t runs = cap < 1 ? 1 : cap;
        if (runs > n_samples) runs = n_samples > 0 ? n_samples : 1;
        p.levels = n_samples > 0 ? runs : 0;
        p.R = n_samples / runs;
        p.K = n_samples - p.R * runs;  // runs that hold one sample more
        p.cap = p.R + 1;
        p.bulk_items = npix * (uint32_t)p.levels;
        p.exclusive = (p.levels >= 1 && p.levels <= 64) ? 1 : 0;
        p.batch = p.exclusive != 0 ? (uint32_t)((64 / p.levels) * p.levels) : 64u;
    } else {
        p.cap = cap < 1 ? 1 : cap;
        p.K = n_samples / p.cap - 1;
        if (p.K < 0) p.K = 0;
        p.R = n_samples - p.K * p.cap;
        int bits = 0;
        for (int32_t r = p.R; r > 0; r >>= 1) ++bits;
        p.levels = p.K + bits;
        p.bulk_items = 0u;
        p.batch = 64u;
        p.exclusive = 0;
    }
    p.by_levels = make_u32_divider((uint32_t)(p.levels > 0 ? p.levels : 1));
    p.by_npix = make_u32_divider(npix);
    return p;
}
__host__ __device__ inline uint64_t item_count(const ItemPlan& p, uint32_t npix) {
    return (uint64_t)npix * (uint64_t)p.levels;
}
// item number -> pixel and samples [first, first + count), relative to the first sample of the call
__device__ __forceinline__ void item_lookup(const ItemPlan& p, uint32_t item, uint32_t& pixel, int& first, int& count) {
    if (p.bulk_items != 0u) {
#if RT_ISSUE_BOUND  // (the scene-graph kernels keep the division: the dividers' registers cost their final-scene variant 5 %)
        pixel = div_u32(item, p.by_levels);
#else
        pixel = item / (uint32_t)p.levels;
#endif
        const int level = (int)(item - pixel * (uint32_t)p.levels);
        first = level * p.R + (level < p.K ? level : p.K);  // (the long runs come first)
        count = level < p.K ? p.cap : p.R;
        return;
    }
#if RT_ISSUE_BOUND
    const uint32_t l = div_u32(item, p.by_npix);
#else
    const uint32_t l = item / p.npix;
#endif
    pixel = item - l * p.npix;
    const int level = (int)l;
    if (level < p.K) {
        first = level * p.cap;
        count = p.cap;
    } else {
        const int g = level - p.K;
        first = p.K * p.cap + p.R - (p.R >> g);
        count = (p.R >> g) - (p.R >> (g + 1));
    }
}

// Accumulation cache of one wave (in LDS): fixed-point sums of the pixels the wave is working on.  A finished
// item adds its sums here; a pixel's sums go to memory (three 64-bit atomics) when another pixel needs its slot
// or when the wave ends.  With pixel-by-pixel item numbering almost all items of a pixel meet in one slot, which
// cuts the atomics that reach memory by the number of items per pixel.
constexpr int ACC_SLOTS = 16;
constexpr uint32_t ACC_EMPTY = 0xFFFFFFFFu;
struct AccumCache {
    int64_t sum[ACC_SLOTS][3];
    uint32_t tag[ACC_SLOTS];
};
// Slot of a pixel: a multiplicative hash, not the low bits -- the pixels a wave works on in a row are about as many
// pixels apart as there are waves (a power of two), and would all claim the same slot.
__device__ __forceinline__ int acc_slot_of(uint32_t pixel) { return (int)((pixel * 0x9E3779B1u) >> 28); }
static_assert(ACC_SLOTS == 16, "acc_slot_of keeps the top four bits of the hash");

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
    AccumCache* cache;  // this wave's accumulation cache (LDS), set up with accum_cache_init; nullptr: none, and the
                        // plan must have been made without the pixel-by-pixel pass
    uint32_t batch;     // items a wave takes from the queue at a time (ItemPlan::batch)
    bool exclusive;     // ItemPlan::exclusive: a pixel's sums reach memory from one wave only, through its cache
};

__device__ __forceinline__ void accum_cache_init(AccumCache* c, int lane) {
    if (lane < ACC_SLOTS) {
        c->tag[lane] = ACC_EMPTY;
        c->sum[lane][0] = c->sum[lane][1] = c->sum[lane][2] = 0;
    }
    lds_wave_sync();
}
__device__ __forceinline__ void accum_slot_to_memory(const WorkQueues& Q, int slot, uint32_t pixel) {
    if (Q.exclusive) {
        add_i64_own(Q.accum + 3ull * pixel + 0, Q.cache->sum[slot][0]);
        add_i64_own(Q.accum + 3ull * pixel + 1, Q.cache->sum[slot][1]);
        add_i64_own(Q.accum + 3ull * pixel + 2, Q.cache->sum[slot][2]);
    } else {
        atomic_add_i64(Q.accum + 3ull * pixel + 0, Q.cache->sum[slot][0]);
        atomic_add_i64(Q.accum + 3ull * pixel + 1, Q.cache->sum[slot][1]);
        atomic_add_i64(Q.accum + 3ull * pixel + 2, Q.cache->sum[slot][2]);
    }
}
// Adds the sums of the lanes with `fl` set (their finished items) to the cache.  All 64 lanes must call it.
// A lane whose pixel already owns its slot -- every item of a pixel but the first -- adds in one step; only the lanes
// that have to claim a slot (writing another pixel's sums back) go through the loop, one trip per distinct pixel.
__device__ __forceinline__ void accum_cache_add(const WorkQueues& Q, int lane, bool fl, uint32_t pixel, int64_t fx, int64_t fy,
                                                int64_t fz) {
    const int my_slot = acc_slot_of(pixel);
    const bool fast = fl && Q.cache->tag[my_slot] == pixel;
    if (fast) {
        __hip_atomic_fetch_add((unsigned long long*)&Q.cache->sum[my_slot][0], (unsigned long long)fx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add((unsigned long long*)&Q.cache->sum[my_slot][1], (unsigned long long)fy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add((unsigned long long*)&Q.cache->sum[my_slot][2], (unsigned long long)fz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    uint64_t todo = ballot(fl && !fast);
    if (todo == 0ull) return;
    lds_wave_sync();  // the additions above come before any slot changes hands
    while (todo != 0ull) {  // one trip per distinct pixel among the remaining lanes
        const int src = __builtin_ctzll(todo);
        const uint32_t p = (uint32_t)__builtin_amdgcn_readlane((int)pixel, src);
        const int slot = acc_slot_of(p);
        const uint32_t tag = (uint32_t)__builtin_amdgcn_readfirstlane((int)Q.cache->tag[slot]);
        if (tag != p) {  // wave-uniform: the slot holds another pixel (or nothing): write it back, take the slot
            if (lane == src) {
                if (tag != ACC_EMPTY) accum_slot_to_memory(Q, slot, tag);
                Q.cache->sum[slot][0] = Q.cache->sum[slot][1] = Q.cache->sum[slot][2] = 0;
                Q.cache->tag[slot] = p;
            }
            lds_wave_sync();
        }
        const bool mine = fl && !fast && pixel == p;
        if (mine) {
            __hip_atomic_fetch_add((unsigned long long*)&Q.cache->sum[slot][0], (unsigned long long)fx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add((unsigned long long*)&Q.cache->sum[slot][1], (unsigned long long)fy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add((unsigned long long*)&Q.cache->sum[slot][2], (unsigned long long)fz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        lds_wave_sync();
        todo &= ~ballot(mine);
    }
}
// Writes every occupied slot back; the wave calls it once, after its last item.
__device__ __forceinline__ void accum_cache_flush(const WorkQueues& Q, int lane) {
    if (Q.cache == nullptr) return;
    lds_wave_sync();
    if (lane < ACC_SLOTS) {
        const uint32_t tag = Q.cache->tag[lane];
        if (tag != ACC_EMPTY) accum_slot_to_memory(Q, lane, tag);
    }
}
struct LaneWork {
    bool has_item;
    bool cached;  // the item's sums go through the wave's accumulation cache (items of the pixel-by-pixel pass)
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
__device__ __forceinline__ bool refill_work(const WorkQueues& Q, const ItemPlan& plan, int32_t sample_begin,
                                            int lane, bool active, LaneWork& w, WaveWork& q) {
    {
        const bool finished = !active && w.has_item && w.s_cur == w.s_end;
        // Items of the pixel-by-pixel plan meet their pixel's other items in the cache; items of the frame-by-frame
        // plan (and samples handed over when a pixel's items are spread over several waves) go to memory directly.
        if (Q.cache != nullptr && ballot(finished && w.cached) != 0ull)
            accum_cache_add(Q, lane, finished && w.cached, w.pixel, w.fx, w.fy, w.fz);
        if (finished && !w.cached) {
            atomic_add_i64(Q.accum + 3ull * w.pixel + 0, w.fx);
            atomic_add_i64(Q.accum + 3ull * w.pixel + 1, w.fy);
            atomic_add_i64(Q.accum + 3ull * w.pixel + 2, w.fz);
        }
        if (finished) w.has_item = false;
    }
    bool need = !active && !w.has_item;
    while (!q.dry && ballot(need) != 0ull) {
        if (q.batch_next == q.batch_end) {
            uint32_t base = 0;
            const uint32_t take = Q.batch;
            if (lane == 0) base = atomicAdd(Q.counter, take);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (base >= Q.n_items) {
                q.dry = true;
                break;
            }
            q.batch_next = base;
            q.batch_end = min(base + take, Q.n_items);
        }
        const uint64_t m = ballot(need);
        const uint32_t rank = rank_below(m);
        const uint32_t avail = q.batch_end - q.batch_next;
        if (need && rank < avail) {
            int first, count;
            item_lookup(plan, q.batch_next + rank, w.pixel, first, count);
            w.cached = q.batch_next + rank < plan.bulk_items;
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
            w.cached = Q.exclusive;  // the giver's pixel is this wave's own: its cache takes the sample's sums as well
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

// Converts one sample colour to 2^-RT_ACCUM_FRAC_BITS fixed point and adds it.  A colour the format cannot hold --
// a channel of RT_ACCUM_SAMPLE_LIMIT or more in magnitude, an infinity, a NaN (every comparison with one is false) --
// is dropped and counted (the reference would carry it into its image).
__device__ __forceinline__ void add_sample(d3 c, int64_t& fx, int64_t& fy, int64_t& fz, uint32_t& n_bad) {
#if RT_ISSUE_BOUND
    // A double has no conversion to a 64-bit integer in hardware (eight instructions per channel).  Below 2^51 the
    // rounded integer can be read off the sum with 1.5 * 2^52 instead: the scaling by a power of two is exact, the
    // one rounding of the multiply-add is to the nearest integer, ties to even -- what __double2ll_rn does --, and the
    // sum's bit pattern is that of 1.5 * 2^52 plus the integer (two's complement).  Three instructions per channel.
    constexpr double SMALL = (double)(1ull << (51 - RT_ACCUM_FRAC_BITS));
    if (ballot(!(fabs(c.x) < SMALL && fabs(c.y) < SMALL && fabs(c.z) < SMALL)) == 0ull) {
        constexpr double MAGIC = 0x1.8p52;
        constexpr long long MAGIC_BITS = 0x4338000000000000ll;
        fx += __double_as_longlong(__builtin_fma(c.x, ACCUM_SCALE, MAGIC)) - MAGIC_BITS;
        fy += __double_as_longlong(__builtin_fma(c.y, ACCUM_SCALE, MAGIC)) - MAGIC_BITS;
        fz += __double_as_longlong(__builtin_fma(c.z, ACCUM_SCALE, MAGIC)) - MAGIC_BITS;
        return;
    }
#endif
    if (fabs(c.x) < RT_ACCUM_SAMPLE_LIMIT && fabs(c.y) < RT_ACCUM_SAMPLE_LIMIT && fabs(c.z) < RT_ACCUM_SAMPLE_LIMIT) {
        fx += __double2ll_rn(c.x * ACCUM_SCALE);
        fy += __double2ll_rn(c.y * ACCUM_SCALE);
        fz += __double2ll_rn(c.z * ACCUM_SCALE);
    } else {
        ++n_bad;
    }
}

}  // namespace rtd
