// Closest hit of one ray per lane against a set of spheres held as a flat list: a float screening pass over all
// spheres (wave-wide, sphere data in scalar registers) followed by the reference's double-precision test of the few
// candidates.  Shared by the sphere-list kernels (the whole world is one set) and the scene-graph kernels (a BVH
// subtree that holds only spheres is one set).
#pragma once

#include "rt_device_common.h"
#include "rt_internal.h"

namespace rtd {

// Section clock of the debug build (-DRT_TAIL_PROBE): elapsed cycles of a wave per kernel section.
// Empty, and compiled away, otherwise.
#ifdef RT_TAIL_PROBE
struct ProbeClock {
    uint64_t sec[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t last = 0;
    __device__ __forceinline__ void start() { last = clock64(); }
    __device__ __forceinline__ void mark(int k) {
        const uint64_t now = clock64();
        sec[k] += now - last;
        last = now;
    }
};
#else
struct ProbeClock {
    __device__ __forceinline__ void start() {}
    __device__ __forceinline__ void mark(int) {}
};
#endif
enum { SEC_REFILL = 0, SEC_SCREEN_DENSE, SEC_SCREEN_SPARSE, SEC_EXACT, SEC_SHADE, SEC_PREP };

// ---------------------------------------------------------------- screening pass
//
// For a ray o + t*d (d normalised to d^) and a sphere (c, r), real roots exist iff
//     D = (oc.d^)^2 - |oc|^2 + r^2 >= 0,      oc = o - c.
// Expanded so that each VALU op has exactly one per-sphere (scalar) operand:
//     hb  = o.d^ - c.d^                                   3 fma
//     acc = (m_o - |o|^2) - k + 2 o.c                     1 sub + 3 fma,  k = |c|^2 - r^2 - m_c
//     q   = hb^2 + acc  = D + m_o + m_c                   1 fma
// Every intermediate is bounded by (|o|+|c|)^2 + r^2, and with eps = 2^-24 the rounding of
// the float inputs (o, d^, c, k) plus the 8 operations perturbs q by less than
// 28 eps ((|o|+|c|)^2 + r^2) <= 56 eps (|o|^2 + |c|^2) + 28 eps r^2.  The margins
//     m_o = 2^-18 |o|^2,   m_c = 2^-18 (|c|^2 + r^2)        (2^-18 = 64 eps)
// exceed that, so a sphere whose reference discriminant (a double, relative error 1e-15)
// is >= 0 always yields q > 0.  The sign bit of q is shifted into the mask: 0 = candidate.
//
// Issue rates measured on gfx950 (tools/ubench): a wave64 v_fma_f32 whose operands are all
// VGPRs issues every ~2.2 cycles, but with an SGPR operand every ~4.2 cycles; v_pk_fma_f32
// (two FMAs per lane) also issues every ~4.1 cycles and accepts an SGPR pair.  The 8 ops are
// therefore done on PAIRS of spheres with packed math: 8 packed ops + 2 v_alignbit per pair,
// 23 cycles per sphere per SIMD at 4 waves/SIMD against 31 for the unpacked form.
typedef float f2 __attribute__((ext_vector_type(2)));

struct ScreenRay {
    f2 ndx, ndy, ndz;     // -d^, both halves equal
    f2 p2ox, p2oy, p2oz;  // 2 o
    f2 od;                // o . d^
    f2 e;                 // m_o - |o|^2
};

__device__ __forceinline__ f2 splat(double x) {
    const float f = (float)x;
    return f2{f, f};
}

__device__ __forceinline__ ScreenRay screen_prepare(d3 o, d3 d, double a /* |d|^2 */) {
    const double inv_len = 1.0 / sqrt(a);
    const d3 dh = inv_len * d;
    const double oo = len2(o);
    ScreenRay s;
    s.ndx = splat(-dh.x);
    s.ndy = splat(-dh.y);
    s.ndz = splat(-dh.z);
    s.p2ox = splat(2.0 * o.x);
    s.p2oy = splat(2.0 * o.y);
    s.p2oz = splat(2.0 * o.z);
    s.od = splat(dot(o, dh));
    s.e = splat(oo * (0x1p-18 - 1.0));
    return s;
}

typedef __attribute__((address_space(4))) SpherePair ConstSpherePair;

__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

// Screens two spheres given as packed operands; appends their sign bits (first sphere first) to mask.
__device__ __forceinline__ uint32_t screen_pair_ops(const ScreenRay& r, f2 cx, f2 cy, f2 cz, f2 k, uint32_t mask) {
    f2 hb = pk_fma(cx, r.ndx, r.od);
    hb = pk_fma(cy, r.ndy, hb);
    hb = pk_fma(cz, r.ndz, hb);
    f2 acc = r.e - k;
    acc = pk_fma(r.p2ox, cx, acc);
    acc = pk_fma(r.p2oy, cy, acc);
    acc = pk_fma(r.p2oz, cz, acc);
    const f2 q = pk_fma(hb, hb, acc);
    mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(q.x), 31);
    return __builtin_amdgcn_alignbit(mask, __float_as_uint(q.y), 31);
}
// The pair *s read through the constant address space (scalar loads, SGPR-pair operands).
__device__ __forceinline__ uint32_t screen_pair(const ScreenRay& r, const ConstSpherePair* s, uint32_t mask) {
    return screen_pair_ops(r, f2{s->cx[0], s->cx[1]}, f2{s->cy[0], s->cy[1]}, f2{s->cz[0], s->cz[1]}, f2{s->k[0], s->k[1]}, mask);
}
// The pair p held in this lane's registers.
__device__ __forceinline__ uint32_t screen_pair(const ScreenRay& r, const SpherePair& p, uint32_t mask) {
    return screen_pair_ops(r, f2{p.cx[0], p.cx[1]}, f2{p.cy[0], p.cy[1]}, f2{p.cz[0], p.cz[1]}, f2{p.k[0], p.k[1]}, mask);
}

__device__ __forceinline__ f2 splat_from_lane(f2 v, int src_lane) {
    const float f = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v.x), src_lane));
    return f2{f, f};
}

// Sparse waves.  When few lanes hold a ray (the end of a launch), walking all spheres with all 64 lanes
// wastes most of the wave.  Instead the wave turns by 90 degrees: lane t keeps spheres 8t..8t+7 of a
// 512-sphere chunk in registers, the rays are taken one at a time (their screening constants broadcast
// from the owning lane), every lane screens its 8 spheres against that ray with the same packed
// operations, and the 8 result bits go to byte t of the ray's row in LDS -- 64 bytes = the same 512-bit
// candidate mask the dense walk produces, at ~1/40 of its cost per ray.
#ifndef RT_SPARSE_MAX_RAYS
#define RT_SPARSE_MAX_RAYS 32
#endif
constexpr int SPARSE_MAX_RAYS = RT_SPARSE_MAX_RAYS;  // dense walk: ~10k cycles per wave; this path: ~300 per ray and chunk
constexpr int ROW_WORDS = 17;        // 16 mask words per ray + 1 pad (bank spread)
constexpr int SPARSE_CHUNK_PAIRS = 256;

// ---------------------------------------------------------------- exact pass
struct SetHit {
    double t;
    int32_t id;
};

// sphere::hit up to the accepted root, B1/sphere.h:24-41, inside the closest-so-far loop of
// hittable_list::hit, B1/hittable_list.h:31-37.  Comparisons keep the reference's form so
// that NaNs fall the same way.
// Centre of sphere j at time tm: moving_sphere::center, B2/moving_sphere.h:27-29.
template <bool TIMED>
__device__ __forceinline__ d3 center_at(const SphereSetDev& S, int j, const SphereExact& e, double tm) {
    d3 center{e.cx, e.cy, e.cz};
    if constexpr (TIMED) {
        const SphereMotion mv = S.motion[j];
        if (mv.moving != 0) center = center + ((tm - mv.time0) / (mv.time1 - mv.time0)) * (d3{mv.c1x, mv.c1y, mv.c1z} - center);
    }
    return center;
}

template <bool TIMED>
__device__ __forceinline__ void exact_one(const SphereSetDev& S, int j, d3 o, d3 d, double a, double tm, double t_min, SetHit& h) {
    const SphereExact e = S.exact[j];
    const d3 oc = o - center_at<TIMED>(S, j, e, tm);
    const double half_b = dot(oc, d);
    const double c = len2(oc) - e.radius * e.radius;
    const double discriminant = half_b * half_b - a * c;
    if (discriminant < 0) return;
    const double sqrtd = sqrt(discriminant);
    double root = (-half_b - sqrtd) / a;
    if (root < t_min || root > h.t) {
        root = (-half_b + sqrtd) / a;
        if (root < t_min || root > h.t) return;
    }
    h.t = root;
    h.id = j;
}

// Closest hit in [t_min, t_max] of one ray per lane against the whole set; id = -1 if there is none.  Must be called
// by all 64 lanes of the wave (wave-level loops); lanes with on == false take no part in the result.
// `list` = this wave's CAND_CAP*64 uint16 slots in LDS, `rows` = its ROW_WORDS*64 words.
// exact(j, h) is the double-precision test of member j: it lowers h.t and sets h.id = j if j is hit in [t_min, h.t].
// The screening image holds, per member, a sphere that the member lies inside.
template <class ExactFn>
__device__ __forceinline__ SetHit closest_hit_with(const SphereSetDev& S, d3 o, d3 d, double t_max, bool on, bool exact_only,
                                                   uint16_t* list, uint32_t* rows, int lane, uint32_t& overflow_count,
                                                   ProbeClock& pc, ExactFn exact) {
    const double a = len2(d);
    SetHit h{t_max, -1};

    if (exact_only) {
        if (on) {
            for (int j = 0; j < S.n; ++j) exact(j, h);
        }
        return h;
    }

    const ScreenRay sr = screen_prepare(o, d, a);
    // A ray with a NaN or an infinity in it (book 3's light sampling produces them: the reference drops such samples at
    // the end) has no meaningful float screen: the sign of a NaN is arbitrary, so it could name padding entries as
    // candidates and drop real ones.  The reference's own comparisons all come out false for such a ray, which makes
    // every object it tests a "hit" -- so the lane tests every member in double, in order, like the reference.
    const double ray_sum = ((o.x + o.y) + o.z) + ((d.x + d.y) + d.z);
    const bool finite_ray = ray_sum - ray_sum == 0.0;
    uint32_t cnt = (on && !finite_ray) ? (uint32_t)CAND_CAP + 1u : 0u;
    on = on && finite_ray;  // such lanes take no part in the screen; the overflow pass below serves them
    const uint64_t on_mask = ballot(on);
    pc.mark(SEC_PREP);
    if (__popcll(on_mask) <= SPARSE_MAX_RAYS) {
        for (int c = 0; c < S.n_chunks; ++c) {
            const SpherePair* mine = S.screen + (size_t)c * SPARSE_CHUNK_PAIRS + (size_t)lane * 4;
            const SpherePair p0 = mine[0], p1 = mine[1], p2 = mine[2], p3 = mine[3];
            for (uint64_t m = on_mask; m != 0ull; m &= m - 1ull) {
                const int src = __builtin_ctzll(m);
                ScreenRay rl;
                rl.ndx = splat_from_lane(sr.ndx, src);
                rl.ndy = splat_from_lane(sr.ndy, src);
                rl.ndz = splat_from_lane(sr.ndz, src);
                rl.p2ox = splat_from_lane(sr.p2ox, src);
                rl.p2oy = splat_from_lane(sr.p2oy, src);
                rl.p2oz = splat_from_lane(sr.p2oz, src);
                rl.od = splat_from_lane(sr.od, src);
                rl.e = splat_from_lane(sr.e, src);
                uint32_t mk = screen_pair(rl, p0, 0u);
                mk = screen_pair(rl, p1, mk);
                mk = screen_pair(rl, p2, mk);
                mk = screen_pair(rl, p3, mk);
                ((uint8_t*)rows)[src * (ROW_WORDS * 4) + lane] = (uint8_t)~mk;  // bit 7 = sphere 8*lane
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int g = 0; g < 2 * SPARSE_CHUNK_PAIRS / SCREEN_GROUP; ++g) {
                // bytes 4g..4g+3 of the row, first byte on top: bit 31 = first sphere of the group
                uint32_t cand = on ? __builtin_bswap32(rows[lane * ROW_WORDS + g]) : 0u;
                while (ballot(cand != 0u) != 0ull) {
                    if (cand != 0u) {
                        const int z = __clz((int)cand);
                        cand &= ~(0x80000000u >> z);
                        const int member = c * 2 * SPARSE_CHUNK_PAIRS + g * SCREEN_GROUP + z;
                        if (member < S.n) {  // never a padding entry, whatever the arithmetic said
                            if (cnt < (uint32_t)CAND_CAP) list[cnt * 64 + lane] = (uint16_t)member;
                            ++cnt;
                        }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        pc.mark(SEC_SCREEN_SPARSE);
    } else {
    // The screen image is immutable for the whole launch: address it through the constant
    // address space so that the wave-uniform reads become scalar loads (SGPR operands).
    const ConstSpherePair* sp = (const ConstSpherePair*)S.screen;
    const int n_groups = S.n_full_groups + (S.n_tail_pairs > 0 ? 1 : 0);
    for (int g = 0; g < n_groups; ++g) {
        uint32_t mask = 0;
        if (g < S.n_full_groups) {
#pragma unroll
            for (int jj = 0; jj < SCREEN_GROUP / 2; ++jj) mask = screen_pair(sr, sp + jj, mask);
            sp += SCREEN_GROUP / 2;
        } else {  // last, partial group: bits of absent spheres read "no candidate"
            mask = ~0u;
#pragma unroll 1
            for (int jj = 0; jj < S.n_tail_pairs; ++jj) mask = screen_pair(sr, sp + jj, mask);
            mask = ~(~mask << (SCREEN_GROUP - 2 * S.n_tail_pairs));
        }
        uint32_t cand = on ? ~mask : 0u;
        while (ballot(cand != 0u) != 0ull) {
            if (cand != 0u) {
                const int z = __clz((int)cand);
                cand &= ~(0x80000000u >> z);
                const int member = g * SCREEN_GROUP + z;
                if (member < S.n) {  // the padded half of an odd last pair is never a candidate
                    if (cnt < (uint32_t)CAND_CAP) list[cnt * 64 + lane] = (uint16_t)member;
                    ++cnt;
                }
            }
        }
    }
    pc.mark(SEC_SCREEN_DENSE);
    }

    const bool overflow = cnt > (uint32_t)CAND_CAP;
    const uint32_t my_n = overflow ? 0u : cnt;
    for (uint32_t i = 0; ballot(i < my_n) != 0ull; ++i) {
        if (i < my_n) exact((int)list[i * 64 + lane], h);
    }
    if (ballot(overflow) != 0ull) {
        if (overflow) {  // more candidates than slots: test the whole list in double
            ++overflow_count;
            for (int j = 0; j < S.n; ++j) exact(j, h);
        }
    }
    pc.mark(SEC_EXACT);
    return h;
}

// The set's members are its spheres themselves (sphere-list kernels).
template <bool TIMED>
__device__ __forceinline__ SetHit closest_hit(const SphereSetDev& S, d3 o, d3 d, double tm, double t_max, bool on, bool exact_only,
                                              uint16_t* list, uint32_t* rows, int lane, uint32_t& overflow_count,
                                              ProbeClock& pc) {
    const double a = len2(d);
    const double t_min = 0.001;  // B1/RayTracing.cpp:17
    return closest_hit_with(S, o, d, t_max, on, exact_only, list, rows, lane, overflow_count, pc,
                            [&](int j, SetHit& h) { exact_one<TIMED>(S, j, o, d, a, tm, t_min, h); });
}

}  // namespace rtd
