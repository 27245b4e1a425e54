// Float screening arithmetic shared by the set walks: the conservative discriminant test of a ray against pairs of
// spheres and the slab test against pairs of axis-aligned boxes, both in packed FP32 with the pair's data either in
// scalar registers (wave-uniform walk) or in the lane's own registers.
#pragma once

#include "rt_device_common.h"
#include "rt_internal.h"

namespace rtd {

// ---------------------------------------------------------------- spheres
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

// The ray constants of the two screens feed float arithmetic whose margins are sized for inputs rounded to float
// (relative 2^-24): a reciprocal or reciprocal square root good to 2^-45 serves as well as the correctly rounded one.
// The hardware's estimate (v_rcp_f64, v_rsq_f64: about 2^-26) and one Newton step: 3 and 5 vector instructions where
// the IEEE division takes 11 and 1 / sqrt(a) 30.  Used where vector issue is the bound (RT_ISSUE_BOUND); arguments
// outside [1e-200, 1e200] give values no screen uses (such rays take the unscreened tests).
__device__ __forceinline__ double reciprocal_for_screen(double x) {
#if RT_ISSUE_BOUND
    const double r = __builtin_amdgcn_rcp(x);
    return __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
#else
    return 1.0 / x;
#endif
}
__device__ __forceinline__ double rsqrt_for_screen(double a) {
#if RT_ISSUE_BOUND
    const double y = __builtin_amdgcn_rsq(a);
    return __builtin_fma(y, __builtin_fma(-(a * y), 0.5 * y, 0.5), y);
#else
    return 1.0 / sqrt(a);
#endif
}

__device__ __forceinline__ ScreenRay screen_prepare(d3 o, d3 d, double a /* |d|^2 */) {
    const double inv_len = rsqrt_for_screen(a);
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

// ---------------------------------------------------------------- boxes
//
// Every side of a box lies in the closed box [lo, hi] (B2/box.h:31-38), so a hit at t in [t_min, t_max] implies that
// t >= 0 and that t lies in all three slab intervals, which have centre (mid_k - o_k)/d_k and half-width
// hw_k/|d_k| (mid = (lo+hi)/2, hw = (hi-lo)/2).  The screen evaluates, in float and on box pairs,
//     tc_k   = mid_k * inv_k + c_k                 inv_k = 1/d_k,  c_k = -o_k inv_k
//     near_k = tc_k - hw'_k |inv_k|,  far_k = tc_k + hw'_k |inv_k|
//     candidate iff  max_k near_k <= min_k far_k  and  min_k far_k >= 0.
// Rounding mid, inv and c to float and the three fused operations perturb near_k and far_k by less than
// 5 eps (B + |o_k|) |inv_k| (eps = 2^-24, B = largest |coordinate| of the set).  The image's half-widths are
// hw' = hw + |mid - float(mid)| + 2^-21 (B + O), rounded up: for rays with |o_k| <= O every float interval contains the
// exact one (8 eps against 5 eps), so a box the reference hits is always a candidate.  O ("cover") is eight times
// the largest coordinate of the whole world; rays the bound does not cover (|o_k| > O, |d_k| outside [1e-30, 1e30],
// non-finite) test every box in double instead.  With those excluded and B, O <= 8e6 (checked at upload) no product
// overflows, so no NaN arises.
struct BoxScreenRay {
    f2 inv[3], ainv[3], c[3];
    bool covered;
};

__device__ __forceinline__ BoxScreenRay box_screen_prepare(d3 o, d3 d, double cover) {
    BoxScreenRay s;
    s.covered = true;
    const double ok[3] = {o.x, o.y, o.z}, dk[3] = {d.x, d.y, d.z};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double ad = fabs(dk[k]), ao = fabs(ok[k]);
        if (!(ad >= 1e-30 && ad <= 1e30 && ao <= cover)) s.covered = false;
        const double inv = reciprocal_for_screen(dk[k]);
        s.inv[k] = splat(inv);
        s.ainv[k] = f2{fabsf(s.inv[k].x), fabsf(s.inv[k].x)};
        s.c[k] = splat(-ok[k] * inv);
    }
    return s;
}

typedef __attribute__((address_space(4))) BoxPair ConstBoxPair;

// Screens two boxes given as packed operands (centre and widened half-width per axis); appends their sign bits (first
// box first; 1 = no candidate) to mask.
__device__ __forceinline__ uint32_t box_screen_pair_ops(const BoxScreenRay& r, const f2 (&mid)[3], const f2 (&hw)[3], uint32_t mask) {
    f2 near_t[3], far_t[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const f2 tc = pk_fma(mid[k], r.inv[k], r.c[k]);
        near_t[k] = pk_fma(hw[k], -r.ainv[k], tc);
        far_t[k] = pk_fma(hw[k], r.ainv[k], tc);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float far_all = fminf(fminf(far_t[0][h], far_t[1][h]), far_t[2][h]);
        const float near_all = fmaxf(fmaxf(near_t[0][h], near_t[1][h]), near_t[2][h]);
        const float q = fminf(far_all - near_all, far_all);  // negative: the intervals do not meet, or the box is behind
        mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(q), 31);
    }
    return mask;
}
// The pair *p read through the constant address space (scalar loads, SGPR-pair operands).
__device__ __forceinline__ uint32_t box_screen_pair(const BoxScreenRay& r, const ConstBoxPair* p, uint32_t mask) {
    const f2 mid[3] = {f2{p->mid[0][0], p->mid[0][1]}, f2{p->mid[1][0], p->mid[1][1]}, f2{p->mid[2][0], p->mid[2][1]}};
    const f2 hw[3] = {f2{p->hw[0][0], p->hw[0][1]}, f2{p->hw[1][0], p->hw[1][1]}, f2{p->hw[2][0], p->hw[2][1]}};
    return box_screen_pair_ops(r, mid, hw, mask);
}
// The pair p held in this lane's registers.
__device__ __forceinline__ uint32_t box_screen_pair(const BoxScreenRay& r, const BoxPair& p, uint32_t mask) {
    const f2 mid[3] = {f2{p.mid[0][0], p.mid[0][1]}, f2{p.mid[1][0], p.mid[1][1]}, f2{p.mid[2][0], p.mid[2][1]}};
    const f2 hw[3] = {f2{p.hw[0][0], p.hw[0][1]}, f2{p.hw[1][0], p.hw[1][1]}, f2{p.hw[2][0], p.hw[2][1]}};
    return box_screen_pair_ops(r, mid, hw, mask);
}

}  // namespace rtd
