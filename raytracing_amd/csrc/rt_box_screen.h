// Closest hit of one ray per lane against a set of axis-aligned boxes held as a flat list: a float slab test of all
// boxes (wave-wide, box data in scalar registers) followed by the reference's double-precision test of the few
// candidates.  Used by the scene-graph kernels for BVH subtrees whose leaves are boxes.
#pragma once

#include "rt_sphere_screen.h"

namespace rtd {

// ---------------------------------------------------------------- screening pass
//
// Every side of a box lies in the closed box [lo, hi] (B2/box.h:31-38), so a hit at t in [t_min, t_max] implies that
// t lies in all three slab intervals  [ (lo_k - o_k)/d_k , (hi_k - o_k)/d_k ]  (ends swapped for d_k < 0) and that
// t >= 0.  The screen evaluates, in float,
//     a_k = lo_k * inv_k + c_lo_k,   b_k = hi_k * inv_k + c_hi_k,        inv_k = 1/d_k
//     near = max(min(a_k, b_k), 0),  far = min(max(a_k, b_k)),           candidate iff near <= far
// with c_lo_k = -o_k inv_k - s mu_k, c_hi_k = -o_k inv_k + s mu_k, s = sign(inv_k): each slab interval is widened by
// mu_k at both ends.  The float image rounds lo down and hi up (which only widens the intervals); rounding inv_k and
// c to float and the fma itself perturb a_k and b_k by less than 2 eps (B + |o_k|) |inv_k|, eps = 2^-24, B = the
// largest |coordinate| of the set; mu_k = 2^-21 (B + |o_k|) |inv_k| is four times that.  So every float interval
// contains the exact one, and a box the reference hits is always a candidate.  Rays the bound does not cover
// (|o_k| > 1e6, |d_k| outside [1e-30, 1e30], non-finite) test every box in double instead; with them excluded and
// B <= 1e6 (checked at upload) no product overflows, so no NaN can arise.
struct BoxScreenRay {
    f2 inv[3], c_lo[3], c_hi[3];
    bool covered;
};

__device__ __forceinline__ BoxScreenRay box_screen_prepare(d3 o, d3 d, float coord_max) {
    BoxScreenRay s;
    s.covered = true;
    const double ok[3] = {o.x, o.y, o.z}, dk[3] = {d.x, d.y, d.z};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double ad = fabs(dk[k]), ao = fabs(ok[k]);
        if (!(ad >= 1e-30 && ad <= 1e30 && ao <= 1e6)) s.covered = false;
        const double inv = 1.0 / dk[k];
        const double base = -ok[k] * inv;
        const double mu = 0x1p-21 * ((double)coord_max + ao) * fabs(inv);
        const double lo_c = inv >= 0 ? base - mu : base + mu;
        const double hi_c = inv >= 0 ? base + mu : base - mu;
        s.inv[k] = splat(inv);
        s.c_lo[k] = splat(lo_c);
        s.c_hi[k] = splat(hi_c);
    }
    return s;
}

typedef __attribute__((address_space(4))) BoxPair ConstBoxPair;

// Screens the two boxes of *p; appends their sign bits (first box first; 1 = no candidate) to mask.
__device__ __forceinline__ uint32_t box_screen_pair(const BoxScreenRay& r, const ConstBoxPair* p, uint32_t mask) {
    const f2 ax = pk_fma(f2{p->lo[0][0], p->lo[0][1]}, r.inv[0], r.c_lo[0]);
    const f2 bx = pk_fma(f2{p->hi[0][0], p->hi[0][1]}, r.inv[0], r.c_hi[0]);
    const f2 ay = pk_fma(f2{p->lo[1][0], p->lo[1][1]}, r.inv[1], r.c_lo[1]);
    const f2 by = pk_fma(f2{p->hi[1][0], p->hi[1][1]}, r.inv[1], r.c_hi[1]);
    const f2 az = pk_fma(f2{p->lo[2][0], p->lo[2][1]}, r.inv[2], r.c_lo[2]);
    const f2 bz = pk_fma(f2{p->hi[2][0], p->hi[2][1]}, r.inv[2], r.c_hi[2]);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float near_t = fmaxf(fmaxf(fminf(ax[h], bx[h]), fminf(ay[h], by[h])), fmaxf(fminf(az[h], bz[h]), 0.0f));
        const float far_t = fminf(fminf(fmaxf(ax[h], bx[h]), fmaxf(ay[h], by[h])), fmaxf(az[h], bz[h]));
        mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(far_t - near_t), 31);
    }
    return mask;
}

// Closest hit in [t_min, t_max] of one ray per lane against the boxes of the set; exact(j, h) is the reference's test
// of box j (lowers h.t and sets h.id = j on a hit).  Must be called by all 64 lanes; `list` as for the spheres.
template <class ExactFn>
__device__ __forceinline__ SetHit closest_box_hit(const BoxSetDev& B, d3 o, d3 d, double t_max, bool on, uint16_t* list,
                                                  int lane, ExactFn exact) {
    SetHit h{t_max, -1};
    const BoxScreenRay sr = box_screen_prepare(o, d, B.coord_max);
    uint32_t cnt = (on && !sr.covered) ? (uint32_t)CAND_CAP + 1u : 0u;
    const ConstBoxPair* bp = (const ConstBoxPair*)B.screen;
    const int n_groups = (B.n + SCREEN_GROUP - 1) / SCREEN_GROUP;
    for (int g = 0; g < n_groups; ++g) {
        uint32_t mask = 0;
#pragma unroll
        for (int jj = 0; jj < SCREEN_GROUP / 2; ++jj) mask = box_screen_pair(sr, bp + jj, mask);
        bp += SCREEN_GROUP / 2;
        const int present = B.n - g * SCREEN_GROUP;  // the image is padded to whole groups; padding is no candidate
        uint32_t cand = on ? ~mask : 0u;
        if (present < SCREEN_GROUP) cand &= ~(0xFFFFFFFFu >> present);
        while (ballot(cand != 0u) != 0ull) {
            if (cand != 0u) {
                const int z = __clz((int)cand);
                cand &= ~(0x80000000u >> z);
                if (cnt < (uint32_t)CAND_CAP) list[cnt * 64 + lane] = (uint16_t)(g * SCREEN_GROUP + z);
                ++cnt;
            }
        }
    }
    const bool overflow = cnt > (uint32_t)CAND_CAP;
    const uint32_t my_n = overflow ? 0u : cnt;
    for (uint32_t i = 0; ballot(i < my_n) != 0ull; ++i) {
        if (i < my_n) exact((int)list[i * 64 + lane], h);
    }
    if (ballot(overflow) != 0ull) {
        if (overflow) {  // more candidates than slots, or a ray the float bound does not cover: every box in double
            for (int j = 0; j < B.n; ++j) exact(j, h);
        }
    }
    return h;
}

}  // namespace rtd
