// Closest hit of one ray per lane against a set of axis-aligned boxes held as a flat list: a float slab test of all
// boxes (wave-wide, box data in scalar registers) followed by the reference's double-precision test of the few
// candidates.  Used by the scene-graph kernels for BVH subtrees whose leaves are boxes.
#pragma once

#include "rt_sphere_screen.h"

namespace rtd {

// ---------------------------------------------------------------- screening pass
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

__device__ __forceinline__ BoxScreenRay box_screen_prepare(d3 o, d3 d, float cover) {
    BoxScreenRay s;
    s.covered = true;
    const double ok[3] = {o.x, o.y, o.z}, dk[3] = {d.x, d.y, d.z};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double ad = fabs(dk[k]), ao = fabs(ok[k]);
        if (!(ad >= 1e-30 && ad <= 1e30 && ao <= (double)cover)) s.covered = false;
        const double inv = 1.0 / dk[k];
        s.inv[k] = splat(inv);
        s.ainv[k] = f2{fabsf(s.inv[k].x), fabsf(s.inv[k].x)};
        s.c[k] = splat(-ok[k] * inv);
    }
    return s;
}

typedef __attribute__((address_space(4))) BoxPair ConstBoxPair;

// Screens the two boxes of *p; appends their sign bits (first box first; 1 = no candidate) to mask.
__device__ __forceinline__ uint32_t box_screen_pair(const BoxScreenRay& r, const ConstBoxPair* p, uint32_t mask) {
    f2 near_t[3], far_t[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const f2 hw{p->hw[k][0], p->hw[k][1]};
        const f2 tc = pk_fma(f2{p->mid[k][0], p->mid[k][1]}, r.inv[k], r.c[k]);
        near_t[k] = pk_fma(hw, -r.ainv[k], tc);
        far_t[k] = pk_fma(hw, r.ainv[k], tc);
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

// Closest hit in [t_min, t_max] of one ray per lane against the boxes of the set; exact(j, h) is the reference's test
// of box j (lowers h.t and sets h.id = j on a hit).  Must be called by all 64 lanes; `list` as for the spheres.
// exact_only must be the same for the whole wave.
template <class ExactFn>
__device__ __forceinline__ SetHit closest_box_hit(const BoxSetDev& B, d3 o, d3 d, double t_max, bool on, bool exact_only,
                                                  uint16_t* list, int lane, ExactFn exact) {
    SetHit h{t_max, -1};
    if (exact_only) {  // validation: no float screen, every box in double
        if (on) {
            for (int j = 0; j < B.n; ++j) exact(j, h);
        }
        return h;
    }
    const BoxScreenRay sr = box_screen_prepare(o, d, B.cover);
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
