// Closest hit of one ray per lane against a set of axis-aligned boxes held as a flat list: a float slab test of all
// boxes (wave-wide, box data in scalar registers) followed by the reference's double-precision test of the few
// candidates.  Used by the scene-graph kernels for BVH subtrees whose leaves are boxes.
#pragma once

#include "rt_sphere_screen.h"

namespace rtd {

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
