// Closest hit of one ray per lane against a set of axis-aligned boxes held as a flat list: a float slab test of all
// boxes (wave-wide, box data in scalar registers) -- or, for a set with a two-level form, of the boxes of its clusters
// and then, item by item, of the members of the clusters a ray meets -- followed by the reference's double-precision
// test of the few candidates.  Used by the scene-graph kernels for BVH subtrees whose leaves are boxes.
#pragma once

#include "rt_sphere_screen.h"

namespace rtd {

// Second level of the two-level walk, as screen_clusters does it for spheres: the queued (ray, cluster) items 64 at a
// time, one per lane; the ray's constants by lane permutation, the cluster's members by vector loads, candidates to the
// owning ray's list through the counters in LDS.
__device__ __forceinline__ void screen_box_clusters(const BoxSetDev& B, const BoxScreenRay& br, bool on, uint16_t* list,
                                                    uint32_t* rows, int lane, uint32_t& cnt) {
    const BoxClustersDev& C = B.clusters;
    uint16_t* items = (uint16_t*)rows;
    uint32_t* counts = rows + ITEM_CAP / 2;
    bool dropped;
    const uint32_t n_items = queue_cluster_items(C.boxes, C.n_clusters, br, on, items, lane, dropped);
    if (dropped) cnt = (uint32_t)CAND_CAP + 1u;
    counts[lane] = cnt;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (uint32_t base = 0; base < n_items; base += 64u) {
        const bool live = base + (uint32_t)lane < n_items;
        const uint32_t item = live ? (uint32_t)items[base + lane] : 0u;
        const int ray = (int)(item >> ITEM_CLUSTER_BITS), c = (int)(item & ((1u << ITEM_CLUSTER_BITS) - 1u));
        BoxScreenRay rl;  // the constants of the item's ray (all lanes take part in the permutations)
        rl.covered = true;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            rl.inv[k] = splat_from_lane_of(br.inv[k], ray);
            rl.ainv[k] = f2{fabsf(rl.inv[k].x), fabsf(rl.inv[k].x)};
            rl.c[k] = splat_from_lane_of(br.c[k], ray);
        }
        if (live) {
            const BoxPair* mp = C.pairs + (size_t)c * (CLUSTER_SIZE / 2);
            uint32_t mk = 0;
#pragma unroll
            for (int jj = 0; jj < CLUSTER_SIZE / 2; ++jj) mk = box_screen_pair(rl, mp[jj], mk);
            uint32_t cand = ~mk << (32 - CLUSTER_SIZE);  // bit 31 = first member of the cluster
            while (cand != 0u) {
                const int z = __clz((int)cand);
                cand &= ~(0x80000000u >> z);
                const int member = (int)C.members[c * CLUSTER_SIZE + z];
                if (member < B.n) {  // never a padding slot
                    const uint32_t slot = __hip_atomic_fetch_add(&counts[ray], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    if (slot < (uint32_t)CAND_CAP) list[slot * 64 + ray] = (uint16_t)member;
                }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    cnt = counts[lane];
}

// Closest hit in [t_min, t_max] of one ray per lane against the boxes of the set; exact_ray(j, o, d, t_max, t) is the
// reference's test of box j for the ray (o, d): on a hit it returns a non-negative number the caller wants back with
// the winner (< 65536) and sets t (<= t_max), otherwise it returns a negative number.  Must be called by all 64
// lanes; `list` and `rows` as for the spheres.  exact_only must be the same for the whole wave.
template <class ExactRayFn>
__device__ __forceinline__ SetHit closest_box_hit(const BoxSetDev& B, d3 o, d3 d, double t_max, bool on, bool exact_only,
                                                  uint16_t* list, uint32_t* rows, int lane, ExactRayFn exact_ray) {
    SetHit h{t_max, -1};
    // member j against this lane's own ray, with the rule for equal t
    auto exact = [&](int j, SetHit& best) {
        double t;
        const int aux = exact_ray(j, o, d, best.t, t);
        if (aux >= 0) {
            best.t = t;
            best.id = j;
            best.aux = aux;
        }
    };
    if (exact_only) {  // validation: no float screen, every box in double
        if (on) {
            for (int j = 0; j < B.n; ++j) exact(j, h);
        }
        return h;
    }
    const BoxScreenRay sr = box_screen_prepare(o, d, B.cover);
    uint32_t cnt = (on && !sr.covered) ? (uint32_t)CAND_CAP + 1u : 0u;
    if (B.clusters.n_clusters > 0 && ballot(on && !sr.covered) == 0ull) {
        screen_box_clusters(B, sr, on, list, rows, lane, cnt);
    } else {
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
    }
    const bool overflow = cnt > (uint32_t)CAND_CAP;
    const uint32_t my_n = overflow ? 0u : cnt;
    // the candidates of the two-level walk come in no particular order: of two boxes hit at exactly the same t the
    // later one in the set wins, as it does when the reference tests them in order (see closest_hit_with)
    if (!spread_exact_tests(my_n, o, d, t_max, 0.0, list, rows, lane, h, [&](int j, d3 ro, d3 rd, double rt, double, double& t) { return exact_ray(j, ro, rd, rt, t); })) {
        for (uint32_t i = 0; ballot(i < my_n) != 0ull; ++i) {
            if (i < my_n) {
                const int j = (int)list[i * 64 + lane];
                SetHit one{h.t, -1};
                exact(j, one);
                if (one.id >= 0 && (one.t < h.t || j > h.id)) h = one;
            }
        }
    }
    if (ballot(overflow) != 0ull) {
        if (overflow) {  // more candidates than slots, or a ray the float bound does not cover: every box in double
            for (int j = 0; j < B.n; ++j) exact(j, h);
        }
    }
    return h;
}

}  // namespace rtd
