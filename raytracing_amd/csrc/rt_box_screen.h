// Closest hit of one ray per lane against a set of axis-aligned boxes held as a flat list: a float slab test of all
// boxes (wave-wide, box data in scalar registers) -- or, for a set with a two-level form, of the boxes of its clusters
// and then, item by item, of the members of the clusters a ray meets -- followed by the reference's double-precision
// test of the few candidates.  Used by the scene-graph kernels for BVH subtrees whose leaves are boxes.
#pragma once

#include "rt_sphere_screen.h"

namespace rtd {

#ifndef RT_SPREAD_BOX_TESTS
#define RT_SPREAD_BOX_TESTS 1
#endif

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

// Double-precision tests handed round the wave.  After the screens a few lanes hold several candidates each and most
// hold none (a set inside a scene graph: ~45 candidates on ~17 rays per visit of the wave, the busiest lane 5 or 6), and
// a loop in which every lane works through its own list runs as long as the busiest lane.  Here the (ray, box) pairs are
// numbered through the wave (prefix sum of the lanes' counts) and tested 64 at a time, one per lane, WHICHEVER lane the
// ray belongs to: the ray's origin, direction and t_max arrive by lane permutation.  Results meet per ray in LDS: the
// smallest t by a 64-bit unsigned minimum on its bit pattern (t >= t_min > 0: the patterns order like the values), then,
// among the tests that reached that t, the largest member index -- "t smaller, or equal and index larger", the rule of
// the per-lane loop below, which is the reference's own order of testing (see closest_hit_with).  Every test starts from
// the ray's own t_max instead of the closest hit so far: a box beyond the closest hit then reports a hit the minimum
// discards, where the loop would have had it rejected -- same winner, same t.
//   flat:  SPREAD_CAP words, the pairs (ray << 16 | member); then 2 SPREAD_CAP words, t of each test (all ones: no hit);
//          then SPREAD_CAP words, member << 16 | aux of each test   -- `rows`, whose queue of the cluster walk is done with
//   slots: 64 x 8 bytes smallest t per ray, then 64 words largest key per ray   -- `list`, once the pairs are copied out
constexpr int SPREAD_CAP = 256;
static_assert(4 * SPREAD_CAP <= ITEM_CAP / 2, "pairs and results of the spread tests use the queue's words");
static_assert(64 * 3 <= CAND_CAP * 32, "per-ray slots of the spread tests use the candidate lists' words");

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
    // the lanes' counts numbered through the wave
    uint32_t upto = my_n;  // inclusive prefix sum
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t below = (uint32_t)__shfl_up((int)upto, off);
        if (lane >= off) upto += below;
    }
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)upto, 63);
    const uint32_t busiest = (uint32_t)wave_max_u32(my_n);
    if (total > (uint32_t)SPREAD_CAP || busiest <= 1u || RT_SPREAD_BOX_TESTS == 0) {  // one round anyway, or too many for the words at hand
        for (uint32_t i = 0; ballot(i < my_n) != 0ull; ++i) {
            if (i < my_n) {
                const int j = (int)list[i * 64 + lane];
                SetHit one{h.t, -1};
                exact(j, one);
                if (one.id >= 0 && (one.t < h.t || j > h.id)) h = one;
            }
        }
    } else if (total != 0u) {
        uint32_t* flat = rows;
        uint64_t* res_t = (uint64_t*)(rows + SPREAD_CAP);
        uint32_t* res_key = rows + 3 * SPREAD_CAP;
        for (uint32_t i = 0; i < busiest; ++i)
            if (i < my_n) flat[upto - my_n + i] = ((uint32_t)lane << 16) | (uint32_t)list[i * 64 + lane];
        lds_wave_sync();
        uint64_t* slot_t = (uint64_t*)list;
        uint32_t* slot_key = (uint32_t*)list + 128;
        slot_t[lane] = ~0ull;
        slot_key[lane] = 0u;
        lds_wave_sync();
        for (uint32_t base = 0; base < total; base += 64u) {
            const bool live = base + (uint32_t)lane < total;
            const uint32_t pair = live ? flat[base + lane] : 0u;
            const int ray = (int)(pair >> 16), j = (int)(pair & 0xFFFFu);
            const d3 ro{lane_value_of(o.x, ray), lane_value_of(o.y, ray), lane_value_of(o.z, ray)};
            const d3 rd{lane_value_of(d.x, ray), lane_value_of(d.y, ray), lane_value_of(d.z, ray)};
            const double rt = lane_value_of(t_max, ray);
            if (live) {
                double t;
                const int aux = exact_ray(j, ro, rd, rt, t);
                const uint64_t bits = aux >= 0 ? (uint64_t)__double_as_longlong(t) : ~0ull;
                res_t[base + lane] = bits;
                res_key[base + lane] = ((uint32_t)j << 16) | (uint32_t)(aux & 0xFFFF);
                if (aux >= 0) __hip_atomic_fetch_min(&slot_t[ray], bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
        }
        lds_wave_sync();
        for (uint32_t base = 0; base < total; base += 64u) {
            if (base + (uint32_t)lane < total) {
                const int ray = (int)(flat[base + lane] >> 16);
                const uint64_t bits = res_t[base + lane];
                if (bits != ~0ull && bits == slot_t[ray])
                    __hip_atomic_fetch_max(&slot_key[ray], res_key[base + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
        }
        lds_wave_sync();
        if (slot_t[lane] != ~0ull) {
            h.t = __longlong_as_double((long long)slot_t[lane]);
            h.id = (int32_t)(slot_key[lane] >> 16);
            h.aux = (int32_t)(slot_key[lane] & 0xFFFFu);
        }
        lds_wave_sync();  // the lists' words are the caller's again
    }
    if (ballot(overflow) != 0ull) {
        if (overflow) {  // more candidates than slots, or a ray the float bound does not cover: every box in double
            for (int j = 0; j < B.n; ++j) exact(j, h);
        }
    }
    return h;
}

}  // namespace rtd
