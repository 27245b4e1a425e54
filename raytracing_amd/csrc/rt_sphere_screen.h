// Closest hit of one ray per lane against a set of spheres held as a flat list: a float screening pass over all
// spheres (wave-wide, sphere data in scalar registers) followed by the reference's double-precision test of the few
// candidates.  Shared by the sphere-list kernels (the whole world is one set) and the scene-graph kernels (a BVH
// subtree that holds only spheres is one set).
#pragma once
#include <type_traits>

#include "rt_device_common.h"
#include "rt_internal.h"
#include "rt_screen_ops.h"

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
    __device__ __forceinline__ void add(int k, uint64_t v) { sec[k] += v; }
};
#else
struct ProbeClock {
    __device__ __forceinline__ void start() {}
    __device__ __forceinline__ void mark(int) {}
    __device__ __forceinline__ void add(int, uint64_t) {}
};
#endif
enum { SEC_REFILL = 0, SEC_SCREEN_DENSE, SEC_SCREEN_SPARSE, SEC_EXACT, SEC_SHADE, SEC_PREP, SEC_SCREEN_MEMBERS };

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
#ifndef RT_SPARSE_MAX_RAYS_TWO_LEVEL
#define RT_SPARSE_MAX_RAYS_TWO_LEVEL 8
#endif
constexpr int SPARSE_MAX_RAYS_TWO_LEVEL = RT_SPARSE_MAX_RAYS_TWO_LEVEL;  // sets with a two-level form: its walk is ~3x cheaper
constexpr int ROW_WORDS = 17;        // 16 mask words per ray + 1 pad (bank spread)
constexpr int SPARSE_CHUNK_PAIRS = 256;

// ---------------------------------------------------------------- dense walks
// All lanes x all n spheres of the pair image `image` (wave-uniform scalar loads; the image is padded to whole pairs
// and, past them, readable up to a whole group).  Candidates are appended to the lane's list as set members:
// position p is member p, or map[p] when MAPPED.  n_set = size of the set the members belong to.
template <bool MAPPED>
__device__ __forceinline__ void screen_all(const ScreenRay& sr, const SpherePair* image, const uint16_t* map, int n, int n_set,
                                           bool on, uint16_t* list, int lane, uint32_t& cnt) {
    // The screen image is immutable for the whole launch: address it through the constant
    // address space so that the wave-uniform reads become scalar loads (SGPR operands).
    const ConstSpherePair* sp = (const ConstSpherePair*)image;
    const int n_full_groups = n / SCREEN_GROUP, n_tail_pairs = (n % SCREEN_GROUP + 1) / 2;
    const int n_groups = n_full_groups + (n_tail_pairs > 0 ? 1 : 0);
    for (int g = 0; g < n_groups; ++g) {
        uint32_t mask = 0;
        if (g < n_full_groups) {
#pragma unroll
            for (int jj = 0; jj < SCREEN_GROUP / 2; ++jj) mask = screen_pair(sr, sp + jj, mask);
            sp += SCREEN_GROUP / 2;
        } else {  // last, partial group: bits of absent spheres read "no candidate"
            mask = ~0u;
#pragma unroll 1
            for (int jj = 0; jj < n_tail_pairs; ++jj) mask = screen_pair(sr, sp + jj, mask);
            mask = ~(~mask << (SCREEN_GROUP - 2 * n_tail_pairs));
        }
        uint32_t cand = on ? ~mask : 0u;
        while (ballot(cand != 0u) != 0ull) {
            if (cand != 0u) {
                const int z = __clz((int)cand);
                cand &= ~(0x80000000u >> z);
                const int pos = g * SCREEN_GROUP + z;
                if (pos < n) {  // the padded half of an odd last pair is never a candidate
                    const int member = MAPPED ? (int)map[pos] : pos;
                    if (member < n_set) {
                        if (cnt < (uint32_t)CAND_CAP) list[cnt * 64 + lane] = (uint16_t)member;
                        ++cnt;
                    }
                }
            }
        }
    }
}

// Two-level walk (SphereClustersDev): the wide spheres and the clusters' boxes wave-wide; every (ray, cluster) meeting
// becomes an item of a wave-wide queue in LDS, and the items are then screened 64 at a time, one per lane, whichever
// lane the ray belongs to -- the rays' screening constants travel by lane permutation, candidates go to the owning
// ray's list through a counter in LDS.  (Left to its own ray alone, every lane would wait for the ray that meets the
// most clusters: ~4 rounds against ~2 on the One-Weekend final scene.)  Every lane that is `on` must be covered by the
// box margins (br.covered).  `rows` = the wave's ROW_WORDS*64 words: the queue, then the 64 counters.  A wave whose
// queue is full sends the rays that still have clusters to note to the test of every member in double.  Candidates
// arrive in no particular order.
constexpr int ITEM_CAP = 2048;         // items per wave and walk
constexpr int ITEM_CLUSTER_BITS = 10;  // an item = lane << 10 | cluster: at most 1024 clusters
static_assert(ITEM_CAP * 2 + 64 * 4 <= ROW_WORDS * 64 * 4, "the queue and the counters share the rows of the sparse walk");

__device__ __forceinline__ f2 splat_from_lane_of(f2 v, int src_lane) {  // src_lane may differ from lane to lane
    const float f = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)__float_as_uint(v.x)));
    return f2{f, f};
}

// First level of a two-level walk (queue_cluster_items below) for a wave in which few lanes hold a ray -- a set inside
// a scene graph is reached by a handful of rays more often than not.  The wave turns by 90 degrees: lane t holds the
// box of cluster t (64 clusters at a time), the rays are taken one after the other, their constants broadcast from the
// owning lane, and the ballot of the lanes whose box the ray meets is the ray's list of clusters.  About 35 vector
// instructions per ray and 64 clusters, against 19 per pair of clusters for all rays at once.  Same arithmetic as
// box_screen_pair_ops, one box per lane.
#ifndef RT_SPARSE_CLUSTER_MAX_RAYS
#define RT_SPARSE_CLUSTER_MAX_RAYS 12
#endif
__device__ __forceinline__ float float_from_lane(float v, int src_lane) {
    return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), src_lane));
}
__device__ __forceinline__ uint32_t queue_cluster_items_few_rays(const BoxPair* boxes, int n_clusters, const BoxScreenRay& br, uint64_t on_mask,
                                                                 uint16_t* items, int lane, bool& dropped) {
    uint32_t n_items = 0;
    for (int first = 0; first < n_clusters; first += 64) {
        const int c = first + lane;
        const bool present = c < n_clusters;
        const BoxPair* bp = boxes + ((present ? c : 0) >> 1);
        const int half = c & 1;
        const float mid[3] = {bp->mid[0][half], bp->mid[1][half], bp->mid[2][half]};
        const float hw[3] = {bp->hw[0][half], bp->hw[1][half], bp->hw[2][half]};
        for (uint64_t m = on_mask; m != 0ull; m &= m - 1ull) {
            const int src = __builtin_ctzll(m);
            float near_t[3], far_t[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float inv = float_from_lane(br.inv[k].x, src), ainv = float_from_lane(br.ainv[k].x, src);
                const float tc = __builtin_fmaf(mid[k], inv, float_from_lane(br.c[k].x, src));
                near_t[k] = __builtin_fmaf(hw[k], -ainv, tc);
                far_t[k] = __builtin_fmaf(hw[k], ainv, tc);
            }
            const float far_all = fminf(fminf(far_t[0], far_t[1]), far_t[2]);
            const float near_all = fmaxf(fmaxf(near_t[0], near_t[1]), near_t[2]);
            const float q = fminf(far_all - near_all, far_all);
            const bool meets = present && (__float_as_uint(q) >> 31) == 0u;
            const uint64_t met = ballot(meets);
            const uint32_t n_met = (uint32_t)__popcll(met);
            if (n_items + n_met <= (uint32_t)ITEM_CAP) {
                if (meets) {
                    const uint32_t before_me = __builtin_amdgcn_mbcnt_hi((uint32_t)(met >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)met, 0u));
                    items[n_items + before_me] = (uint16_t)((src << ITEM_CLUSTER_BITS) | c);
                }
                n_items += n_met;
            } else if (n_met != 0u && lane == src) {
                dropped = true;
            }
        }
    }
    return n_items;
}

// First level of a two-level walk: the wave tests every lane's ray against the boxes of all clusters (`boxes`, padded
// to whole groups of SCREEN_GROUP) and queues one item, lane << ITEM_CLUSTER_BITS | cluster, per meeting in `items`
// (ITEM_CAP slots; positions from the ballot, no atomics).  Returns the number of items; `dropped` is set in lanes that
// still had clusters to note when the queue was full.
__device__ __forceinline__ uint32_t queue_cluster_items(const BoxPair* boxes, int n_clusters, const BoxScreenRay& br, bool on,
                                                        uint16_t* items, int lane, bool& dropped) {
    uint32_t n_items = 0;  // the same in every lane
    dropped = false;
    {
        const uint64_t on_mask = ballot(on);
        if (__popcll(on_mask) <= RT_SPARSE_CLUSTER_MAX_RAYS) return queue_cluster_items_few_rays(boxes, n_clusters, br, on_mask, items, lane, dropped);
    }
    const ConstBoxPair* bp = (const ConstBoxPair*)boxes;
    const int n_groups = (n_clusters + SCREEN_GROUP - 1) / SCREEN_GROUP;
    for (int g = 0; g < n_groups; ++g) {
        uint32_t mask = 0;
#pragma unroll
        for (int jj = 0; jj < SCREEN_GROUP / 2; ++jj) mask = box_screen_pair(br, bp + jj, mask);
        bp += SCREEN_GROUP / 2;
        const int present = n_clusters - g * SCREEN_GROUP;  // the image is padded to whole groups
        uint32_t cand = on ? ~mask : 0u;
        if (present < SCREEN_GROUP) cand &= ~(0xFFFFFFFFu >> present);
        for (uint64_t busy = ballot(cand != 0u); busy != 0ull; busy = ballot(cand != 0u)) {
            const bool room = n_items + 64u <= (uint32_t)ITEM_CAP;
            if (cand != 0u) {
                const int z = __clz((int)cand);
                cand &= ~(0x80000000u >> z);
                const uint32_t before_me = __builtin_amdgcn_mbcnt_hi((uint32_t)(busy >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)busy, 0u));
                if (room) items[n_items + before_me] = (uint16_t)((lane << ITEM_CLUSTER_BITS) | (g * SCREEN_GROUP + z));
                else dropped = true;
            }
            if (room) n_items += (uint32_t)__popcll(busy);
        }
    }
    return n_items;
}

__device__ __forceinline__ void screen_clusters(const SphereSetDev& S, const ScreenRay& sr, const BoxScreenRay& br, bool on,
                                                uint16_t* list, uint32_t* rows, int lane, uint32_t& cnt, ProbeClock& pc) {
    const SphereClustersDev& C = S.clusters;
    uint16_t* items = (uint16_t*)rows;
    uint32_t* counts = rows + ITEM_CAP / 2;
    bool dropped;
    const uint32_t n_items = queue_cluster_items(C.boxes, C.n_clusters, br, on, items, lane, dropped);
    if (dropped) cnt = (uint32_t)CAND_CAP + 1u;
    if (C.n_wide > 0) screen_all<true>(sr, C.wide_pairs, C.wide_members, C.n_wide, S.n, on, list, lane, cnt);
    counts[lane] = cnt;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    pc.mark(SEC_SCREEN_DENSE);
    for (uint32_t base = 0; base < n_items; base += 64u) {
        pc.add(7, 1);
        const bool live = base + (uint32_t)lane < n_items;
        const uint32_t item = live ? (uint32_t)items[base + lane] : 0u;
        const int ray = (int)(item >> ITEM_CLUSTER_BITS), c = (int)(item & ((1u << ITEM_CLUSTER_BITS) - 1u));
        ScreenRay rl;  // the constants of the item's ray (all lanes take part in the permutations)
        rl.ndx = splat_from_lane_of(sr.ndx, ray);
        rl.ndy = splat_from_lane_of(sr.ndy, ray);
        rl.ndz = splat_from_lane_of(sr.ndz, ray);
        rl.p2ox = splat_from_lane_of(sr.p2ox, ray);
        rl.p2oy = splat_from_lane_of(sr.p2oy, ray);
        rl.p2oz = splat_from_lane_of(sr.p2oz, ray);
        rl.od = splat_from_lane_of(sr.od, ray);
        rl.e = splat_from_lane_of(sr.e, ray);
        if (live) {
            const SpherePair* mp = C.pairs + (size_t)c * (CLUSTER_SIZE / 2);
            uint32_t mk = 0;
#pragma unroll
            for (int jj = 0; jj < CLUSTER_SIZE / 2; ++jj) mk = screen_pair(rl, mp[jj], mk);
            uint32_t cand = ~mk << (32 - CLUSTER_SIZE);  // bit 31 = first member of the cluster
            while (cand != 0u) {
                const int z = __clz((int)cand);
                cand &= ~(0x80000000u >> z);
                const int member = (int)C.members[c * CLUSTER_SIZE + z];
                if (member < S.n) {  // never a padding slot, whatever the arithmetic said
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

// ---------------------------------------------------------------- exact pass
struct SetHit {
    double t;
    int32_t id;
    int32_t aux = -1;  // whatever the caller's exact test wants to know about the winner (the scene graph: its node)
};

// sphere::hit up to the accepted root, B1/sphere.h:24-41, inside the closest-so-far loop of
// hittable_list::hit, B1/hittable_list.h:31-37.  Comparisons keep the reference's form so
// that NaNs fall the same way.
// Centre of sphere j at time tm: moving_sphere::center, B2/moving_sphere.h:27-29.
template <bool TIMED>
__device__ __forceinline__ d3 center_at(const SphereSetDev& S, int j, const SphereExact& e, double tm) {
    d3 center{e.cx, e.cy, e.cz};
    if constexpr (TIMED) {
        const SphereMotion* mv = S.motion + j;
        const d3 travel{mv->vx, mv->vy, mv->vz};
        const int64_t moving = mv->moving;
        if (moving != 0) {
            // (time - time0) / (time1 - time0): over [0, 1] -- the interval every moving sphere of the reference has -- the
            // quotient is `time` itself, bit for bit (time - 0 = time, 1 - 0 = 1, time / 1 = time)
            double fraction = tm;
            if (moving == 2) fraction = (tm - mv->time0) / (mv->time1 - mv->time0);
            center = center + fraction * travel;
        }
    }
    return center;
}

// A ray whose roots may take the short division and square root (rt_device_common.h) for every sphere the float screen
// holds (coordinates and radii of at most 1e15 < 2^50): |d|^2 within [2^-60, 2^60] and origin coordinates of at most 2^60
// bound both numerators, -half_b -+ sqrt(discriminant), by 2^94 and the discriminant by 2^190, so nothing is rescaled
// from above.  From below the division needs no bound here: a numerator under 2^-200 gives a root under 2^-140 either
// way, which t_min (1e-3 in every caller) rejects, as it does the root the division itself would give; the square root
// keeps its own test of the discriminant's lower end.
struct RootsOf {
    SharedDivisor over_a;  // |d|^2
    bool fast;             // the whole wave's rays are of that kind (wave-uniform: one scalar branch per use)
};
__device__ __forceinline__ RootsOf roots_of(d3 o, d3 d, bool on, bool allow) {
    RootsOf r;
    r.over_a = shared_divisor(len2(d));
    const bool tame = r.over_a.tame && fabs(o.x) <= 0x1p60 && fabs(o.y) <= 0x1p60 && fabs(o.z) <= 0x1p60;
    r.fast = allow && RT_SHARED_DIVISOR != 0 && ballot(on && !tame) == 0ull;
    return r;
}

template <bool TIMED>
__device__ __forceinline__ void exact_one(const SphereSetDev& S, int j, d3 o, d3 d, const RootsOf& ray, double tm, double t_min, SetHit& h) {
    const double a = ray.over_a.a;  // |d|^2, the denominator of both roots (and of every other member's roots)
    const SphereExact e = S.exact[j];
    const d3 oc = o - center_at<TIMED>(S, j, e, tm);
    const double half_b = dot(oc, d);
    const double c = len2(oc) - e.radius * e.radius;
    const double discriminant = half_b * half_b - a * c;
    if (ray.fast) {
        double sqrtd;
        if (discriminant >= 0x1p-700) {
            sqrtd = sqrt_of_tame(discriminant);
        } else {
            if (discriminant < 0) return;
            sqrtd = sqrt(discriminant);  // zero, or next to nothing (or a NaN)
        }
        double root = quotient_of_tame(-half_b - sqrtd, a, ray.over_a.r);
        if (root < t_min || root > h.t) {
            root = quotient_of_tame(-half_b + sqrtd, a, ray.over_a.r);
            if (root < t_min || root > h.t) return;
        }
        h.t = root;
        h.id = j;
        return;
    }
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

// Double-precision tests handed round the wave.  After the screens a few lanes hold several candidates each and most
// hold none (a set of boxes inside a scene graph: ~45 candidates on ~17 rays per visit of the wave, the busiest lane 5 or 6; a
// dense set of spheres: ~80 on 5 rays), and
// a loop in which every lane works through its own list runs as long as the busiest lane.  Here the (ray, member) pairs are
// numbered through the wave (prefix sum of the lanes' counts) and tested 64 at a time, one per lane, WHICHEVER lane the
// ray belongs to: the ray's origin, direction and t_max arrive by lane permutation.  Results meet per ray in LDS: the
// smallest t by a 64-bit unsigned minimum on its bit pattern (t >= t_min > 0: the patterns order like the values), then,
// among the tests that reached that t, the largest member index -- "t smaller, or equal and index larger", the rule of
// the per-lane loop below, which is the reference's own order of testing (see closest_hit_with).  Every test starts from
// the ray's own t_max instead of the closest hit so far: a member beyond the closest hit then reports a hit the minimum
// discards, where the loop would have had it rejected -- same winner, same t.
//   flat:  SPREAD_CAP words, the pairs (ray << 16 | member); then 2 SPREAD_CAP words, t of each test (all ones: no hit);
//          then SPREAD_CAP words, member << 16 | aux of each test   -- `rows`, whose queue of the cluster walk is done with
//   slots: 64 x 8 bytes smallest t per ray, then 64 words largest key per ray   -- `list`, once the pairs are copied out
#ifndef RT_SPREAD_CAP
#define RT_SPREAD_CAP 256
#endif
constexpr int SPREAD_CAP = RT_SPREAD_CAP;
static_assert(4 * SPREAD_CAP <= ITEM_CAP / 2, "pairs and results of the spread tests use the queue's words");
// spread_exact_tests reads and writes `rows + SPREAD_CAP` and `list` as 64-bit words (ds_min_u64): the callers declare
// both arrays 16-byte aligned, and every wave's slice and the offset inside it must be whole 8-byte words
static_assert((SPREAD_CAP * 4) % 8 == 0 && (ROW_WORDS * 64 * 4) % 8 == 0 && (CAND_CAP * 64 * 2) % 8 == 0,
              "64-bit LDS operations need 8-byte aligned slices");
static_assert(64 * 3 <= CAND_CAP * 32, "per-ray slots of the spread tests use the candidate lists' words");

#ifndef RT_SPREAD_EXACT_TESTS
#define RT_SPREAD_EXACT_TESTS 1
#endif
// my_n = this lane's candidates (list[i * 64 + lane], i < my_n); exact_ray(j, o, d, t_max, time, t): the reference's test
// of member j for the ray (o, d, time) -- on a hit a non-negative number the caller wants back with the winner (< 65536) and t
// (<= t_max), otherwise a negative number.  Returns false, with nothing done, when the per-lane loop is as good (no lane
// has two candidates) or the pairs do not fit; otherwise h holds the winner of the lanes that have one.  All 64 lanes
// must call it; `list` and `rows` are overwritten.
template <class ExactRayFn>
__device__ __forceinline__ bool spread_exact_tests(uint32_t my_n, d3 o, d3 d, double t_max, double ray_time, uint16_t* list, uint32_t* rows,
                                                   int lane, SetHit& h, ExactRayFn exact_ray) {
    uint32_t upto = my_n;  // inclusive prefix sum: the lanes' counts numbered through the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t below = (uint32_t)__shfl_up((int)upto, off);
        if (lane >= off) upto += below;
    }
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)upto, 63);
    const uint32_t busiest = wave_max_u32(my_n);
    if (RT_SPREAD_EXACT_TESTS == 0 || total > (uint32_t)SPREAD_CAP || busiest <= 1u) return false;
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
        const double rt = lane_value_of(t_max, ray), rtm = lane_value_of(ray_time, ray);
        if (live) {
            double t;
            const int aux = exact_ray(j, ro, rd, rt, rtm, t);
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
    lds_wave_sync();  // the words are the caller's again
    return true;
}

// Closest hit in [t_min, t_max] of one ray per lane against the whole set; id = -1 if there is none.  Must be called
// by all 64 lanes of the wave (wave-level loops); lanes with on == false take no part in the result.
// `list` = this wave's CAND_CAP*64 uint16 slots in LDS, `rows` = its ROW_WORDS*64 words.
// exact(j, h) is the double-precision test of member j: it lowers h.t and sets h.id = j if j is hit in [t_min, h.t].
// The screening image holds, per member, a sphere that the member lies inside.
// exact_ray: the test in the form spread_exact_tests wants, or NoSpread{} (every lane works through its own candidates:
// the sphere-list kernels, whose lanes all hold a few).
struct NoSpread {};
template <class ExactFn, class ExactRayFn = NoSpread>
__device__ __forceinline__ SetHit closest_hit_with(const SphereSetDev& S, d3 o, d3 d, double t_max, bool on, bool exact_only,
                                                   uint16_t* list, uint32_t* rows, int lane, uint32_t& overflow_count,
                                                   ProbeClock& pc, ExactFn exact, ExactRayFn exact_ray = {}, double ray_time = 0.0) {
    const double a = len2(d);
    SetHit h{t_max, -1};

    if (exact_only) {
        if (on) {
            for (int j = 0; j < S.n; ++j) exact(j, h);
        }
        return h;
    }

    const ScreenRay sr = screen_prepare(o, d, a);
    // The float screen's error bound needs a ray it can hold: origin coordinates of at most 1e15 (like the spheres':
    // no float product overflows) and a direction of length 1e-100 to 1e100 (no product of the reference's own test
    // overflows or underflows either).  Any other ray -- a NaN or an infinity (book 3's light sampling produces them: the reference drops such samples at the end), or values
    // so large that the reference's own products overflow -- has no meaningful screen: a NaN's sign is arbitrary, so it
    // could name padding entries as candidates and drop real ones.  The reference's comparisons tend to come out false
    // for such a ray, which makes every object it tests a "hit" -- so the lane tests every member in double, in
    // order, with the reference's own arithmetic.
    const bool finite_ray = fabs(o.x) <= 1e15 && fabs(o.y) <= 1e15 && fabs(o.z) <= 1e15 && a >= 1e-200 && a <= 1e200;
    uint32_t cnt = (on && !finite_ray) ? (uint32_t)CAND_CAP + 1u : 0u;
    on = on && finite_ray;  // such lanes take no part in the screen; the overflow pass below serves them
    const uint64_t on_mask = ballot(on);
    const bool two_level = S.clusters.n_clusters > 0;
    BoxScreenRay br;
    br.covered = false;
    if (two_level) br = box_screen_prepare(o, d, S.clusters.cover);
    pc.mark(SEC_PREP);
    const bool by_clusters = two_level && ballot(on && !br.covered) == 0ull;
    if (!by_clusters && __popcll(on_mask) <= (two_level ? SPARSE_MAX_RAYS_TWO_LEVEL : SPARSE_MAX_RAYS)) {
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
    } else if (by_clusters) {
        screen_clusters(S, sr, br, on, list, rows, lane, cnt, pc);
        pc.mark(SEC_SCREEN_MEMBERS);
    } else {
        screen_all<false>(sr, S.screen, nullptr, S.n, S.n, on, list, lane, cnt);
        pc.mark(SEC_SCREEN_DENSE);
    }

    const bool overflow = cnt > (uint32_t)CAND_CAP;
    const uint32_t my_n = overflow ? 0u : cnt;
    // The reference tests the members in list order and accepts a root equal to the closest so far, so of two members
    // hit at exactly the same t the later one wins.  The candidates of the two-level walk come in no particular order:
    // the same rule stated on the indices gives the same winner for any order.
    bool spread = false;
    if constexpr (!std::is_same<ExactRayFn, NoSpread>::value) spread = spread_exact_tests(my_n, o, d, t_max, ray_time, list, rows, lane, h, exact_ray);
    if (!spread) {
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
        if (overflow) {  // more candidates than slots: test the whole list in double
            ++overflow_count;
            for (int j = 0; j < S.n; ++j) exact(j, h);
        }
    }
    pc.mark(SEC_EXACT);
    return h;
}

// The set's members are its spheres themselves (sphere-list kernels).  RT_SPREAD_LIST_TESTS: their double-precision tests
// handed round the wave as well.
#ifndef RT_TIMED_SHARED_DIVISOR
#define RT_TIMED_SHARED_DIVISOR 1
#endif
#ifndef RT_SPREAD_LIST_TESTS
#define RT_SPREAD_LIST_TESTS 0
#endif
template <bool TIMED>
__device__ __forceinline__ SetHit closest_hit(const SphereSetDev& S, d3 o, d3 d, double tm, double t_max, bool on, bool exact_only,
                                              uint16_t* list, uint32_t* rows, int lane, uint32_t& overflow_count,
                                              ProbeClock& pc) {
    const double t_min = 0.001;  // B1/RayTracing.cpp:17
    // (exact_only: the set may hold spheres beyond the float screen's range, which the bounds of roots_of do not cover)
    const RootsOf a = roots_of(o, d, on, !exact_only && (!TIMED || RT_TIMED_SHARED_DIVISOR != 0));
#if RT_SPREAD_LIST_TESTS
    return closest_hit_with(
        S, o, d, t_max, on, exact_only, list, rows, lane, overflow_count, pc,
        [&](int j, SetHit& h) { exact_one<TIMED>(S, j, o, d, a, tm, t_min, h); },
        [&](int j, d3 ro, d3 rd, double rt, double rtm, double& t) -> int {
            SetHit one{rt, -1};
            exact_one<TIMED>(S, j, ro, rd, roots_of(ro, rd, true, false), rtm, t_min, one);
            t = one.t;
            return one.id >= 0 ? 0 : -1;
        },
        tm);
#else
    return closest_hit_with(S, o, d, t_max, on, exact_only, list, rows, lane, overflow_count, pc,
                            [&](int j, SetHit& h) { exact_one<TIMED>(S, j, o, d, a, tm, t_min, h); });
#endif
}

}  // namespace rtd
