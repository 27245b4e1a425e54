// Device-side building blocks shared by the gfx950 kernels: Philox4x32-10, the draw
// convention, and small double-precision vector helpers that keep the reference's
// operation order.  This translation unit family is compiled with -ffp-contract=off: every
// double expression below rounds exactly as the reference's x86-64 build does; fused
// multiply-adds appear only where written explicitly (the float screening pass).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "raytracing_amd.h"

namespace rtd {

// ---------------------------------------------------------------- random numbers
// Philox4x32-10 blocks (Salmon et al., SC'11; Random123 constants) with counter
// (pixel, sample, site, block index); key = the 64-bit seed.  Philox is a sixth of all vector
// instructions of a render kernel, so rejection loops use all four words of every block:
//   site 0       camera ray of the sample: block 0 = [jitter u, jitter v, lens x, lens y]; attempt
//                k >= 1 of the unit-disk rejection loop (B1/vec3.h:145-151) takes words 0,1 (k odd)
//                or 2,3 (k even) of block (k+1)/2; block TRY_TIME = [shutter time, -, -, -]
//   site 1+b     ray segment b: attempts 4g, 4g+1, 4g+2 of the unit-sphere rejection loop
//                (B1/vec3.h:114-120) take words 0,1,2 = x,y,z of blocks 3g, 3g+1, 3g+2, attempt 4g+3
//                takes the words 3 of those blocks (x, y, z in that order); a dielectric's Schlick
//                draw (B1/material.h:72) is word 0 of block 0; block TRY_MEDIUM+i = the draw of the
//                constant_medium with node index i (B2/constant_medium.h:48) made while
//                intersecting the segment.
// A draw is (word >> 1) * 2^-31: the 31-bit grid of rand()/(RAND_MAX+1.0) (B1/rtweekend.h:24-27).
constexpr uint32_t TRY_TIME = 0x80000000u;
constexpr uint32_t TRY_MEDIUM = 0x40000000u;

struct u32x4 {
    uint32_t x, y, z, w;
};

__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__gfx950__)
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#else
    return a ^ b ^ c;
#endif
}

__device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                               uint32_t k1) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(RT_PHILOX_OPAQUE_KEY)
    // (sphere-list kernels only: the scene-graph kernels, bound by latency rather than issue slots, lose 1-3 % to it)
    // The key is the same for the whole launch, so the compiler would work out the twenty round keys once and hold them
    // in scalar registers across the render loop -- more than there are: they end up spilled to vector lanes and read
    // back one by one in every round.  Made opaque here, the keys of a block are derived where they are used, by
    // scalar additions that cost no vector issue slot.
    asm volatile("" : "+s"(k0), "+s"(k1));
#endif
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        // (a ^ b ^ key as ONE instruction: gfx950's three-input bit operation with the truth table of the triple
        // exclusive or, 0x96.  With the key in a scalar register the compiler emits two v_xor_b32 for the plain
        // expression -- forty of a block's hundred vector instructions instead of twenty.)
        const uint32_t n0 = xor3((uint32_t)(p1 >> 32), c1, k0);
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = xor3((uint32_t)(p0 >> 32), c3, k1);
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0;
        c1 = n1;
        c2 = n2;
        c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return u32x4{c0, c1, c2, c3};
}

__device__ __forceinline__ double u01(uint32_t word) { return (double)(word >> 1) * 4.656612873077392578125e-10; }
// random_double(MIN, MAX), B1/rtweekend.h:37-40
__device__ __forceinline__ double u_range(uint32_t word, double lo, double hi) { return lo + (hi - lo) * u01(word); }

// random_double(-1, 1) = -1 + 2 * (k * 2^-31) for the draw's 31-bit integer k: every step is exact, and the value is
// (k - 2^30) * 2^-30 -- one subtraction, one conversion and one scaling instead of five operations.
__device__ __forceinline__ double u_signed(uint32_t word) {
#if defined(RT_ISSUE_BOUND) && RT_ISSUE_BOUND
    return (double)((int32_t)(word >> 1) - (1 << 30)) * 0x1p-30;
#else
    return u_range(word, -1, 1);
#endif
}

// ---------------------------------------------------------------- vec3 (B1/vec3.h)
struct d3 {
    double x, y, z;
};
__device__ __forceinline__ d3 operator+(d3 a, d3 b) { return d3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ d3 operator-(d3 a, d3 b) { return d3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ d3 operator*(d3 a, d3 b) { return d3{a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ d3 operator*(double t, d3 a) { return d3{t * a.x, t * a.y, t * a.z}; }
__device__ __forceinline__ d3 operator-(d3 a) { return d3{-a.x, -a.y, -a.z}; }
__device__ __forceinline__ double dot(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ double len2(d3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
// operator/(vec3, double) multiplies by the reciprocal (B1/vec3.h:96-98)
__device__ __forceinline__ d3 div_by(d3 a, double t) { return (1 / t) * a; }
__device__ __forceinline__ d3 unit_vector(d3 a) { return div_by(a, sqrt(len2(a))); }
__device__ __forceinline__ d3 reflect(d3 v, d3 n) { return v - (2 * dot(v, n)) * n; }  // B1/vec3.h:134-136

__device__ __forceinline__ uint64_t ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// ---------------------------------------------------------------- several quotients over one denominator
// An IEEE double division is eleven vector instructions on gfx950: the denominator's reciprocal (v_rcp_f64 and two
// Newton steps, five instructions), the quotient's estimate, its exact remainder and one correction (three), and three
// more that rescale operands with extreme exponents and patch the special cases (v_div_scale twice, v_div_fixup).
// Where several numerators meet the same denominator -- every root of a ray's sphere tests is (...) / |d|^2, both
// jitter coordinates of a sample are divided by W-1 and H-1 -- the reciprocal is worked out once (SharedDivisor) and a
// quotient costs the three instructions in the middle.  Same instructions on the same operands as the compiler's own
// sequence whenever that sequence rescales nothing and patches nothing, i.e. the result is the correctly rounded
// quotient bit for bit: denominator within [2^-60, 2^60], numerator zero or within [2^-200, 2^200] in magnitude (the
// hardware rescales for exponents beyond +-700 or so only).  Any other operand takes the division itself.
// RT_ISSUE_BOUND: set by the translation unit of the sphere-list kernels, which are bound by vector issue -- every
// instruction saved is time saved.  The scene-graph kernels wait on loads and are short of registers: the same
// rewrites change nothing there or cost (the final scene's variant loses 8 % to them), so their unit leaves it unset.
#ifndef RT_ISSUE_BOUND
#define RT_ISSUE_BOUND 0
#endif
#ifndef RT_SHARED_DIVISOR
#define RT_SHARED_DIVISOR 1
#endif
struct SharedDivisor {
    double a;   // the denominator
    double r;   // its reciprocal after two Newton steps, as the division sequence computes it
    bool tame;  // a within [2^-60, 2^60]
};
__device__ __forceinline__ SharedDivisor shared_divisor(double a) {
    SharedDivisor s;
    s.a = a;
    s.tame = a >= 0x1p-60 && a <= 0x1p60;
    double r = __builtin_amdgcn_rcp(a);
    r = __builtin_fma(r, __builtin_fma(-a, r, 1.0), r);
    r = __builtin_fma(r, __builtin_fma(-a, r, 1.0), r);
    s.r = r;
    return s;
}
// n / s.a for a numerator known to be +0 or within [2^-200, 2^200] and a tame denominator
__device__ __forceinline__ double quotient_of_tame(double n, double a, double r) {
    const double q = n * r;
    return __builtin_fma(__builtin_fma(-a, q, n), r, q);
}
// n / s.a
__device__ __forceinline__ double quotient(double n, const SharedDivisor& s) {
#if RT_SHARED_DIVISOR
    const double m = fabs(n);
    if (s.tame && m >= 0x1p-200 && m <= 0x1p200) return quotient_of_tame(n, s.a, s.r);
#endif
    return n / s.a;
}

// The same for square roots: the compiler's sqrt(double) is a reciprocal square root refined to the correctly rounded
// root (v_rsq_f64 and ten multiply-adds) wrapped in eight more instructions that rescale arguments below 2^-767 and pass
// zeros and infinities through.  For an argument within [2^-700, 2^700] the ten alone give the same bits.
__device__ __forceinline__ double sqrt_of_tame(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
    g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
    return g;
}
__device__ __forceinline__ double sqrt_shared(double x) {
#if RT_SHARED_DIVISOR
    if (x >= 0x1p-700 && x <= 0x1p700) return sqrt_of_tame(x);
#endif
    return sqrt(x);
}
// unit_vector(v) = (1 / sqrt(|v|^2)) * v (B1/vec3.h:96-98, 153-155): one range test covers root and reciprocal
// (all lanes of the wave that call it take the same way: one scalar branch)
__device__ __forceinline__ d3 unit_vector_shared(d3 v) {
#if RT_SHARED_DIVISOR
    const double l2 = len2(v);
    if (ballot(!(l2 >= 0x1p-120 && l2 <= 0x1p120)) == 0ull) {
        const double l = sqrt_of_tame(l2);
        return quotient_of_tame(1.0, l, shared_divisor(l).r) * v;
    }
#endif
    return unit_vector(v);
}

__device__ __forceinline__ d3 refract(d3 uv, d3 n, double etai_over_etat) {            // B1/vec3.h:138-143
    const double cos_theta = fmin(dot(-uv, n), 1.0);
    const d3 r_out_perp = etai_over_etat * (uv + cos_theta * n);
#if RT_ISSUE_BOUND
    const d3 r_out_parallel = (-sqrt_shared(fabs(1.0 - len2(r_out_perp)))) * n;
#else
    const d3 r_out_parallel = (-sqrt(fabs(1.0 - len2(r_out_perp)))) * n;
#endif
    return r_out_perp + r_out_parallel;
}

__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// What the lanes of this wave wrote to LDS before is visible to all of them after.
__device__ __forceinline__ void lds_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// A double of another lane: src_lane the same for the whole wave / different from lane to lane.
__device__ __forceinline__ double lane_value(double v, int src_lane) {
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, src_lane);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), src_lane);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
__device__ __forceinline__ double lane_value_of(double v, int src_lane) {
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(uint32_t)b);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(uint32_t)(b >> 32));
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {  // the same in every lane
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t other = (uint32_t)__shfl_xor((int)v, off);
        v = other > v ? other : v;
    }
    return v;
}
// number of set bits of `m` below this lane
__device__ __forceinline__ uint32_t rank_below(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// (time - time0) / (time1 - time0) of moving_sphere::center (B2/moving_sphere.h:28).  Over the interval [0, 1] -- the
// one every moving sphere of the reference has -- the quotient is `time` itself, bit for bit (time - 0 = time,
// 1 - 0 = 1, time / 1 = time), and the division is skipped unless some lane of the wave holds another interval.
__device__ __forceinline__ double motion_fraction(double time, double time0, double time1) {
    if (time0 == 0.0 && time1 == 1.0) return time;
    return (time - time0) / (time1 - time0);
}

// n / width for n < 2^28 with the multiplier and shift of make_width_divider (rt_internal.h)
__device__ __forceinline__ uint32_t div_by_width(uint32_t n, uint32_t mul, uint32_t shift) {
    return (uint32_t)(((uint64_t)n * (uint64_t)mul) >> shift);
}

constexpr double ACCUM_SCALE = (double)(1ull << RT_ACCUM_FRAC_BITS);

}  // namespace rtd
