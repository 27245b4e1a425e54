// The short forms of division, square root and the unit-sphere test that the sphere-list kernels use
// (raytracing_amd/csrc/rt_device_common.h, rt_render_common.h) against the operations they stand for, on the device:
//   quotient_of_tame(n, a, shared_divisor(a).r)   vs  n / a          a in [2^-60, 2^60], |n| in [2^-200, 2^200], and n = 0
//   sqrt_of_tame(x)                               vs  sqrt(x)        x in [2^-700, 2^700]
//   inside_unit_sphere(x, y, z)                   vs  !(len2(p) >= 1), p = u_range(word, -1, 1)
//   u_signed(word)                                vs  u_range(word, -1, 1)   for every 31-bit draw
// Operands: random mantissas with exponents spread over the whole allowed range, plus the ends of the ranges, and for
// the sphere test random words plus triples searched for whose |p|^2 lands within about 2^10 of 2^60 (where the
// integer test hands over to the doubles).  Prints the number of results that differ bit for bit (expected: 0 everywhere), and checks
// the host's own n / a and sqrt(x) (glibc, correctly rounded) against the device's as well.
//   /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I../../include -I../../raytracing_amd/csrc short_roots.hip -o short_roots
#define RT_ISSUE_BOUND 1
#include "rt_render_common.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

using namespace rtd;

__global__ void k_div(int n, const double* num, const double* den, double* fast, double* slow) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fast[i] = quotient_of_tame(num[i], den[i], shared_divisor(den[i]).r);
    slow[i] = num[i] / den[i];
}
__global__ void k_sqrt(int n, const double* x, double* fast, double* slow) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fast[i] = sqrt_of_tame(x[i]);
    slow[i] = sqrt(x[i]);
}
__global__ void k_unit(int n, const double* x, double* fast, double* slow) {  // x, y, z of vector i at 3i..3i+2
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const d3 v{x[3 * i], x[3 * i + 1], x[3 * i + 2]};
    const d3 a = unit_vector_shared(v), b = unit_vector(v);
    fast[3 * i] = a.x; fast[3 * i + 1] = a.y; fast[3 * i + 2] = a.z;
    slow[3 * i] = b.x; slow[3 * i + 1] = b.y; slow[3 * i + 2] = b.z;
}
__global__ void k_sphere(int n, const uint32_t* w, uint8_t* fast, uint8_t* slow) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t x = w[3 * i], y = w[3 * i + 1], z = w[3 * i + 2];
    fast[i] = inside_unit_sphere(x, y, z) ? 1 : 0;
    const d3 p{u_range(x, -1, 1), u_range(y, -1, 1), u_range(z, -1, 1)};
    slow[i] = !(len2(p) >= 1) ? 1 : 0;
}

__global__ void k_signed(uint32_t first, unsigned long long* differing) {  // all 2^31 draws, 2^8 per thread
    const uint32_t base = first + (blockIdx.x * blockDim.x + threadIdx.x) * 256u;
    unsigned long long d = 0;
    for (uint32_t i = 0; i < 256u; ++i) {
        const uint32_t word = (base + i) << 1;
        const double a = u_signed(word), b = u_range(word, -1, 1), c = u_signed(word | 1u);
        d += (__double_as_longlong(a) != __double_as_longlong(b)) + (__double_as_longlong(a) != __double_as_longlong(c));
    }
    if (d) atomicAdd(differing, d);
}

template <class T>
static T* to_dev(const std::vector<T>& v) {
    T* p = nullptr;
    (void)hipMalloc(&p, v.size() * sizeof(T));
    (void)hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    return p;
}
template <class T>
static std::vector<T> from_dev(const T* p, size_t n) {
    std::vector<T> v(n);
    (void)hipMemcpy(v.data(), p, n * sizeof(T), hipMemcpyDeviceToHost);
    return v;
}
static long long differing(const std::vector<double>& a, const std::vector<double>& b) {
    long long d = 0;
    for (size_t i = 0; i < a.size(); ++i) d += std::memcmp(&a[i], &b[i], 8) != 0;
    return d;
}

int main() {
    const int n = 1 << 24;
    std::mt19937_64 g(11);
    auto mantissa = [&]() { return 1.0 + (double)(g() >> 12) * 0x1p-52; };
    auto spread = [&](int lo, int hi) { return std::ldexp(mantissa(), lo + (int)(g() % (uint64_t)(hi - lo))); };
    int bad = 0;

    {   // u_signed against u_range(word, -1, 1), every 31-bit draw
        unsigned long long* d;
        (void)hipMalloc(&d, 8);
        (void)hipMemset(d, 0, 8);
        hipLaunchKernelGGL(k_signed, dim3((1u << 31) / 256u / 256u), dim3(256), 0, 0, 0u, d);
        const auto v = from_dev(d, 1);
        std::printf("draw in (-1,1) all 2^31 draws: u_signed differs from u_range(word, -1, 1) in %llu\n", v[0]);
        bad += v[0] != 0;
    }
    {   // division
        std::vector<double> num(n), den(n);
        for (int i = 0; i < n; ++i) {
            den[i] = spread(-60, 60);
            num[i] = (g() & 1 ? -1.0 : 1.0) * spread(-200, 200);
        }
        const double ends_d[4] = {0x1p-60, 0x1p60, 1.0, std::nextafter(2.0, 1.0)};
        const double ends_n[6] = {0.0, 0x1p-200, 0x1p200, -0x1p200, std::nextafter(0x1p-199, 0.0), 1.0};
        for (int i = 0; i < 24; ++i) {
            den[i] = ends_d[i % 4];
            num[i] = ends_n[i / 4];
        }
        double *dn = to_dev(num), *dd = to_dev(den), *f, *s;
        (void)hipMalloc(&f, n * 8); (void)hipMalloc(&s, n * 8);
        hipLaunchKernelGGL(k_div, dim3(n / 256), dim3(256), 0, 0, n, dn, dd, f, s);
        const auto vf = from_dev(f, n), vs = from_dev(s, n);
        std::vector<double> host(n);
        for (int i = 0; i < n; ++i) host[i] = num[i] / den[i];
        const long long a = differing(vf, vs), b = differing(vs, host);
        std::printf("division      %d operand pairs: short form differs from n / a in %lld, the device's n / a from the host's in %lld\n", n, a, b);
        bad += a != 0 || b != 0;
    }
    {   // square root
        std::vector<double> x(n);
        for (int i = 0; i < n; ++i) x[i] = spread(-700, 700);
        x[0] = 0x1p-700; x[1] = 0x1p700; x[2] = 1.0; x[3] = std::nextafter(4.0, 0.0); x[4] = std::nextafter(1.0, 2.0);
        double *dx = to_dev(x), *f, *s;
        (void)hipMalloc(&f, n * 8); (void)hipMalloc(&s, n * 8);
        hipLaunchKernelGGL(k_sqrt, dim3(n / 256), dim3(256), 0, 0, n, dx, f, s);
        const auto vf = from_dev(f, n), vs = from_dev(s, n);
        std::vector<double> host(n);
        for (int i = 0; i < n; ++i) host[i] = std::sqrt(x[i]);
        const long long a = differing(vf, vs), b = differing(vs, host);
        std::printf("square root   %d arguments: short form differs from sqrt(x) in %lld, the device's sqrt(x) from the host's in %lld\n", n, a, b);
        bad += a != 0 || b != 0;
    }
    {   // unit vector (range test for the whole wave, root and reciprocal)
        const int m = n / 4;
        std::vector<double> x(3 * (size_t)m);
        for (int i = 0; i < m; ++i) {
            const int e = (i & 1023) < 1000 ? (int)(g() % 100) - 50 : (int)(g() % 600) - 300;  // some waves leave the range
            for (int c = 0; c < 3; ++c) x[3 * (size_t)i + c] = (g() & 1 ? -1.0 : 1.0) * std::ldexp(mantissa(), e - (int)(g() % 3));
        }
        double *dx = to_dev(x), *f, *s;
        (void)hipMalloc(&f, 3 * (size_t)m * 8); (void)hipMalloc(&s, 3 * (size_t)m * 8);
        hipLaunchKernelGGL(k_unit, dim3(m / 256), dim3(256), 0, 0, m, dx, f, s);
        const long long a = differing(from_dev(f, 3 * (size_t)m), from_dev(s, 3 * (size_t)m));
        std::printf("unit vector   %d vectors: short form differs from unit_vector in %lld components\n", m, a);
        bad += a != 0;
    }
    {   // unit-sphere test
        std::vector<uint32_t> w(3 * (size_t)n);
        for (auto& v : w) v = (uint32_t)g();
        // near the surface: mx, my at random, mz = the integer root of what is missing to 2^60; the sum then misses 2^60 by
        // up to 2 mz, so about one triple in 2^20 lands inside the band of +-1024 -- 2^29 of them are tried
        int made = 0;
        for (long long tries = 0; tries < (1ll << 29) && made < n / 4; ++tries) {
            const uint64_t r = g();
            const int64_t mx = (int64_t)(r & 0x3FFFFFFFu) - (1 << 29), my = (int64_t)((r >> 32) & 0x3FFFFFFFu) - (1 << 29);
            const uint64_t rest = (1ull << 60) - (uint64_t)(mx * mx) - (uint64_t)(my * my);
            int64_t mz = (int64_t)std::sqrt((double)rest);
            while ((uint64_t)(mz * mz) > rest) --mz;
            while ((uint64_t)((mz + 1) * (mz + 1)) <= rest) ++mz;
            const uint64_t below = rest - (uint64_t)(mz * mz), above = (uint64_t)((mz + 1) * (mz + 1)) - rest;
            if (below > 1200 && above > 1200) continue;
            if (above < below) ++mz;
            if (mz >= (1 << 30)) continue;
            const size_t at = 3 * (size_t)made++;
            w[at] = (uint32_t)((mx + (1 << 30)) << 1) | (uint32_t)(r >> 63);
            w[at + 1] = (uint32_t)((my + (1 << 30)) << 1);
            w[at + 2] = (uint32_t)(((r >> 62 & 1 ? mz : -mz) + (1 << 30)) << 1);
        }
        uint32_t* dw = to_dev(w);
        uint8_t *f, *s;
        (void)hipMalloc(&f, n); (void)hipMalloc(&s, n);
        hipLaunchKernelGGL(k_sphere, dim3(n / 256), dim3(256), 0, 0, n, dw, f, s);
        const auto vf = from_dev(f, n), vs = from_dev(s, n);
        long long a = 0, inside = 0, band = 0;
        for (int i = 0; i < n; ++i) {
            a += vf[i] != vs[i];
            inside += vs[i];
            const int64_t mx = (int64_t)(w[3 * (size_t)i] >> 1) - (1 << 30), my = (int64_t)(w[3 * (size_t)i + 1] >> 1) - (1 << 30),
                          mz = (int64_t)(w[3 * (size_t)i + 2] >> 1) - (1 << 30);
            const uint64_t S = (uint64_t)(mx * mx) + (uint64_t)(my * my) + (uint64_t)(mz * mz);
            band += S >= (1ull << 60) - 1024 && S <= (1ull << 60) + 1024;
        }
        std::printf("sphere test   %d word triples (%lld inside, %d made to land near the surface, %lld within the band the doubles decide): "
                    "integer test differs from the reference's in %lld\n", n, inside, made, band, a);
        bad += a != 0;
    }
    if (hipDeviceSynchronize() != hipSuccess) { std::printf("device error\n"); return 2; }
    std::printf(bad ? "MISMATCH\n" : "all short forms agree bit for bit\n");
    return bad ? 1 : 0;
}
