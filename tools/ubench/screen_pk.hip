// Micro-benchmark: the screening sequence on PAIRS of spheres with packed FP32 math
// (v_pk_fma_f32 / v_pk_add_f32, sphere pair in an SGPR pair) and two sign-collection forms.
//   form 0: 8 packed ops + 2 v_alignbit per pair
//   form 1: 8 packed ops (last with clamp -> 0/1 indicator) + 1 packed fma accumulating bits
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
struct P { f2 cx, cy, cz, k; };   // one pair of spheres, 32 bytes
typedef __attribute__((address_space(4))) P CP;
__device__ __forceinline__ f2 pfma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
template <int F>
__global__ void __launch_bounds__(256) k(const P* sph, int ngroups, int reps, unsigned* out, unsigned long long* clk) {
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    float dx = 0.3f + threadIdx.x * 1e-3f, dy = 0.5f + blockIdx.x * 1e-4f, dz = 0.8f, od = 0.1f * blockIdx.x, e = -2.0f;
    float px = 1.0f + threadIdx.x, py = 2.0f + threadIdx.x * 1e-2f, pz = 3.0f;
    unsigned acc = 0;
    for (int r = 0; r < reps; ++r) {
        for (int g = 0; g < ngroups; ++g) {
            const CP* sp = (const CP*)(sph + g * 16);
            unsigned mask = 0;
            f2 bits = {0.f, 0.f};
            const f2 ndx = {-dx, -dx}, ndy = {-dy, -dy}, ndz = {-dz, -dz}, odd = {od, od}, ee = {e, e};
            const f2 pxx = {px, px}, pyy = {py, py}, pzz = {pz, pz};
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const f2 cx = sp[j].cx, cy = sp[j].cy, cz = sp[j].cz, kk = sp[j].k;
                f2 hb = pfma(cx, ndx, odd);
                hb = pfma(cy, ndy, hb);
                hb = pfma(cz, ndz, hb);
                f2 a = ee - kk;
                a = pfma(pxx, cx, a);
                a = pfma(pyy, cy, a);
                a = pfma(pzz, cz, a);
                f2 q = pfma(hb, hb, a);
                if (F == 0) {
                    mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(q.x), 31);
                    mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(q.y), 31);
                } else {
                    q = __builtin_elementwise_min(__builtin_elementwise_max(q, f2{0.f, 0.f}), f2{1.f, 1.f});
                    bits = pfma(bits, f2{2.f, 2.f}, q);
                }
            }
            if (F == 1) mask = (unsigned)bits.x ^ ((unsigned)bits.y << 16);
            acc ^= mask;
            e += 1e-4f;
        }
        od += 1e-3f;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = __builtin_readcyclecounter() - t0; clk[2 * blockIdx.x + 1] = wall_clock64() - w0; }
}
int main() {
    const int ngroups = 16, reps = 1500;
    std::vector<P> h(ngroups * 16);
    for (size_t i = 0; i < h.size(); ++i) { float x = (float)(i % 23) - 11.f; h[i] = P{{x, x + 0.5f}, {0.2f, 0.2f}, {(float)(i / 23) - 11.f, 3.f}, {0.04f, 0.05f}}; }
    P* d; unsigned* o; unsigned long long* c;
    hipMalloc(&d, h.size() * sizeof(P)); hipMemcpy(d, h.data(), h.size() * sizeof(P), hipMemcpyHostToDevice);
    hipMalloc(&o, 256 * 8 * 256 * 4); hipMalloc(&c, 256 * 8 * 16);
    std::vector<unsigned long long> hc(256 * 8 * 2);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    for (int f = 0; f < 2; ++f)
        for (int bpc : {2, 4, 6, 8}) {
            int blocks = p.multiProcessorCount * bpc;
            auto kk = f == 0 ? k<0> : k<1>;
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipLaunchKernelGGL(kk, dim3(blocks), dim3(256), 0, 0, d, ngroups, 10, o, c);
            hipEventRecord(a); hipLaunchKernelGGL(kk, dim3(blocks), dim3(256), 0, 0, d, ngroups, reps, o, c); hipEventRecord(b);
            hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
            hipMemcpy(hc.data(), c, blocks * 16, hipMemcpyDeviceToHost);
            double mhz = 0; for (int i = 0; i < blocks; ++i) mhz += (double)hc[2 * i] / (double)hc[2 * i + 1] * 100.0; mhz /= blocks;
            double tests_per_simd = (double)bpc * reps * ngroups * 32;
            double cycles = ms * 1e-3 * mhz * 1e6;
            printf("pk form %d waves/SIMD %d clock %.0f MHz: %.2f cycles per sphere test per SIMD, %.1f Gtests/s\n", f, bpc, mhz,
                   cycles / tests_per_simd, (double)blocks * 256 * reps * ngroups * 32 / ms / 1e6);
        }
    return 0;
}
