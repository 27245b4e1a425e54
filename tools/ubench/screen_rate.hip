// Micro-benchmark: issue rate of the 9-op sphere screening sequence (scalar-loaded sphere,
// per-lane ray) as a function of resident waves per SIMD.  Prints VALU wave-instructions per
// cycle per SIMD against the 0.5 peak of a wave64 FP32 op on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct S { float cx, cy, cz, k; };
typedef __attribute__((address_space(4))) S CS;
template <int VARIANT>
__global__ void __launch_bounds__(256) k(const S* sph, int ngroups, int reps, unsigned* out, unsigned long long* clk) {
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    float dx = 0.3f + threadIdx.x * 1e-3f, dy = 0.5f, dz = 0.8f, od = 0.1f * blockIdx.x, e = -2.0f;
    float px = 1.0f + threadIdx.x, py = 2.0f, pz = 3.0f;
    unsigned acc = 0;
    for (int r = 0; r < reps; ++r) {
        for (int g = 0; g < ngroups; ++g) {
            const CS* sp = (const CS*)(sph + g * 32);
            unsigned mask = 0;
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const float cx = sp[j].cx, cy = sp[j].cy, cz = sp[j].cz, kk = sp[j].k;
                float hb = __builtin_fmaf(-cx, dx, od);
                hb = __builtin_fmaf(-cy, dy, hb);
                hb = __builtin_fmaf(-cz, dz, hb);
                float a = e - kk;
                a = __builtin_fmaf(px, cx, a);
                a = __builtin_fmaf(py, cy, a);
                a = __builtin_fmaf(pz, cz, a);
                const float q = __builtin_fmaf(hb, hb, a);
                if (VARIANT == 0) mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(q), 31);
                else mask ^= __float_as_uint(q);
            }
            acc ^= mask;
        }
        od += 1e-3f;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0) {
        clk[2 * blockIdx.x] = __builtin_readcyclecounter() - t0;
        clk[2 * blockIdx.x + 1] = wall_clock64() - w0;
    }
}
int main() {
    const int ngroups = 16, reps = 2000;
    std::vector<S> h(ngroups * 32);
    for (size_t i = 0; i < h.size(); ++i) h[i] = S{(float)(i % 23) - 11.f, 0.2f, (float)(i / 23) - 11.f, 0.04f};
    S* d; unsigned* o;
    hipMalloc(&d, h.size() * sizeof(S)); hipMemcpy(d, h.data(), h.size() * sizeof(S), hipMemcpyHostToDevice);
    hipMalloc(&o, 256 * 8 * 256 * sizeof(unsigned));
    unsigned long long* c; hipMalloc(&c, 256 * 8 * 2 * sizeof(unsigned long long));
    std::vector<unsigned long long> hc(256 * 8 * 2);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const double ghz = p.clockRate / 1e6;
    for (int variant = 0; variant < 2; ++variant)
    for (int bpc = 1; bpc <= 8; ++bpc) {   // 256-thread blocks per CU = waves per SIMD
        int blocks = p.multiProcessorCount * bpc;
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        auto kk = variant == 0 ? k<0> : k<1>;
        hipLaunchKernelGGL(kk, dim3(blocks), dim3(256), 0, 0, d, ngroups, 10, o, c);
        hipEventRecord(a); hipLaunchKernelGGL(kk, dim3(blocks), dim3(256), 0, 0, d, ngroups, reps, o, c); hipEventRecord(b);
        hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
        hipMemcpy(hc.data(), c, blocks * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double mhz = 0; for (int i = 0; i < blocks; ++i) mhz += (double)hc[2 * i] / (double)hc[2 * i + 1] * 100.0; mhz /= blocks;
        double tests = (double)blocks * 256 * reps * ngroups * 32;
        double instr_per_simd = (double)bpc * reps * ngroups * 32 * 9;    // wave-instructions issued on one SIMD
        double cycles = ms * 1e-3 * ghz * 1e9;
        printf("variant %d shader clock %.0f MHz  ", variant, mhz);
        printf("waves/SIMD %d  %.2f ms  %.1f Gtests/s  VALU wave-instr/cycle/SIMD %.3f (peak 0.5)  flop-equivalent %.1f TFLOP/s (23 flop/test)\n",
               bpc, ms, tests / ms / 1e6, instr_per_simd / cycles, tests * 23 / ms / 1e9);
    }
    return 0;
}
