// Micro-benchmark: what limits the issue rate of the sphere screening sequence?
//   form 0: as in the render kernel (scalar loads inside the loop, VOP3 fma with neg modifiers)
//   form 1: same arithmetic, sphere constants held in SGPRs for the whole loop (no scalar loads)
//   form 2: scalar loads, but pre-negated direction so that every FMA is a 4-byte VOP2 v_fmac
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct S { float cx, cy, cz, k; };
typedef __attribute__((address_space(4))) S CS;
template <int F>
__global__ void __launch_bounds__(256) k(const S* sph, int ngroups, int reps, unsigned* out, unsigned long long* clk) {
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    float dx = 0.3f + threadIdx.x * 1e-3f, dy = 0.5f + blockIdx.x * 1e-4f, dz = 0.8f, od = 0.1f * blockIdx.x, e = -2.0f;
    float px = 1.0f + threadIdx.x, py = 2.0f, pz = 3.0f;
    unsigned acc = 0;
    for (int r = 0; r < reps; ++r) {
        for (int g = 0; g < ngroups; ++g) {
            const CS* sp = (const CS*)(sph + (F == 1 ? 0 : g * 32));
            unsigned mask = 0;
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const int jj = F == 1 ? (j & 3) : j;
                const float cx = sp[jj].cx, cy = sp[jj].cy, cz = sp[jj].cz, kk = sp[jj].k;
                float hb;
                if (F == 2) {
                    hb = od;
                    hb = __builtin_fmaf(cx, dx, hb);
                    hb = __builtin_fmaf(cy, dy, hb);
                    hb = __builtin_fmaf(cz, dz, hb);
                } else {
                    hb = __builtin_fmaf(-cx, dx, od);
                    hb = __builtin_fmaf(-cy, dy, hb);
                    hb = __builtin_fmaf(-cz, dz, hb);
                }
                float a = e - kk;
                a = __builtin_fmaf(px, cx, a);
                a = __builtin_fmaf(py, cy, a);
                a = __builtin_fmaf(pz, cz, a);
                const float q = __builtin_fmaf(hb, hb, a);
                mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(q), 31);
            }
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
    std::vector<S> h(ngroups * 32);
    for (size_t i = 0; i < h.size(); ++i) h[i] = S{(float)(i % 23) - 11.f, 0.2f, (float)(i / 23) - 11.f, 0.04f};
    S* d; unsigned* o; unsigned long long* c;
    hipMalloc(&d, h.size() * sizeof(S)); hipMemcpy(d, h.data(), h.size() * sizeof(S), hipMemcpyHostToDevice);
    hipMalloc(&o, 256 * 8 * 256 * 4); hipMalloc(&c, 256 * 8 * 16);
    std::vector<unsigned long long> hc(256 * 8 * 2);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    for (int f = 0; f < 3; ++f)
        for (int bpc : {2, 4, 8}) {
            int blocks = p.multiProcessorCount * bpc;
            auto kk = f == 0 ? k<0> : f == 1 ? k<1> : k<2>;
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipLaunchKernelGGL(kk, dim3(blocks), dim3(256), 0, 0, d, ngroups, 10, o, c);
            hipEventRecord(a); hipLaunchKernelGGL(kk, dim3(blocks), dim3(256), 0, 0, d, ngroups, reps, o, c); hipEventRecord(b);
            hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
            hipMemcpy(hc.data(), c, blocks * 16, hipMemcpyDeviceToHost);
            double mhz = 0; for (int i = 0; i < blocks; ++i) mhz += (double)hc[2 * i] / (double)hc[2 * i + 1] * 100.0; mhz /= blocks;
            double tests_per_simd = (double)bpc * reps * ngroups * 32;
            double cycles = ms * 1e-3 * mhz * 1e6;
            printf("form %d waves/SIMD %d clock %.0f MHz: %.2f cycles per sphere test per SIMD, %.1f Gtests/s\n", f, bpc, mhz,
                   cycles / tests_per_simd, (double)blocks * 256 * reps * ngroups * 32 / ms / 1e6);
        }
    return 0;
}
