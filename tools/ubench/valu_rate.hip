// Micro-benchmark: raw issue rate of wave64 VALU FP32 FMAs on gfx950 in three operand forms.
//   variant 0: v_fma_f32 with VGPR operands only (8 independent chains)
//   variant 1: v_fma_f32 with one SGPR operand, constants loaded once (no scalar loads in the loop)
//   variant 2: v_pk_fma_f32 (two FMAs per lane per instruction), VGPR operands
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int V>
__global__ void __launch_bounds__(256) k(const float* cst, int reps, float* out, unsigned long long* clk) {
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    float x[8];
    f2 y[8];
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 1e-3f + i; y[i] = f2{x[i], x[i] + 0.5f}; }
    const float a = 1.0f + threadIdx.x * 1e-6f, b = 1e-3f;
    const float s0 = cst[0], s1 = cst[1];   // uniform -> SGPRs
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (V == 0) x[i] = __builtin_fmaf(x[i], a, b);
                if (V == 1) x[i] = __builtin_fmaf(x[i], s0, (i & 1) ? b : a);
                if (V == 2) y[i] = __builtin_elementwise_fma(y[i], f2{a, a}, f2{b, b});
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += x[i] + y[i].x + y[i].y + s1;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = __builtin_readcyclecounter() - t0; clk[2 * blockIdx.x + 1] = wall_clock64() - w0; }
}
int main() {
    const int reps = 20000;
    float hc2[2] = {0.999f, 0.0f};
    float *cst, *o; unsigned long long* c;
    hipMalloc(&cst, 8); hipMemcpy(cst, hc2, 8, hipMemcpyHostToDevice);
    hipMalloc(&o, 256 * 8 * 256 * 4); hipMalloc(&c, 256 * 8 * 16);
    std::vector<unsigned long long> hc(256 * 8 * 2);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    for (int v = 0; v < 3; ++v)
        for (int bpc : {1, 2, 4, 8}) {
            int blocks = p.multiProcessorCount * bpc;
            auto kk = v == 0 ? k<0> : v == 1 ? k<1> : k<2>;
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipLaunchKernelGGL(kk, dim3(blocks), dim3(256), 0, 0, cst, 10, o, c);
            hipEventRecord(a); hipLaunchKernelGGL(kk, dim3(blocks), dim3(256), 0, 0, cst, reps, o, c); hipEventRecord(b);
            hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
            hipMemcpy(hc.data(), c, blocks * 16, hipMemcpyDeviceToHost);
            double mhz = 0; for (int i = 0; i < blocks; ++i) mhz += (double)hc[2 * i] / (double)hc[2 * i + 1] * 100.0; mhz /= blocks;
            double instr_per_simd = (double)bpc * reps * 128;
            double cycles = ms * 1e-3 * mhz * 1e6;
            double flops = (double)blocks * 256 * reps * 128 * 2 * (v == 2 ? 2 : 1);
            printf("variant %d waves/SIMD %d clock %.0f MHz: %.3f VALU wave-instr/cycle/SIMD, %.1f TFLOP/s\n", v, bpc, mhz,
                   instr_per_simd / cycles, flops / ms / 1e9);
        }
    return 0;
}
