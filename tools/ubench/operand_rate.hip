// Micro-benchmark (inline asm, exact instruction streams): issue rate of wave64 v_fma_f32 /
// v_fmac_f32 by operand kind, 8 independent accumulators, no memory instructions in the loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP8(X) X X X X X X X X
template <int F>
__global__ void __launch_bounds__(256) k(int reps, float* out, unsigned long long* clk, float sa, float sb) {
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, x = 0.999f, y = 1e-3f;
    for (int r = 0; r < reps; ++r) {
        if (F == 0) {  // VOP3, VGPR operands only
            asm volatile(REP8("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
        } else if (F == 1) {  // VOP3, one SGPR operand
            asm volatile(REP8("v_fma_f32 %0, %8, %0, %9\n v_fma_f32 %1, %8, %1, %9\n v_fma_f32 %2, %8, %2, %9\n v_fma_f32 %3, %8, %3, %9\n"
                              "v_fma_f32 %4, %8, %4, %9\n v_fma_f32 %5, %8, %5, %9\n v_fma_f32 %6, %8, %6, %9\n v_fma_f32 %7, %8, %7, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(sa), "v"(y));
        } else if (F == 2) {  // VOP2 v_fmac, SGPR src0
            asm volatile(REP8("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                              "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(sa), "v"(y));
        } else if (F == 3) {  // VOP2 v_fmac, VGPR operands
            asm volatile(REP8("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                              "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
        } else if (F == 4) {  // dependent pairs: chain of 2 accumulators only (latency exposure)
            asm volatile(REP8("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n"
                              "v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(sa), "v"(y));
        } else {  // v_alignbit_b32
            asm volatile(REP8("v_alignbit_b32 %0, %0, %8, 31\n v_alignbit_b32 %1, %1, %8, 31\n v_alignbit_b32 %2, %2, %8, 31\n v_alignbit_b32 %3, %3, %8, 31\n"
                              "v_alignbit_b32 %4, %4, %8, 31\n v_alignbit_b32 %5, %5, %8, 31\n v_alignbit_b32 %6, %6, %8, 31\n v_alignbit_b32 %7, %7, %8, 31\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + sb;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = __builtin_readcyclecounter() - t0; clk[2 * blockIdx.x + 1] = wall_clock64() - w0; }
}
int main() {
    const int reps = 20000;
    float* o; unsigned long long* c;
    hipMalloc(&o, 256 * 8 * 256 * 4); hipMalloc(&c, 256 * 8 * 16);
    std::vector<unsigned long long> hc(256 * 8 * 2);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const char* names[] = {"v_fma VGPR", "v_fma SGPR", "v_fmac SGPR", "v_fmac VGPR", "v_fmac SGPR 2 chains", "v_alignbit"};
    for (int f = 0; f < 6; ++f)
        for (int bpc : {1, 2, 4, 8}) {
            int blocks = p.multiProcessorCount * bpc;
            auto kk = f == 0 ? k<0> : f == 1 ? k<1> : f == 2 ? k<2> : f == 3 ? k<3> : f == 4 ? k<4> : k<5>;
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipLaunchKernelGGL(kk, dim3(blocks), dim3(256), 0, 0, 10, o, c, 0.999f, 0.f);
            hipEventRecord(a); hipLaunchKernelGGL(kk, dim3(blocks), dim3(256), 0, 0, reps, o, c, 0.999f, 0.f); hipEventRecord(b);
            hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
            hipMemcpy(hc.data(), c, blocks * 16, hipMemcpyDeviceToHost);
            double mhz = 0; for (int i = 0; i < blocks; ++i) mhz += (double)hc[2 * i] / (double)hc[2 * i + 1] * 100.0; mhz /= blocks;
            double instr_per_simd = (double)bpc * reps * 64;
            double cycles = ms * 1e-3 * mhz * 1e6;
            printf("%-22s waves/SIMD %d clock %.0f MHz: %.2f cycles per wave-instruction per SIMD\n", names[f], bpc, mhz, cycles / instr_per_simd);
        }
    return 0;
}
