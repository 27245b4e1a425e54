// Micro-benchmark (inline asm): throughput of the FP32 matrix instruction v_mfma_f32_16x16x4_f32 (16x16 outputs,
// K = 4, 4 accumulator registers per lane) and of v_mfma_f32_32x32x2_f32, alone and interleaved with packed FP32
// vector instructions: do the two pipes overlap, and how many vector issue slots does a matrix instruction take?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP4(X) X X X X
// F: 0 = 4 MFMA 16x16x4 per step; 1 = the same + 8 v_pk_fma_f32; 2 = 8 v_pk_fma_f32 only; 3 = 4 MFMA + 16 pk; 4 = 16 pk only;
//    5 = 2 MFMA 32x32x2 per step
template <int F>
__global__ void __launch_bounds__(256) k(int reps, float* out, unsigned long long* clk) {
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    f16v d0 = {}, d1 = {};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    f2 p0 = {1, 2}, p1 = {3, 4}, p2 = {5, 6}, p3 = {7, 8}, p4 = {1, 2}, p5 = {3, 4}, p6 = {5, 6}, p7 = {7, 8}, x = {0.999f, 0.998f}, y = {1e-3f, 2e-3f};
#define MF4 "v_mfma_f32_16x16x4_f32 %0, %4, %5, %0\n v_mfma_f32_16x16x4_f32 %1, %4, %5, %1\n v_mfma_f32_16x16x4_f32 %2, %4, %5, %2\n v_mfma_f32_16x16x4_f32 %3, %4, %5, %3\n"
#define PK8 "v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n" \
            "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
    for (int r = 0; r < reps; ++r) {
        if (F == 0 || F == 1 || F == 3) {
            asm volatile(REP4(MF4) : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b));
        }
        if (F == 1 || F == 2) {
            asm volatile(REP4(PK8) : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(x), "v"(y));
        }
        if (F == 3 || F == 4) {
            asm volatile(REP4(PK8) REP4(PK8) : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(x), "v"(y));
        }
        if (F == 5) {
            asm volatile(REP4("v_mfma_f32_32x32x2_f32 %0, %2, %3, %0\n v_mfma_f32_32x32x2_f32 %1, %2, %3, %1\n") : "+v"(d0), "+v"(d1) : "v"(a), "v"(b));
        }
    }
    float s = c0.x + c1.y + c2.z + c3.w + d0[0] + d1[5] + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = __builtin_readcyclecounter() - t0; clk[2 * blockIdx.x + 1] = wall_clock64() - w0; }
}
typedef void (*kern_t)(int, float*, unsigned long long*);
int main() {
    const int reps = 2000;
    float* o; unsigned long long* c;
    hipMalloc(&o, 256 * 8 * 256 * 4); hipMalloc(&c, 256 * 8 * 16);
    std::vector<unsigned long long> hc(256 * 8 * 2);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const char* names[] = {"16 mfma16x16x4", "16 mfma + 32 pk_fma (interleaved blocks)", "32 pk_fma", "16 mfma + 64 pk_fma", "64 pk_fma", "8 mfma32x32x2"};
    kern_t ks[] = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>};
    for (int f = 0; f < 6; ++f)
        for (int bpc : {1, 2, 4}) {
            int blocks = p.multiProcessorCount * bpc;
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipLaunchKernelGGL(ks[f], dim3(blocks), dim3(256), 0, 0, 10, o, c);
            hipEventRecord(a); hipLaunchKernelGGL(ks[f], dim3(blocks), dim3(256), 0, 0, reps, o, c); hipEventRecord(b);
            hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
            hipMemcpy(hc.data(), c, blocks * 16, hipMemcpyDeviceToHost);
            double mhz = 0; for (int i = 0; i < blocks; ++i) mhz += (double)hc[2 * i] / (double)hc[2 * i + 1] * 100.0; mhz /= blocks;
            double cycles = ms * 1e-3 * mhz * 1e6;
            printf("%-42s waves/SIMD %d clock %.0f MHz: %.1f cycles per loop step per wave (SIMD time / (waves*steps))\n", names[f], bpc, mhz, cycles / ((double)bpc * reps));
        }
    return 0;
}
