// Micro-benchmark (inline asm): issue rate of the wave64 instruction classes used outside the screening pass
// (Philox: 32x32->64 multiplies and xors; shading: FP64 add/mul/fma, rcp, rsq, div helpers), 8 independent
// accumulators per lane, no memory instructions in the loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP8(X) X X X X X X X X
#define OPS8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
template <int F>
__global__ void __launch_bounds__(256) k(int reps, double* out, unsigned long long* clk, unsigned sa) {
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    double a0 = threadIdx.x + 1.0, a1 = 1.5, a2 = 2.5, a3 = 3.5, a4 = 4.5, a5 = 5.5, a6 = 6.5, a7 = 7.5, x = 0.999, y = 1e-3;
    unsigned long long u0 = threadIdx.x, u1 = 1, u2 = 2, u3 = 3, u4 = 4, u5 = 5, u6 = 6, u7 = 7;
    unsigned long long sm = 0x5555555555555555ull + sa;
    unsigned m = 0xD2511F53u + sa, i0 = threadIdx.x, i1 = 1, i2 = 2, i3 = 3, i4 = 4, i5 = 5, i6 = 6, i7 = 7;
    for (int r = 0; r < reps; ++r) {
        if (F == 0) {
            asm volatile(REP8("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                              "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
        } else if (F == 1) {
            asm volatile(REP8("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n"
                              "v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
        } else if (F == 2) {
            asm volatile(REP8("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                              "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(y), "v"(y));
        } else if (F == 3) {
            asm volatile(REP8("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n"
                              "v_rcp_f64 %4, %4\n v_rcp_f64 %5, %5\n v_rcp_f64 %6, %6\n v_rcp_f64 %7, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(y), "v"(y));
        } else if (F == 4) {
            asm volatile(REP8("v_rsq_f64 %0, %0\n v_rsq_f64 %1, %1\n v_rsq_f64 %2, %2\n v_rsq_f64 %3, %3\n"
                              "v_rsq_f64 %4, %4\n v_rsq_f64 %5, %5\n v_rsq_f64 %6, %6\n v_rsq_f64 %7, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(y), "v"(y));
        } else if (F == 5) {  // 32x32 -> 64 multiply-add
            asm volatile(REP8("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
                              "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n")
                         : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(m), "v"(i0) : "vcc");
        } else if (F == 6) {
            asm volatile(REP8("v_mul_hi_u32 %0, %0, %8\n v_mul_hi_u32 %1, %1, %8\n v_mul_hi_u32 %2, %2, %8\n v_mul_hi_u32 %3, %3, %8\n"
                              "v_mul_hi_u32 %4, %4, %8\n v_mul_hi_u32 %5, %5, %8\n v_mul_hi_u32 %6, %6, %8\n v_mul_hi_u32 %7, %7, %8\n")
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(m), "v"(m));
        } else if (F == 7) {
            asm volatile(REP8("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
                              "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n")
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(m), "v"(m));
        } else if (F == 8) {
            asm volatile(REP8("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n"
                              "v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8\n")
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(m), "v"(m));
        } else if (F == 9) {  // 24-bit multiplies (full-rate candidates)
            asm volatile(REP8("v_mul_u32_u24 %0, %0, %8\n v_mul_u32_u24 %1, %1, %8\n v_mul_u32_u24 %2, %2, %8\n v_mul_u32_u24 %3, %3, %8\n"
                              "v_mul_hi_u32_u24 %4, %4, %8\n v_mul_hi_u32_u24 %5, %5, %8\n v_mul_hi_u32_u24 %6, %6, %8\n v_mul_hi_u32_u24 %7, %7, %8\n")
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(m), "v"(m));
        } else if (F == 10) {  // xor with 3 inputs
            asm volatile(REP8("v_lshl_add_u32 %0, %0, 1, %8\n v_lshl_add_u32 %1, %1, 1, %8\n v_lshl_add_u32 %2, %2, 1, %8\n v_lshl_add_u32 %3, %3, 1, %8\n"
                              "v_lshl_add_u32 %4, %4, 1, %8\n v_lshl_add_u32 %5, %5, 1, %8\n v_lshl_add_u32 %6, %6, 1, %8\n v_lshl_add_u32 %7, %7, 1, %8\n")
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(m), "v"(i0));
        } else if (F == 12) {  // v_cndmask, mask in an SGPR pair (VOP3)
            asm volatile(REP8("v_cndmask_b32_e64 %0, %0, %8, %9\n v_cndmask_b32_e64 %1, %1, %8, %9\n v_cndmask_b32_e64 %2, %2, %8, %9\n v_cndmask_b32_e64 %3, %3, %8, %9\n"
                              "v_cndmask_b32_e64 %4, %4, %8, %9\n v_cndmask_b32_e64 %5, %5, %8, %9\n v_cndmask_b32_e64 %6, %6, %8, %9\n v_cndmask_b32_e64 %7, %7, %8, %9\n")
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(m), "s"(sm));
        } else if (F == 13) {  // compare + select pairs, as compiled code does
            asm volatile(REP8("v_cmp_lt_u32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_u32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n"
                              "v_cmp_lt_u32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_lt_u32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %8, vcc\n")
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(m), "v"(m) : "vcc");
        } else if (F == 14) {  // plain moves
            asm volatile(REP8("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n"
                              "v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n")
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(m), "v"(m));
        } else if (F == 15) {  // compare only (writes vcc)
            asm volatile(REP8("v_cmp_lt_u32 vcc, %0, %8\n v_cmp_lt_u32 vcc, %1, %8\n v_cmp_lt_u32 vcc, %2, %8\n v_cmp_lt_u32 vcc, %3, %8\n"
                              "v_cmp_lt_u32 vcc, %4, %8\n v_cmp_lt_u32 vcc, %5, %8\n v_cmp_lt_u32 vcc, %6, %8\n v_cmp_lt_u32 vcc, %7, %8\n")
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(m), "v"(m) : "vcc");
        } else if (F == 16) {  // v_cmp_f64 + 2 selects (a double select)
            asm volatile(REP8("v_cmp_lt_f64 vcc, %0, %8\n v_cndmask_b32 %4, %4, %9, vcc\n v_cndmask_b32 %5, %5, %9, vcc\n v_cmp_lt_f64 vcc, %1, %8\n"
                              "v_cndmask_b32 %6, %6, %9, vcc\n v_cndmask_b32 %7, %7, %9, vcc\n v_cmp_lt_f64 vcc, %2, %8\n v_cmp_lt_f64 vcc, %3, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(x), "v"(m) : "vcc");
        } else {  // v_cndmask
            asm volatile(REP8("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                              "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n")
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(m), "v"(m) : "vcc");
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (double)(u0 + u1 + u2 + u3 + u4 + u5 + u6 + u7) +
                                          (double)(i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7);
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = __builtin_readcyclecounter() - t0; clk[2 * blockIdx.x + 1] = wall_clock64() - w0; }
}
typedef void (*kern_t)(int, double*, unsigned long long*, unsigned);
int main() {
    const int reps = 5000;
    double* o; unsigned long long* c;
    hipMalloc(&o, 256 * 8 * 256 * 8); hipMalloc(&c, 256 * 8 * 16);
    std::vector<unsigned long long> hc(256 * 8 * 2);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const char* names[] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_rcp_f64", "v_rsq_f64", "v_mad_u64_u32", "v_mul_hi_u32", "v_mul_lo_u32",
                           "v_xor_b32", "v_mul(_hi)_u32_u24", "v_lshl_add_u32", "v_cndmask_b32 vcc", "v_cndmask_b32 sgpr mask", "v_cmp+v_cndmask", "v_mov_b32", "v_cmp_lt_u32", "v_cmp_f64+2 cndmask mix"};
    kern_t ks[] = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>, k<7>, k<8>, k<9>, k<10>, k<11>, k<12>, k<13>, k<14>, k<15>, k<16>};
    for (int f = 0; f < 17; ++f)
        for (int bpc : {1, 4}) {
            int blocks = p.multiProcessorCount * bpc;
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipLaunchKernelGGL(ks[f], dim3(blocks), dim3(256), 0, 0, 10, o, c, 0u);
            hipEventRecord(a); hipLaunchKernelGGL(ks[f], dim3(blocks), dim3(256), 0, 0, reps, o, c, 0u); hipEventRecord(b);
            hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
            hipMemcpy(hc.data(), c, blocks * 16, hipMemcpyDeviceToHost);
            double mhz = 0; for (int i = 0; i < blocks; ++i) mhz += (double)hc[2 * i] / (double)hc[2 * i + 1] * 100.0; mhz /= blocks;
            double instr_per_simd = (double)bpc * reps * 64;
            double cycles = ms * 1e-3 * mhz * 1e6;
            printf("%-20s waves/SIMD %d clock %.0f MHz: %.2f cycles per wave-instruction per SIMD\n", names[f], bpc, mhz, cycles / instr_per_simd);
        }
    return 0;
}
