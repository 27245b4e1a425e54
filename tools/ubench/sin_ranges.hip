// sin() of the device libm against the host libm (glibc) by magnitude of the argument: per decade of |x|, how many
// results differ from the host's and by how much (in ulps and absolutely).  Measured on gfx950: about 3.1 % of the
// results differ, by one ulp, at every magnitude from 0.1 to 1e12 -- the device's argument reduction is as good as the
// host's.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
__global__ void k(int n, const double* x, double* o_dev) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    o_dev[i] = sin(x[i]);
}
static long long ulps(double x, double y) {
    long long a, b;
    std::memcpy(&a, &x, 8);
    std::memcpy(&b, &y, 8);
    if (a < 0) a = (long long)0x8000000000000000ull - a;
    if (b < 0) b = (long long)0x8000000000000000ull - b;
    return a > b ? a - b : b - a;
}
int main() {
    const int n = 1 << 20;
    std::vector<double> x(n), o1(n);
    std::mt19937_64 g(11);
    double *dx, *d1;
    (void)hipMalloc(&dx, n * 8); (void)hipMalloc(&d1, n * 8);
    for (int dec = -1; dec <= 11; ++dec) {
        const double lo = std::pow(10.0, dec), hi = 10.0 * lo;
        for (int i = 0; i < n; ++i) x[i] = (lo + (hi - lo) * ((double)(g() >> 11) * 0x1p-53)) * ((g() & 1) ? 1.0 : -1.0);
        (void)hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, n, dx, d1);
        (void)hipMemcpy(o1.data(), d1, n * 8, hipMemcpyDeviceToHost);
        long long diff1 = 0, worst1 = 0;
        double abs1 = 0;
        for (int i = 0; i < n; ++i) {
            const double h = std::sin(x[i]);
            const long long u1 = ulps(h, o1[i]);
            diff1 += u1 != 0;
            if (u1 > worst1) worst1 = u1;
            abs1 = std::fmax(abs1, std::fabs(h - o1[i]));
        }
        std::printf("|x| in [1e%d, 1e%d): device sin differs in %.3f %% (worst %lld ulp, %.1e abs)\n", dec, dec + 1, 100.0 * diff1 / n, worst1, abs1);
    }
    return 0;
}
