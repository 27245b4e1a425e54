// How often the device's double-precision log / sin / cos / atan2 / acos differ from the host libm (glibc), on the kind
// of arguments the kernels pass: log of 31-bit uniforms, sin / cos of moderate angles, atan2 / acos of unit-vector
// components.  Prints the number of differing results and the largest difference in ulps.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
__global__ void k(int n, const double* a, const double* b, double* o_log, double* o_sin, double* o_cos, double* o_atan2, double* o_acos) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    o_log[i] = log(a[i]);
    o_sin[i] = sin(b[i]);
    o_cos[i] = cos(b[i]);
    o_atan2[i] = atan2(-b[i], a[i] - 0.5);
    o_acos[i] = acos(1.0 - 2.0 * a[i]);
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
    const int n = 1 << 24;
    std::vector<double> a(n), b(n);
    std::mt19937_64 g(7);
    for (int i = 0; i < n; ++i) {
        a[i] = (double)((uint32_t)(g() >> 33) | 1u) * 4.656612873077392578125e-10;  // (0, 1) on the 31-bit grid
        b[i] = ((double)(g() >> 11) * 0x1p-53 - 0.5) * 80.0;                          // angles in (-40, 40)
    }
    double *da, *db, *d[5];
    (void)hipMalloc(&da, n * 8); (void)hipMalloc(&db, n * 8);
    for (auto& p : d) (void)hipMalloc(&p, n * 8);
    (void)hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, n, da, db, d[0], d[1], d[2], d[3], d[4]);
    std::vector<double> o(n);
    const char* names[5] = {"log(u)", "sin(x)", "cos(x)", "atan2(-x, u-0.5)", "acos(1-2u)"};
    for (int f = 0; f < 5; ++f) {
        (void)hipMemcpy(o.data(), d[f], n * 8, hipMemcpyDeviceToHost);
        long long diff = 0, worst = 0;
        for (int i = 0; i < n; ++i) {
            const double h = f == 0 ? std::log(a[i]) : f == 1 ? std::sin(b[i]) : f == 2 ? std::cos(b[i])
                           : f == 3 ? std::atan2(-b[i], a[i] - 0.5) : std::acos(1.0 - 2.0 * a[i]);
            const long long u = ulps(h, o[i]);
            if (u) ++diff;
            if (u > worst) worst = u;
        }
        std::printf("%-18s %9lld of %d results differ from the host libm (%.2e), largest difference %lld ulp\n", names[f], diff, n,
                    (double)diff / n, worst);
    }
    return 0;
}
