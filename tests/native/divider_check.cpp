// make_u32_divider / div_u32 (raytracing_amd/csrc/rt_internal.h: work item -> pixel) and make_width_divider /
// div_by_width's arithmetic (pixel -> row) against the integer division they stand for.  Host code only.
#include <cstdint>
#include <cstdio>
#include <random>

#include "rt_internal.h"

int main() {
    std::mt19937_64 g(3);
    long long bad = 0, checked = 0;
    auto check = [&](uint32_t d, int random_numerators) {
        const rtd::U32Divider v = rtd::make_u32_divider(d);
        const uint32_t top = (0xFFFFFFFFu / d) * d;
        const uint32_t edge[] = {0u, 1u, d - 1u, d, d + 1u, 2u * d - 1u, 2u * d, 0x7FFFFFFFu, 0x80000000u, 0xFFFFFFFEu, 0xFFFFFFFFu, top, top - 1u};
        for (uint32_t n : edge) {
            bad += rtd::div_u32(n, v) != n / d;
            ++checked;
        }
        for (int i = 0; i < random_numerators; ++i) {
            const uint32_t n = (uint32_t)g();
            bad += rtd::div_u32(n, v) != n / d;
            ++checked;
        }
    };
    for (uint32_t d = 1; d <= 70000u; ++d) check(d, 200);  // every run count and small frame the plans can name
    for (int i = 0; i < 100000; ++i) {
        check((uint32_t)g() | 1u, 50);
        check((uint32_t)(g() >> (g() % 31 + 33)) | 1u, 50);
    }
    for (uint32_t d : {0xFFFFFFFFu, 0x80000000u, 0x80000001u, 960000u, 810000u, 640000u, 1u << 28}) check(d, 100000);
    // the frame's width divider: exact for every pixel index below 2^28
    for (int32_t width : {2, 3, 7, 200, 400, 600, 800, 1200, 1920, 4096, 16383, 1 << 20}) {
        const rtd::WidthDivider w = rtd::make_width_divider(width);
        for (int i = 0; i < 200000; ++i) {
            const uint32_t n = (uint32_t)g() & 0x0FFFFFFFu;
            bad += (uint32_t)(((uint64_t)n * (uint64_t)w.mul) >> w.shift) != n / (uint32_t)width;
            ++checked;
        }
    }
    std::printf("%lld of %lld quotients differ\n", bad, checked);
    return bad != 0;
}
