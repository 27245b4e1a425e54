// Host-only pieces of the C ABI: camera set-up, tone mapping and the P3 writer.  Compiled
// with -ffp-contract=off like the kernels: these restate double arithmetic of the reference.
#include <cmath>
#include <cstdio>

#include "raytracing_amd.h"

namespace {
struct v3 {
    double x, y, z;
};
inline v3 operator+(v3 a, v3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline v3 operator-(v3 a, v3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline v3 operator*(double t, v3 a) { return {t * a.x, t * a.y, t * a.z}; }
inline v3 over(v3 a, double t) { return (1 / t) * a; }  // B1/vec3.h:96-98
inline double length(v3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
inline v3 unit(v3 a) { return over(a, length(a)); }
inline v3 cross(v3 u, v3 v) { return {u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x}; }
inline void put(double* dst, v3 a) {
    dst[0] = a.x;
    dst[1] = a.y;
    dst[2] = a.z;
}
}  // namespace

extern "C" {

// camera::camera, B1/camera.h:17-31 and B2/camera.h:19-35
void rt_camera_init(rt_camera_desc* out, const double lookfrom[3], const double lookat[3], const double vup[3],
                    double vfov_degrees, double aspect_ratio, double aperture, double focus_dist, double time0,
                    double time1) {
    const double pi = std::acos(-1.0);                // B1/rtweekend.h:17
    const double theta = vfov_degrees * pi / 180.0;   // B1/rtweekend.h:20-22
    const double h = std::tan(theta / 2);
    const double viewport_height = 2.0 * h;
    const double viewport_width = aspect_ratio * viewport_height;
    const v3 from{lookfrom[0], lookfrom[1], lookfrom[2]}, at{lookat[0], lookat[1], lookat[2]};
    const v3 w = unit(from - at);
    const v3 u = unit(cross(v3{vup[0], vup[1], vup[2]}, w));
    const v3 v = cross(w, u);
    const v3 horizontal = (focus_dist * viewport_width) * u;
    const v3 vertical = (focus_dist * viewport_height) * v;
    const v3 llc = from - over(horizontal, 2) - over(vertical, 2) - focus_dist * w;
    put(out->origin, from);
    put(out->lower_left_corner, llc);
    put(out->horizontal, horizontal);
    put(out->vertical, vertical);
    put(out->u, u);
    put(out->v, v);
    put(out->w, w);
    out->lens_radius = aperture / 2;
    out->time0 = time0;
    out->time1 = time1;
}

// write_color, B1/color.h:8-23
void rt_write_color(const double* accum_rgb, int width, int height, int samples_per_pixel, uint8_t* rgb_out) {
    const double scale = 1.0 / samples_per_pixel;
    const size_t n = (size_t)3 * (size_t)width * (size_t)height;
    for (size_t k = 0; k < n; ++k) {
        double c = std::sqrt(scale * accum_rgb[k]);
        if (c != c) {  // NaN: the reference prints INT_MIN here
            rgb_out[k] = 0;
            continue;
        }
        if (c < 0.0) c = 0.0;      // clamp, B1/rtweekend.h:42-46
        if (c > 0.999) c = 0.999;
        rgb_out[k] = (uint8_t)static_cast<int>(256 * c);
    }
}

// The P3 stream of B1/RayTracing.cpp:113 and B1/color.h:20-22.
int rt_write_ppm(const char* path, const uint8_t* rgb, int width, int height) {
    if (!path || !rgb || width <= 0 || height <= 0) return RT_ERR_INVALID_ARGUMENT;
    FILE* fp = std::fopen(path, "wb");
    if (!fp) return RT_ERR_INVALID_ARGUMENT;
    std::fprintf(fp, "P3\n%d %d\n255\n", width, height);
    const size_t n = (size_t)width * (size_t)height;
    for (size_t k = 0; k < n; ++k) std::fprintf(fp, "%d %d %d\n", rgb[3 * k], rgb[3 * k + 1], rgb[3 * k + 2]);
    return std::fclose(fp) == 0 ? RT_OK : RT_ERR_INVALID_ARGUMENT;
}

}  // extern "C"
