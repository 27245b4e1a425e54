// What the three host programs share after their scene and camera are built: the device side of the reference's
// main() -- B1/RayTracing.cpp:113-129, B2/RayTracing.cpp:361-379: the render loop, write_color per pixel, "Done." --
// through the C ABI, on one GPU (rt_context_*) or on several (rt_group_*: samples split over the GPUs, one RCCL
// all-reduce per frame).
#pragma once

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "raytracing_amd.h"

namespace app {

struct options {
    int gpus = 1;                 // --gpus N: devices 0..N-1
    std::string mode = "render";  // --mode render | firsthit
};

// Consumes "--gpus N" / "--mode M" at argv[i]; returns false if argv[i] is something else.
inline bool parse_common(options& o, int argc, char** argv, int& i) {
    const std::string a = argv[i];
    if (a == "--gpus" && i + 1 < argc) {
        o.gpus = std::atoi(argv[++i]);
        return true;
    }
    if (a == "--mode" && i + 1 < argc) {
        o.mode = argv[++i];
        return true;
    }
    return false;
}

inline int die(const char* what, int rc, const char* detail) {
    std::fprintf(stderr, "%s: %s (%s)\n", what, rt_error_string(rc), detail ? detail : "");
    return 1;
}

// Renders `p` (the whole frame) of `scene` seen through `cam` and writes the reference's P3 stream to stdout; with
// mode "firsthit", writes instead the closest-hit object of every pixel centre as a P2 image whose grey value is the
// object's index in the world list plus one (0 = the ray leaves the world; indices above 65534 saturate).
inline int run(const options& o, const rt_scene_desc& scene, const rt_camera_desc& cam, const rt_render_params& p,
               const char* label) {
    if (o.gpus < 1) {
        std::fprintf(stderr, "--gpus must be at least 1\n");
        return 2;
    }
    if (o.mode != "render" && o.mode != "firsthit") {
        std::fprintf(stderr, "unknown mode %s (render, firsthit)\n", o.mode.c_str());
        return 2;
    }
    rt_context* ctx = nullptr;
    rt_group* grp = nullptr;
    int rc;
    if (o.gpus == 1) {
        if ((rc = rt_context_create(0, nullptr, &ctx)) != RT_OK) return die("rt_context_create", rc, rt_last_error_detail(nullptr));
        if ((rc = rt_scene_upload(ctx, &scene)) != RT_OK) return die("rt_scene_upload", rc, rt_last_error_detail(ctx));
    } else {
        std::vector<int> devices((size_t)o.gpus);
        for (int k = 0; k < o.gpus; ++k) devices[(size_t)k] = k;
        if ((rc = rt_group_create(devices.data(), o.gpus, 1, &grp)) != RT_OK) return die("rt_group_create", rc, rt_group_last_error_detail(nullptr));
        if ((rc = rt_group_scene_upload(grp, &scene)) != RT_OK) return die("rt_group_scene_upload", rc, rt_group_last_error_detail(grp));
        ctx = rt_group_context(grp, 0, 0);
    }

    if (o.mode == "firsthit") {
        std::vector<int32_t> ids((size_t)p.width * p.height);
        if ((rc = rt_first_hit(ctx, &cam, p.width, p.height, ids.data())) != RT_OK) return die("rt_first_hit", rc, rt_last_error_detail(ctx));
        std::printf("P2\n%d %d\n65535\n", p.width, p.height);
        for (int32_t id : ids) std::printf("%d\n", id < 0 ? 0 : (id >= 65535 ? 65535 : id + 1));
    } else {
        std::vector<double> sums((size_t)3 * p.width * p.height);
        rt_stats st;
        const auto t0 = std::chrono::steady_clock::now();
        if (grp) {
            if ((rc = rt_group_render(grp, &cam, &p, sums.data(), &st)) != RT_OK) return die("rt_group_render", rc, rt_group_last_error_detail(grp));
        } else {
            if ((rc = rt_render(ctx, &cam, &p, sums.data(), &st)) != RT_OK) return die("rt_render", rc, rt_last_error_detail(ctx));
        }
        const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        // kernel_ms adds up the launches of all GPUs, which run side by side: the rate is quoted on the call
        std::fprintf(stderr, "%s, %dx%d, %d spp on %d GPU(s): call %.1f ms (%.1f Msamples/s, %.2f rays/sample), kernels %.1f ms\n", label,
                     p.width, p.height, p.sample_count, o.gpus, wall * 1e3, (double)st.samples / wall / 1e6,
                     (double)st.rays / (double)(st.samples ? st.samples : 1), st.kernel_ms);
        std::vector<uint8_t> rgb(sums.size());
        rt_write_color(sums.data(), p.width, p.height, p.sample_count, rgb.data());
        if ((rc = rt_write_ppm("/dev/stdout", rgb.data(), p.width, p.height)) != RT_OK) return die("rt_write_ppm", rc, "");
    }
    std::fprintf(stderr, "\nDone.\n");  // B1/RayTracing.cpp:129
    if (grp) rt_group_destroy(grp);
    else rt_context_destroy(ctx);
    return 0;
}

}  // namespace app
