// Counterpart of the reference's book-1 program (B1/RayTracing.cpp:79-132): build the world and
// the camera with the reference's own calls, hand both to the GPU library through the C ABI,
// write the P3 image to stdout.
//
//   rt_one_weekend [--scene final|three] [--width W] [--spp N] [--depth D] [--seed S]
//                  [--gpus N] [--mode render|firsthit] [--dump-scene file.rtsc] [--no-render]  > image.ppm
//
// Defaults are the reference's literals: 1200x800, 500 spp, depth 50, final scene.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "raytracing_amd.h"
#include "rt_scene_io.h"
#include "app_common.hpp"
#include "../raytracing_amd/host/scenes_b1.hpp"

using namespace rtw;

int main(int argc, char** argv) {
    std::string which = "final", dump;
    int image_width = 1200, samples_per_pixel = 500, max_depth = 50;
    unsigned long long seed = 1;
    app::options common;
    bool render = true;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() { return i + 1 < argc ? argv[++i] : ""; };
        if (app::parse_common(common, argc, argv, i)) continue;
        if (a == "--scene") which = next();
        else if (a == "--width") image_width = std::atoi(next());
        else if (a == "--spp") samples_per_pixel = std::atoi(next());
        else if (a == "--depth") max_depth = std::atoi(next());
        else if (a == "--seed") seed = std::strtoull(next(), nullptr, 10);
        else if (a == "--dump-scene") dump = next();
        else if (a == "--no-render") render = false;
        else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }

    // Image, World, Camera: B1/RayTracing.cpp:82-110 (final) / README.md:2354-2360 (three spheres)
    double aspect_ratio;
    hittable_list world;
    point3 lookfrom, lookat;
    double dist_to_focus, aperture;
    if (which == "final") {
        aspect_ratio = 3.0 / 2.0;
        world = random_scene();
        lookfrom = point3(12, 2, 3);
        lookat = point3(0, 0, 0);
        dist_to_focus = 10.0;
        aperture = 0.1;
    } else if (which == "three") {
        aspect_ratio = 16.0 / 9.0;
        world = three_sphere_scene();
        lookfrom = point3(3, 3, 2);
        lookat = point3(0, 0, -1);
        dist_to_focus = (lookfrom - lookat).length();
        aperture = 2.0;
    } else {
        std::fprintf(stderr, "unknown scene %s\n", which.c_str());
        return 2;
    }
    const int image_height = static_cast<int>(image_width / aspect_ratio);
    camera cam(lookfrom, lookat, vec3(0, 1, 0), 20.0, aspect_ratio, aperture, dist_to_focus);

    flattener flat;
    const rt_scene_desc scene = flat.flatten(world);
    if (!dump.empty() && rt_scene_file_write(dump.c_str(), &scene) != RT_OK) {
        std::fprintf(stderr, "cannot write %s\n", dump.c_str());
        return 1;
    }
    if (!render) return 0;

    rt_render_params p;
    std::memset(&p, 0, sizeof(p));
    p.width = image_width;
    p.height = image_height;
    p.sample_begin = 0;
    p.sample_count = samples_per_pixel;
    p.max_depth = max_depth;
    p.seed = seed;
    return app::run(common, scene, cam.desc(), p, which.c_str());
}
