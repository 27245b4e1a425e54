// Counterpart of the reference's book-2 / book-3 program (B2/RayTracing.cpp:260-380): pick a scene the way
// its switch does, build it with the reference's own calls, render on the GPU through the C ABI, write P3.
//
//   rt_next_week [--scene N|name] [--width W] [--spp N] [--depth D] [--seed S] [--earth file.ppm]
//                [--gpus N] [--mode render|firsthit] [--dump-scene file.rtsc] [--no-render]  > image.ppm
//
// Scenes: 1 bouncing, 2 two_spheres, 3 two_perlin, 4 earth, 5 simple_light, 6 cornell_box, 7 cornell_smoke,
// 8 final (the reference's default).  Width / spp default to the reference's literals for that scene.
// The earth image is read as binary PPM; without one the texture is "missing" and renders cyan, which is
// also what the unmodified reference does outside its author's machine.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "raytracing_amd.h"
#include "rt_scene_io.h"
#include "app_common.hpp"
#include "../raytracing_amd/host/scenes_b2.hpp"

using namespace rtw;

int main(int argc, char** argv) {
    std::string which = "8", dump, earth_path = "earthmap.ppm";
    int width_arg = 0, spp_arg = 0, max_depth = 50;
    unsigned long long seed = 1;
    app::options common;
    bool render = true;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() { return i + 1 < argc ? argv[++i] : ""; };
        if (app::parse_common(common, argc, argv, i)) continue;
        if (a == "--scene") which = next();
        else if (a == "--width") width_arg = std::atoi(next());
        else if (a == "--spp") spp_arg = std::atoi(next());
        else if (a == "--depth") max_depth = std::atoi(next());
        else if (a == "--seed") seed = std::strtoull(next(), nullptr, 10);
        else if (a == "--earth") earth_path = next();
        else if (a == "--dump-scene") dump = next();
        else if (a == "--no-render") render = false;
        else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    static const char* names[] = {"", "bouncing", "two_spheres", "two_perlin", "earth", "simple_light", "cornell_box", "cornell_smoke", "final"};
    int scene_no = std::atoi(which.c_str());
    for (int k = 1; k <= 8; ++k) if (which == names[k]) scene_no = k;

    // B2/RayTracing.cpp:264-351
    double aspect_ratio = 16.0 / 9.0;
    int image_width = 400, samples_per_pixel = 100;
    hittable_list world;
    point3 lookfrom, lookat;
    double vfov = 40.0, aperture = 0.0;
    color background(0, 0, 0);
    const color sky(0.70, 0.80, 1.00);
    switch (scene_no) {
    case 1: world = book2::random_scene(); background = sky; lookfrom = point3(13, 2, 3); lookat = point3(0, 0, 0); vfov = 20.0; aperture = 0.1; break;
    case 2: world = book2::two_spheres(); background = sky; lookfrom = point3(13, 2, 3); lookat = point3(0, 0, 0); vfov = 20.0; break;
    case 3: world = book2::two_perlin_spheres(); background = sky; lookfrom = point3(13, 2, 3); lookat = point3(0, 0, 0); vfov = 20.0; break;
    case 4: world = book2::earth(earth_path.c_str()); background = sky; lookfrom = point3(0, 0, 12); lookat = point3(0, 0, 0); vfov = 20.0; break;
    case 5: world = book2::simple_light(); samples_per_pixel = 400; lookfrom = point3(26, 3, 6); lookat = point3(0, 2, 0); vfov = 20.0; break;
    case 6: world = book2::cornell_box(); aspect_ratio = 1.0; image_width = 600; samples_per_pixel = 200;
            lookfrom = point3(278, 278, -800); lookat = point3(278, 278, 0); vfov = 40.0; break;
    case 7: world = book2::cornell_smoke(); aspect_ratio = 1.0; image_width = 600; samples_per_pixel = 10;
            lookfrom = point3(278, 278, -800); lookat = point3(278, 278, 0); vfov = 40.0; break;
    default:
    case 8: world = book2::final_scene(earth_path.c_str()); aspect_ratio = 1.0; image_width = 800; samples_per_pixel = 10000;
            lookfrom = point3(478, 278, -600); lookat = point3(278, 278, 0); vfov = 40.0; break;
    }
    if (width_arg > 0) image_width = width_arg;
    if (spp_arg > 0) samples_per_pixel = spp_arg;
    const vec3 vup(0, 1, 0);
    const double dist_to_focus = 10.0;
    const int image_height = static_cast<int>(image_width / aspect_ratio);
    camera cam(lookfrom, lookat, vup, vfov, aspect_ratio, aperture, dist_to_focus, 0.0, 1.0);

    flattener flat;
    const rt_scene_desc scene = flat.flatten(world, RT_SKY_BACKGROUND, background);
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
    return app::run(common, scene, cam.desc(), p, names[scene_no >= 1 && scene_no <= 8 ? scene_no : 8]);
}
