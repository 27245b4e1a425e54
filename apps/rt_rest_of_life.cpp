// The main() of "Ray Tracing: The Rest of Your Life" (final listing of the public book text; the reference
// repository holds no book-3 code): Cornell box with a glass sphere, light sampled by importance, rendered on the
// GPU through the C ABI with RT_RENDER_IMPORTANCE, written as P3.
//
//   rt_rest_of_life [--width W] [--spp N] [--depth D] [--seed S] [--plain] [--gpus N] [--mode render|firsthit]
//                   [--dump-scene file.rtsc] [--no-render]
//
// --plain renders the same world without importance sampling (book-2 ray_color): both converge to the same image
// up to the one-sided light, which is what the tests use in place of a reference image.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "raytracing_amd.h"
#include "rt_scene_io.h"
#include "app_common.hpp"
#include "../raytracing_amd/host/scenes_b3.hpp"

using namespace rtw;

int main(int argc, char** argv) {
    int image_width = 600, samples_per_pixel = 1000, max_depth = 50;
    unsigned long long seed = 1;
    app::options common;
    bool render = true, plain = false;
    std::string dump;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() { return i + 1 < argc ? argv[++i] : ""; };
        if (app::parse_common(common, argc, argv, i)) continue;
        if (a == "--width") image_width = std::atoi(next());
        else if (a == "--spp") samples_per_pixel = std::atoi(next());
        else if (a == "--depth") max_depth = std::atoi(next());
        else if (a == "--seed") seed = std::strtoull(next(), nullptr, 10);
        else if (a == "--plain") plain = true;
        else if (a == "--dump-scene") dump = next();
        else if (a == "--no-render") render = false;
        else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    const double aspect_ratio = 1.0 / 1.0;
    const int image_height = static_cast<int>(image_width / aspect_ratio);
    book3::cornell_scene sc = book3::cornell_box();
    const color background(0, 0, 0);
    const point3 lookfrom(278, 278, -800), lookat(278, 278, 0);
    const vec3 vup(0, 1, 0);
    camera cam(lookfrom, lookat, vup, 40.0, aspect_ratio, 0.0, 10.0, 0.0, 1.0);

    flattener flat;
    const rt_scene_desc scene = flat.flatten(sc.world, RT_SKY_BACKGROUND, background, &sc.lights);
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
    p.flags = plain ? 0 : RT_RENDER_IMPORTANCE;
    return app::run(common, scene, cam.desc(), p, plain ? "cornell box, plain" : "cornell box, light sampled by importance");
}
