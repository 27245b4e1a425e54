// Scene builders of book 1, written against the recording layer rtw.hpp.
//
//   random_scene()        B1/RayTracing.cpp:33-77   (config 2: the final random-spheres scene)
//   three_sphere_scene()  B1/RayTracing.cpp:92-101  (config 1: kept there as a comment block)
//
// The reference draws random numbers inside argument lists, whose evaluation order C++14
// leaves to the compiler; its canonical build (g++) evaluates arguments last to first.  The
// draws are sequenced explicitly here, so this file rolls the canonical 487-object scene
// under any compiler: tests compare its output byte for byte with the scene dumped from the
// reference program.
#pragma once

#include "rtw.hpp"

namespace rtw {

inline hittable_list random_scene() {
    hittable_list world;

    auto ground_material = make_shared<lambertian>(color(0.5, 0.5, 0.5));
    world.add(make_shared<sphere>(point3(0, -1000, 0), 1000, ground_material));

    for (int a = -11; a < 11; ++a) {
        for (int b = -11; b < 11; ++b) {
            const double choose_mat = random_double();
            // point3 center(a + 0.9*random_double(), 0.2, b + 0.9*random_double()): z's draw comes first
            const double rz = random_double();
            const double rx = random_double();
            const point3 center(a + 0.9 * rx, 0.2, b + 0.9 * rz);

            if ((center - point3(4, 0.2, 0)).length() > 0.9) {
                shared_ptr<material> sphere_material;
                if (choose_mat < 0.8) {
                    // color::random() * color::random(): the right operand is rolled first
                    const color right = color::random();
                    const color left = color::random();
                    sphere_material = make_shared<lambertian>(left * right);
                } else if (choose_mat < 0.95) {
                    const color albedo = color::random(0.5, 1);
                    const double fuzz = random_double(0, 0.5);
                    sphere_material = make_shared<metal>(albedo, fuzz);
                } else {
                    sphere_material = make_shared<dielectric>(1.5);
                }
                world.add(make_shared<sphere>(center, 0.2, sphere_material));
            }
        }
    }

    world.add(make_shared<sphere>(point3(0, 1, 0), 1.0, make_shared<dielectric>(1.5)));
    world.add(make_shared<sphere>(point3(-4, 1, 0), 1.0, make_shared<lambertian>(color(0.4, 0.2, 0.1))));
    world.add(make_shared<sphere>(point3(4, 1, 0), 1.0, make_shared<metal>(color(0.7, 0.6, 0.5), 0.0)));
    return world;
}

inline hittable_list three_sphere_scene() {
    hittable_list world;
    auto material_ground = make_shared<lambertian>(color(0.8, 0.8, 0.0));
    auto material_center = make_shared<lambertian>(color(0.1, 0.2, 0.5));
    auto material_left = make_shared<dielectric>(1.5);
    auto material_right = make_shared<metal>(color(0.8, 0.6, 0.2), 0.0);
    world.add(make_shared<sphere>(point3(0.0, -100.5, -1.0), 100.0, material_ground));
    world.add(make_shared<sphere>(point3(0.0, 0.0, -1.0), 0.5, material_center));
    world.add(make_shared<sphere>(point3(-1.0, 0.0, -1.0), 0.5, material_left));
    world.add(make_shared<sphere>(point3(-1.0, 0.0, -1.0), -0.45, material_left));
    world.add(make_shared<sphere>(point3(1.0, 0.0, -1.0), 0.5, material_right));
    return world;
}

}  // namespace rtw
