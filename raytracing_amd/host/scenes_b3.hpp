// The Cornell box of "Ray Tracing: The Rest of Your Life" (final listing of the public book text): the book-2
// box with the light wrapped in flip_face, the tall block, a glass sphere in place of the short block, and the
// list of shapes that light is sampled towards.  The reference repository holds no book-3 code (its third
// directory is a copy of book 2), so this builder follows the book, not a reference file.
#pragma once

#include "rtw.hpp"

namespace rtw {
namespace book3 {

struct cornell_scene {
    hittable_list world;
    hittable_list lights;
};

inline cornell_scene cornell_box() {
    cornell_scene s;
    hittable_list& objects = s.world;
    auto red = make_shared<lambertian>(color(.65, .05, .05));
    auto white = make_shared<lambertian>(color(.73, .73, .73));
    auto green = make_shared<lambertian>(color(.12, .45, .15));
    auto light = make_shared<diffuse_light>(color(15, 15, 15));

    objects.add(make_shared<yz_rect>(0, 555, 0, 555, 555, green));
    objects.add(make_shared<yz_rect>(0, 555, 0, 555, 0, red));
    objects.add(make_shared<flip_face>(make_shared<xz_rect>(213, 343, 227, 332, 554, light)));
    objects.add(make_shared<xz_rect>(0, 555, 0, 555, 555, white));
    objects.add(make_shared<xz_rect>(0, 555, 0, 555, 0, white));
    objects.add(make_shared<xy_rect>(0, 555, 0, 555, 555, white));

    shared_ptr<hittable> box1 = make_shared<box>(point3(0, 0, 0), point3(165, 330, 165), white);
    box1 = make_shared<rotate_y>(box1, 15);
    box1 = make_shared<translate>(box1, vec3(265, 0, 295));
    objects.add(box1);

    auto glass = make_shared<dielectric>(1.5);
    objects.add(make_shared<sphere>(point3(190, 90, 190), 90, glass));

    s.lights.add(make_shared<xz_rect>(213, 343, 227, 332, 554, shared_ptr<material>()));
    s.lights.add(make_shared<sphere>(point3(190, 90, 190), 90, shared_ptr<material>()));
    return s;
}

}  // namespace book3
}  // namespace rtw
