// TEST INFRASTRUCTURE -- not part of the product.
//
// Drives the UNMODIFIED reference program of "Ray Tracing In a Weekend" (book 1) to produce
// golden data.  The reference's single translation unit is #included from where it lies
// under /root/reference (the Makefile passes its directory with -I); nothing of it is
// copied.  Built only in the development container, into oracle/_ref/.
//
// Always build with g++ (not clang): the reference rolls its scene with rand() inside
// function-argument lists, whose evaluation order differs between the two compilers.
//
// Usage: ref_b1 <command> ...
//   scene   <c1|c2> <out.rtsc>                       dump the scene as an RTSCENE1 file
//   render  <c1|c2> <W> <H> <spp> <depth> <srand|-> <out.f64> [out.ppm]
//           the exact loop of RayTracing.cpp:113-127; raw double colour sums (row 0 = top)
//           and, optionally, the P3 stream the reference would print.  "-" = do not call
//           srand (continue the stream after the scene build, as the reference does).
//   firsthit <c1|c2> <W> <H> <out.i32>               pixel-centre closest-hit object ids
//   kat3                                              facts about the rolled c2 scene (JSON)
#define main reference_main
#include "RayTracing.cpp"
#undef main

#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "rt_scene_io.h"

namespace {

struct scene_setup {
    hittable_list world;
    point3 lookfrom, lookat;
    double vfov, aspect, aperture, focus;
};

// c1: the three-sphere scene kept as a comment in RayTracing.cpp:92-101, with the camera of
//     README.md:2354-2360 (lookfrom (3,3,2), lookat (0,0,-1), vfov 20, aperture 2, focus on lookat).
// c2: random_scene() + the camera of RayTracing.cpp:104-110.
scene_setup make_setup(const std::string& which) {
    scene_setup s;
    if (which == "c1") {
        auto material_ground = make_shared<lambertian>(color(0.8, 0.8, 0.0));
        auto material_center = make_shared<lambertian>(color(0.1, 0.2, 0.5));
        auto material_left = make_shared<dielectric>(1.5);
        auto material_right = make_shared<metal>(color(0.8, 0.6, 0.2), 0.0);
        s.world.add(make_shared<sphere>(point3(0.0, -100.5, -1.0), 100.0, material_ground));
        s.world.add(make_shared<sphere>(point3(0.0, 0.0, -1.0), 0.5, material_center));
        s.world.add(make_shared<sphere>(point3(-1.0, 0.0, -1.0), 0.5, material_left));
        s.world.add(make_shared<sphere>(point3(-1.0, 0.0, -1.0), -0.45, material_left));
        s.world.add(make_shared<sphere>(point3(1.0, 0.0, -1.0), 0.5, material_right));
        s.lookfrom = point3(3, 3, 2);
        s.lookat = point3(0, 0, -1);
        s.vfov = 20.0;
        s.aspect = 16.0 / 9.0;
        s.aperture = 2.0;
        s.focus = (s.lookfrom - s.lookat).length();
    } else if (which == "c2") {
        s.world = random_scene();
        s.lookfrom = point3(12, 2, 3);
        s.lookat = point3(0, 0, 0);
        s.vfov = 20.0;
        s.aspect = 3.0 / 2.0;
        s.aperture = 0.1;
        s.focus = 10.0;
    } else {
        std::fprintf(stderr, "unknown scene '%s'\n", which.c_str());
        std::exit(2);
    }
    return s;
}

int dump_scene(const hittable_list& world, const char* path) {
    std::vector<rt_node> nodes;
    std::vector<int32_t> children;
    std::vector<rt_material> mats;
    std::vector<rt_texture> texs;
    std::vector<const material*> seen;

    rt_node root;
    std::memset(&root, 0, sizeof(root));
    root.kind = RT_NODE_LIST;
    root.material = -1;
    nodes.push_back(root);

    for (const auto& obj : world.objects) {
        const sphere* sp = dynamic_cast<const sphere*>(obj.get());
        if (!sp) {
            std::fprintf(stderr, "non-sphere object in a book-1 scene\n");
            return 1;
        }
        const material* m = sp->mat_ptr.get();
        int mi = -1;
        for (size_t k = 0; k < seen.size(); ++k)
            if (seen[k] == m) mi = (int)k;
        if (mi < 0) {
            rt_material rm;
            std::memset(&rm, 0, sizeof(rm));
            rm.texture = -1;
            if (auto* l = dynamic_cast<const lambertian*>(m)) {
                rt_texture t;
                std::memset(&t, 0, sizeof(t));
                t.kind = RT_TEX_SOLID;
                t.a = t.b = -1;
                for (int c = 0; c < 3; ++c) t.p[c] = l->albedo[c];
                texs.push_back(t);
                rm.kind = RT_MAT_LAMBERTIAN;
                rm.texture = (int)texs.size() - 1;  // the colour lives in the texture (as in book 2)
            } else if (auto* me = dynamic_cast<const metal*>(m)) {
                rm.kind = RT_MAT_METAL;
                for (int c = 0; c < 3; ++c) rm.albedo[c] = me->albedo[c];
                rm.param = me->fuzz;
            } else if (auto* d = dynamic_cast<const dielectric*>(m)) {
                rm.kind = RT_MAT_DIELECTRIC;
                rm.param = d->ir;
            } else {
                std::fprintf(stderr, "unknown material class\n");
                return 1;
            }
            mats.push_back(rm);
            seen.push_back(m);
            mi = (int)mats.size() - 1;
        }
        rt_node n;
        std::memset(&n, 0, sizeof(n));
        n.kind = RT_NODE_SPHERE;
        n.material = mi;
        n.child0 = n.child1 = -1;
        for (int c = 0; c < 3; ++c) n.p[c] = sp->center[c];
        n.p[3] = sp->radius;
        nodes.push_back(n);
        children.push_back((int32_t)nodes.size() - 1);
    }
    nodes[0].child0 = 0;
    nodes[0].child1 = (int32_t)children.size();

    rt_scene_desc d;
    std::memset(&d, 0, sizeof(d));
    d.abi_version = RT_ABI_VERSION;
    d.root = 0;
    d.n_nodes = (int32_t)nodes.size();
    d.n_children = (int32_t)children.size();
    d.n_materials = (int32_t)mats.size();
    d.n_textures = (int32_t)texs.size();
    d.nodes = nodes.data();
    d.children = children.data();
    d.materials = mats.data();
    d.textures = texs.data();
    d.sky_mode = RT_SKY_GRADIENT;
    return rt_scene_file_write(path, &d);
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 2) {
        std::fprintf(stderr, "usage: see the head of ref_b1_harness.cpp\n");
        return 2;
    }
    std::string cmd = argv[1];

    if (cmd == "kat3") {
        hittable_list world = random_scene();
        int next = rand();
        int nl = 0, nm = 0, nd = 0;
        for (auto& o : world.objects) {
            auto* sp = dynamic_cast<sphere*>(o.get());
            if (dynamic_cast<lambertian*>(sp->mat_ptr.get())) ++nl;
            if (dynamic_cast<metal*>(sp->mat_ptr.get())) ++nm;
            if (dynamic_cast<dielectric*>(sp->mat_ptr.get())) ++nd;
        }
        auto* s1 = dynamic_cast<sphere*>(world.objects[1].get());
        auto* s2 = dynamic_cast<sphere*>(world.objects[2].get());
        std::printf("{\"objects\": %zu, \"lambertian\": %d, \"metal\": %d, \"dielectric\": %d,\n",
                    world.objects.size(), nl, nm, nd);
        std::printf(" \"obj1_center\": [%.17g, %.17g, %.17g],\n", s1->center[0], s1->center[1], s1->center[2]);
        std::printf(" \"obj2_center\": [%.17g, %.17g, %.17g],\n", s2->center[0], s2->center[1], s2->center[2]);
        // how many rand() calls random_scene() made: replay the default-seeded stream
        srand(1);
        int calls = 0;
        while (rand() != next) ++calls;
        std::printf(" \"scene_build_rand_calls\": %d,\n", calls);
        std::printf(" \"next_rand\": %d}\n", next);
        return 0;
    }

    if (argc < 3) return 2;
    scene_setup S = make_setup(argv[2]);

    if (cmd == "scene") {
        if (argc < 4) return 2;
        return dump_scene(S.world, argv[3]);
    }

    if (cmd == "render") {
        if (argc < 9) return 2;
        const int image_width = std::atoi(argv[3]);
        const int image_height = std::atoi(argv[4]);
        const int samples_per_pixel = std::atoi(argv[5]);
        const int max_depth = std::atoi(argv[6]);
        if (std::strcmp(argv[7], "-") != 0) srand((unsigned)std::strtoul(argv[7], nullptr, 10));
        camera cam(S.lookfrom, S.lookat, vec3(0, 1, 0), S.vfov, S.aspect, S.aperture, S.focus);

        std::vector<double> sums((size_t)3 * image_width * image_height);
        std::ofstream ppm;
        if (argc >= 10) {
            ppm.open(argv[9], std::ios::binary);
            ppm << "P3\n" << image_width << " " << image_height << "\n255\n";
        }
        size_t k = 0;
        for (int j = image_height - 1; j >= 0; --j) {
            for (int i = 0; i < image_width; ++i) {
                color pixel_color(0, 0, 0);
                for (int s = 0; s < samples_per_pixel; ++s) {
                    auto u = (i + random_double()) / (image_width - 1);
                    auto v = (j + random_double()) / (image_height - 1);
                    ray r = cam.get_ray(u, v);
                    pixel_color += ray_color(r, S.world, max_depth);
                }
                sums[k++] = pixel_color.x();
                sums[k++] = pixel_color.y();
                sums[k++] = pixel_color.z();
                if (ppm.is_open()) write_color(ppm, pixel_color, samples_per_pixel);
            }
        }
        int next = rand();
        FILE* fp = std::fopen(argv[8], "wb");
        if (!fp) return 1;
        std::fwrite(sums.data(), sizeof(double), sums.size(), fp);
        std::fclose(fp);
        std::printf("{\"next_rand\": %d}\n", next);
        return 0;
    }

    if (cmd == "firsthit") {
        if (argc < 6) return 2;
        const int W = std::atoi(argv[3]);
        const int H = std::atoi(argv[4]);
        // Camera members are private: recompute the constructor's values (camera.h:17-31).
        auto theta = degrees_to_radians(S.vfov);
        auto h = tan(theta / 2);
        auto viewport_height = 2.0 * h;
        auto viewport_width = S.aspect * viewport_height;
        vec3 w = unit_vector(S.lookfrom - S.lookat);
        vec3 u = unit_vector(cross(vec3(0, 1, 0), w));
        vec3 v = cross(w, u);
        point3 origin = S.lookfrom;
        vec3 horizontal = S.focus * viewport_width * u;
        vec3 vertical = S.focus * viewport_height * v;
        point3 llc = origin - horizontal / 2 - vertical / 2 - S.focus * w;

        std::vector<int32_t> ids((size_t)W * H);
        size_t k = 0;
        for (int j = H - 1; j >= 0; --j) {
            for (int i = 0; i < W; ++i) {
                double s = (i + 0.5) / (W - 1);
                double t = (j + 0.5) / (H - 1);
                vec3 offset = u * 0.0 + v * 0.0;
                ray r(origin + offset, llc + s * horizontal + t * vertical - origin - offset);
                // hittable_list::hit (hittable_list.h:26-39), keeping the winner's index.
                hit_record tmp;
                double closest = infinity;
                int32_t id = -1;
                for (size_t o = 0; o < S.world.objects.size(); ++o) {
                    if (S.world.objects[o]->hit(r, 0.001, closest, tmp)) {
                        closest = tmp.t;
                        id = (int32_t)o;
                    }
                }
                ids[k++] = id;
            }
        }
        FILE* fp = std::fopen(argv[5], "wb");
        if (!fp) return 1;
        std::fwrite(ids.data(), sizeof(int32_t), ids.size(), fp);
        std::fclose(fp);
        return 0;
    }

    std::fprintf(stderr, "unknown command\n");
    return 2;
}
