// TEST INFRASTRUCTURE -- not part of the product.
//
// Drives the UNMODIFIED reference program of "Ray Tracing: The Next Week" (book 2; book 3's code
// directory is a byte-identical copy) to produce golden data.  The reference's translation unit
// is #included from where it lies under /root/reference (-I from the Makefile); nothing of it is
// copied.  g++ only (argument-evaluation order, see ref_b1_harness.cpp).
//
// Usage: ref_b2 <command> <scene> ...
//   scenes: bouncing (random_scene), two_spheres, two_perlin, earth (as shipped: the texture path
//           does not exist here, so the reference falls back to cyan), earth_real (same globe, the
//           image loaded from the reference tree by the reference's own loader), simple_light,
//           cornell_box, cornell_smoke, final
//   scene    <scene> <out.rtsc>
//   render   <scene> <W> <H> <spp> <depth> <srand|-> <out.f64>     the loop of RayTracing.cpp:363-377
//   firsthit <scene> <W> <H> <out.i32>     pixel-centre rays at shutter time 0, ids = index in world.objects
//   info     <scene>                        JSON: rand() calls made by the scene build, next rand()
#include <algorithm>
#include <cassert>
#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <emmintrin.h>
#include <fstream>
#include <iostream>
#include <limits>
#include <map>
#include <memory>
#include <random>
#include <string>
#include <utility>
#include <vector>

// The Perlin tables, solid_color::color_value and image_texture::data are private members; they are
// read here, never written.  All standard headers are included above, before the keyword is remapped.
#define private public
#define main reference_main
#include "RayTracing.cpp"
#undef main
#undef private

#include "rt_scene_io.h"

namespace {

struct scene_setup {
    hittable_list world;
    point3 lookfrom, lookat;
    double vfov = 40.0, aspect = 16.0 / 9.0, aperture = 0.0;
    color background = color(0, 0, 0);
    int width = 400, spp = 100;
};

// The switch of RayTracing.cpp:279-351, selected by name.
scene_setup make_setup(const std::string& which) {
    scene_setup s;
    const color sky(0.70, 0.80, 1.00);
    if (which == "bouncing") {
        s.world = random_scene(); s.background = sky; s.lookfrom = point3(13, 2, 3); s.lookat = point3(0, 0, 0);
        s.vfov = 20.0; s.aperture = 0.1;
    } else if (which == "two_spheres") {
        s.world = two_spheres(); s.background = sky; s.lookfrom = point3(13, 2, 3); s.lookat = point3(0, 0, 0); s.vfov = 20.0;
    } else if (which == "two_perlin") {
        s.world = two_perlin_spheres(); s.background = sky; s.lookfrom = point3(13, 2, 3); s.lookat = point3(0, 0, 0); s.vfov = 20.0;
    } else if (which == "earth") {
        s.world = earth(); s.background = sky; s.lookfrom = point3(0, 0, 12); s.lookat = point3(0, 0, 0); s.vfov = 20.0;
    } else if (which == "earth_real") {
        auto earth_texture = make_shared<image_texture>("/root/reference/Ray Tracing The Next Week/code/external/earthmap.jpg");
        s.world = hittable_list(make_shared<sphere>(point3(0, 0, 0), 2, make_shared<lambertian>(earth_texture)));
        s.background = sky; s.lookfrom = point3(0, 0, 12); s.lookat = point3(0, 0, 0); s.vfov = 20.0;
    } else if (which == "simple_light") {
        s.world = simple_light(); s.spp = 400; s.lookfrom = point3(26, 3, 6); s.lookat = point3(0, 2, 0); s.vfov = 20.0;
    } else if (which == "cornell_box") {
        s.world = cornell_box(); s.aspect = 1.0; s.width = 600; s.spp = 200;
        s.lookfrom = point3(278, 278, -800); s.lookat = point3(278, 278, 0); s.vfov = 40.0;
    } else if (which == "cornell_smoke") {
        s.world = cornell_smoke(); s.aspect = 1.0; s.width = 600; s.spp = 10;
        s.lookfrom = point3(278, 278, -800); s.lookat = point3(278, 278, 0); s.vfov = 40.0;
    } else if (which == "final") {
        s.world = final_scene(); s.aspect = 1.0; s.width = 800; s.spp = 10000;
        s.lookfrom = point3(478, 278, -600); s.lookat = point3(278, 278, 0); s.vfov = 40.0;
    } else {
        std::fprintf(stderr, "unknown scene '%s'\n", which.c_str());
        std::exit(2);
    }
    return s;
}

struct dumper {
    std::vector<rt_node> nodes;
    std::vector<int32_t> children;
    std::vector<rt_material> mats;
    std::vector<rt_texture> texs;
    std::vector<rt_perlin> perlins;
    std::vector<rt_image> images;
    std::vector<std::vector<uint8_t>> image_bytes;
    std::map<const material*, int> mat_ids;
    std::map<const texture*, int> tex_ids;

    int tex(const texture* t) {
        auto it = tex_ids.find(t);
        if (it != tex_ids.end()) return it->second;
        rt_texture r;
        std::memset(&r, 0, sizeof(r));
        r.a = r.b = -1;
        if (auto* s = dynamic_cast<const solid_color*>(t)) {
            r.kind = RT_TEX_SOLID;
            for (int c = 0; c < 3; ++c) r.p[c] = s->color_value[c];
        } else if (auto* ch = dynamic_cast<const checker_texture*>(t)) {
            r.kind = RT_TEX_CHECKER;
            r.a = tex(ch->even.get());
            r.b = tex(ch->odd.get());
        } else if (auto* n = dynamic_cast<const noise_texture*>(t)) {
            r.kind = RT_TEX_NOISE;
            rt_perlin p;
            for (int i = 0; i < 256; ++i) {
                for (int c = 0; c < 3; ++c) p.ranvec[i][c] = n->noise.ranvec[i][c];
                p.perm_x[i] = n->noise.perm_x[i];
                p.perm_y[i] = n->noise.perm_y[i];
                p.perm_z[i] = n->noise.perm_z[i];
            }
            perlins.push_back(p);
            r.a = (int)perlins.size() - 1;
            r.p[0] = n->scale;
        } else if (auto* im = dynamic_cast<const image_texture*>(t)) {
            r.kind = RT_TEX_IMAGE;
            if (im->data) {
                image_bytes.emplace_back(im->data, im->data + (size_t)3 * im->width * im->height);
                rt_image ri;
                ri.width = im->width;
                ri.height = im->height;
                ri.rgb = nullptr;  // fixed up before writing
                images.push_back(ri);
                r.a = (int)images.size() - 1;
            }
        } else {
            std::fprintf(stderr, "unknown texture class\n");
            std::exit(1);
        }
        texs.push_back(r);
        return tex_ids[t] = (int)texs.size() - 1;
    }

    int mat(const material* m) {
        auto it = mat_ids.find(m);
        if (it != mat_ids.end()) return it->second;
        rt_material r;
        std::memset(&r, 0, sizeof(r));
        r.texture = -1;
        if (auto* l = dynamic_cast<const lambertian*>(m)) {
            r.kind = RT_MAT_LAMBERTIAN;
            r.texture = tex(l->albedo.get());
        } else if (auto* me = dynamic_cast<const metal*>(m)) {
            r.kind = RT_MAT_METAL;
            for (int c = 0; c < 3; ++c) r.albedo[c] = me->albedo[c];
            r.param = me->fuzz;
        } else if (auto* d = dynamic_cast<const dielectric*>(m)) {
            r.kind = RT_MAT_DIELECTRIC;
            r.param = d->ir;
        } else if (auto* dl = dynamic_cast<const diffuse_light*>(m)) {
            r.kind = RT_MAT_DIFFUSE_LIGHT;
            r.texture = tex(dl->emit.get());
        } else if (auto* iso = dynamic_cast<const isotropic*>(m)) {
            r.kind = RT_MAT_ISOTROPIC;
            r.texture = tex(iso->albedo.get());
        } else {
            std::fprintf(stderr, "unknown material class\n");
            std::exit(1);
        }
        mats.push_back(r);
        return mat_ids[m] = (int)mats.size() - 1;
    }

    int push(int kind, int material) {
        rt_node n;
        std::memset(&n, 0, sizeof(n));
        n.kind = kind;
        n.material = material;
        n.child0 = n.child1 = -1;
        nodes.push_back(n);
        return (int)nodes.size() - 1;
    }

    int list(int kind, const std::vector<shared_ptr<hittable>>& objs) {
        const int self = push(kind, -1);
        std::vector<int32_t> kids;
        for (auto& o : objs) kids.push_back(emit(o.get()));
        nodes[self].child0 = (int32_t)children.size();
        nodes[self].child1 = (int32_t)kids.size();
        children.insert(children.end(), kids.begin(), kids.end());
        return self;
    }

    // One node per object; an object referenced from two places is emitted twice (the graph becomes a tree).
    int emit(const hittable* h) {
        if (auto* s = dynamic_cast<const sphere*>(h)) {
            const int i = push(RT_NODE_SPHERE, mat(s->mat_ptr.get()));
            for (int c = 0; c < 3; ++c) nodes[i].p[c] = s->center[c];
            nodes[i].p[3] = s->radius;
            return i;
        }
        if (auto* m = dynamic_cast<const moving_sphere*>(h)) {
            const int i = push(RT_NODE_MOVING_SPHERE, mat(m->mat_ptr.get()));
            for (int c = 0; c < 3; ++c) { nodes[i].p[c] = m->center0[c]; nodes[i].p[3 + c] = m->center1[c]; }
            nodes[i].p[6] = m->time0; nodes[i].p[7] = m->time1; nodes[i].p[8] = m->radius;
            return i;
        }
        if (auto* r = dynamic_cast<const xy_rect*>(h)) {
            const int i = push(RT_NODE_XY_RECT, mat(r->mp.get()));
            const double v[5] = {r->x0, r->x1, r->y0, r->y1, r->k};
            std::memcpy(nodes[i].p, v, sizeof(v));
            return i;
        }
        if (auto* r = dynamic_cast<const xz_rect*>(h)) {
            const int i = push(RT_NODE_XZ_RECT, mat(r->mp.get()));
            const double v[5] = {r->x0, r->x1, r->z0, r->z1, r->k};
            std::memcpy(nodes[i].p, v, sizeof(v));
            return i;
        }
        if (auto* r = dynamic_cast<const yz_rect*>(h)) {
            const int i = push(RT_NODE_YZ_RECT, mat(r->mp.get()));
            const double v[5] = {r->y0, r->y1, r->z0, r->z1, r->k};
            std::memcpy(nodes[i].p, v, sizeof(v));
            return i;
        }
        if (auto* b = dynamic_cast<const box*>(h)) {
            const int i = list(RT_NODE_BOX, b->sides.objects);
            for (int c = 0; c < 3; ++c) { nodes[i].p[c] = b->box_min[c]; nodes[i].p[3 + c] = b->box_max[c]; }
            return i;
        }
        if (auto* l = dynamic_cast<const hittable_list*>(h)) return list(RT_NODE_LIST, l->objects);
        if (auto* b = dynamic_cast<const bvh_node*>(h)) {
            const int i = push(RT_NODE_BVH, -1);
            for (int c = 0; c < 3; ++c) { nodes[i].p[c] = b->box.min()[c]; nodes[i].p[3 + c] = b->box.max()[c]; }
            const int l = emit(b->left.get());
            const int r = emit(b->right.get());
            nodes[i].child0 = l;
            nodes[i].child1 = r;
            return i;
        }
        if (auto* t = dynamic_cast<const translate*>(h)) {
            const int i = push(RT_NODE_TRANSLATE, -1);
            for (int c = 0; c < 3; ++c) nodes[i].p[c] = t->offset[c];
            const int k = emit(t->ptr.get());
            nodes[i].child0 = k;
            return i;
        }
        if (auto* r = dynamic_cast<const rotate_y*>(h)) {
            const int i = push(RT_NODE_ROTATE_Y, -1);
            nodes[i].p[0] = r->sin_theta; nodes[i].p[1] = r->cos_theta; nodes[i].p[2] = r->hasbox ? 1.0 : 0.0;
            for (int c = 0; c < 3; ++c) { nodes[i].p[3 + c] = r->bbox.min()[c]; nodes[i].p[6 + c] = r->bbox.max()[c]; }
            const int k = emit(r->ptr.get());
            nodes[i].child0 = k;
            return i;
        }
        if (auto* m = dynamic_cast<const constant_medium*>(h)) {
            const int i = push(RT_NODE_MEDIUM, mat(m->phase_function.get()));
            nodes[i].p[0] = m->neg_inv_density;
            const int k = emit(m->boundary.get());
            nodes[i].child0 = k;
            return i;
        }
        std::fprintf(stderr, "unknown hittable class\n");
        std::exit(1);
    }

    int write(const scene_setup& S, const char* path) {
        const int root = emit(&S.world);
        for (size_t i = 0; i < images.size(); ++i) images[i].rgb = image_bytes[i].data();
        rt_scene_desc d;
        std::memset(&d, 0, sizeof(d));
        d.abi_version = RT_ABI_VERSION;
        d.root = root;
        d.n_nodes = (int32_t)nodes.size(); d.n_children = (int32_t)children.size();
        d.n_materials = (int32_t)mats.size(); d.n_textures = (int32_t)texs.size();
        d.n_perlins = (int32_t)perlins.size(); d.n_images = (int32_t)images.size();
        d.nodes = nodes.data(); d.children = children.data(); d.materials = mats.data();
        d.textures = texs.data(); d.perlins = perlins.data(); d.images = images.data();
        d.sky_mode = RT_SKY_BACKGROUND;
        for (int c = 0; c < 3; ++c) d.background[c] = S.background[c];
        return rt_scene_file_write(path, &d);
    }
};

}  // namespace

int main(int argc, char** argv) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: see the head of ref_b2_harness.cpp\n");
        return 2;
    }
    const std::string cmd = argv[1];
    scene_setup S = make_setup(argv[2]);
    const vec3 vup(0, 1, 0);
    const double dist_to_focus = 10.0;

    if (cmd == "info") {
        int next = rand();
        srand(1);
        int calls = 0;
        while (rand() != next) ++calls;
        std::printf("{\"scene_build_rand_calls\": %d, \"next_rand\": %d, \"top_level_objects\": %zu, \"aspect\": %.17g, "
                    "\"vfov\": %.17g, \"aperture\": %.17g, \"lookfrom\": [%.17g, %.17g, %.17g], \"lookat\": [%.17g, %.17g, %.17g], "
                    "\"background\": [%.17g, %.17g, %.17g], \"width\": %d, \"spp\": %d}\n",
                    calls, next, S.world.objects.size(), S.aspect, S.vfov, S.aperture, S.lookfrom[0], S.lookfrom[1],
                    S.lookfrom[2], S.lookat[0], S.lookat[1], S.lookat[2], S.background[0], S.background[1], S.background[2],
                    S.width, S.spp);
        return 0;
    }
    if (cmd == "scene") {
        if (argc < 4) return 2;
        dumper d;
        return d.write(S, argv[3]);
    }
    if (cmd == "render") {
        if (argc < 9) return 2;
        const int image_width = std::atoi(argv[3]), image_height = std::atoi(argv[4]);
        const int samples_per_pixel = std::atoi(argv[5]), max_depth = std::atoi(argv[6]);
        if (std::strcmp(argv[7], "-") != 0) srand((unsigned)std::strtoul(argv[7], nullptr, 10));
        camera cam(S.lookfrom, S.lookat, vup, S.vfov, S.aspect, S.aperture, dist_to_focus, 0.0, 1.0);
        std::vector<double> sums((size_t)3 * image_width * image_height);
        size_t k = 0;
        for (int j = image_height - 1; j >= 0; --j) {
            for (int i = 0; i < image_width; ++i) {
                color pixel_color(0, 0, 0);
                for (int s = 0; s < samples_per_pixel; ++s) {
                    auto u = (i + random_double()) / (image_width - 1);
                    auto v = (j + random_double()) / (image_height - 1);
                    ray r = cam.get_ray(u, v);
                    pixel_color += ray_color(r, S.background, S.world, max_depth);
                }
                sums[k++] = pixel_color.x(); sums[k++] = pixel_color.y(); sums[k++] = pixel_color.z();
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
        const int W = std::atoi(argv[3]), H = std::atoi(argv[4]);
        camera cam(S.lookfrom, S.lookat, vup, S.vfov, S.aspect, S.aperture, dist_to_focus, 0.0, 1.0);
        std::vector<int32_t> ids((size_t)W * H);
        size_t k = 0;
        for (int j = H - 1; j >= 0; --j) {
            for (int i = 0; i < W; ++i) {
                const double s = (i + 0.5) / (W - 1), t = (j + 0.5) / (H - 1);
                vec3 offset = cam.u * 0.0 + cam.v * 0.0;
                ray r(cam.origin + offset, cam.lower_left_corner + s * cam.horizontal + t * cam.vertical - cam.origin - offset, 0.0);
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
