/*
 * rt_oracle.c -- TEST INFRASTRUCTURE.  CPU restatement of the reference path tracer.
 *
 * This file is the parity checker of the repository, not a product code path: only the
 * tests, the smoke check and the cpu_baseline leg of the benchmark may load it.
 *
 * It restates, in plain C and IEEE double with the reference's operation order, the hot
 * path of huris/RayTracing: the sample loop, ray_color, every hittable::hit, every
 * material::scatter / emitted and every texture::value.  Citations: B1 = "Ray Tracing In
 * a Weekend/code", B2 = "Ray Tracing The Next Week/code" under /root/reference.
 *
 * Parity status: PINNED.  With rng_mode = ORC_RNG_RAND the restatement draws from the C
 * library's rand() in the reference's call order and reproduces the reference program bit
 * for bit (colour sums and P3 output of the config-1 scene, reduced config-2 renders,
 * first-hit ids; the fixtures under tests/golden are produced by the unmodified reference,
 * see oracle/make_golden.py).  With rng_mode = ORC_RNG_PHILOX the same code draws from
 * Philox4x32-10 keyed by (pixel, sample, site, block), which is what the GPU kernels do, so
 * GPU output can be compared number for number.
 *
 * One part is PARITY UNPINNED: ray_color_pdf and its helpers (RT_RENDER_IMPORTANCE, "The Rest
 * Of Your Life").  The reference repository holds no book-3 code, so they restate the public
 * book text and no output of the reference exists to check them against; tests/test_book3.py
 * checks them against plain path tracing of the same world instead.
 *
 * Build: gcc -O2 -ffp-contract=off (no -ffast-math, no -march): see oracle/Makefile.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "raytracing_amd.h"

#define ORC_RNG_RAND 0
#define ORC_RNG_PHILOX 1

#define ORC_ASSOC_REFERENCE 0 /* a1*(a2*(...*L)) as the recursion multiplies (B1/RayTracing.cpp:22) */
#define ORC_ASSOC_FORWARD 1   /* running throughput product, the order the GPU kernels use        */

#define ORC_MAX_DEPTH 4096

typedef struct { double e[3]; } v3;

/* ------------------------------------------------------------------ Philox4x32-10
 * Salmon et al., "Parallel random numbers: as easy as 1, 2, 3" (SC'11); constants of the
 * Random123 library.  Known-answer vectors are checked in tests/test_philox.py.          */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* ------------------------------------------------------------------ random source
 * rand mode:   every draw is rand() / (RAND_MAX + 1.0)  (B1/rtweekend.h:24-27), consumed
 *              in the order the reference compiled with g++ consumes them.
 * philox mode: a "call" selects the block (pixel, sample, site, block index); draw(slot)
 *              converts word `slot` of that block: (word >> 1) / 2^31, the same 31-bit grid.
 * Sites: 0 = camera ray; 1+b = ray segment b (its medium draws and the scatter after it).
 * Block indices: 0,1,2,.. feed the rejection-sampling attempts with no word wasted (all four
 *        words of a block are used: see get_ray and random_in_unit_sphere);
 *        ORC_TRY_TIME = shutter time; ORC_TRY_MEDIUM + i = the draw of the constant_medium
 *        whose node index is i.                                                          */
#define ORC_TRY_TIME 0x80000000u
#define ORC_TRY_MEDIUM 0x40000000u

typedef struct {
    int mode;
    uint32_t key[2];
    uint32_t pixel, sample;
    uint32_t out[4];
    uint32_t medium_draws; /* per segment */
    uint64_t n_draws;
} orc_rng;

static inline void rng_call(orc_rng* g, uint32_t site, uint32_t tri) {
    if (g->mode == ORC_RNG_PHILOX) {
        uint32_t ctr[4] = {g->pixel, g->sample, site, tri};
        orc_philox4x32_10(ctr, g->key, g->out);
    }
}
static inline double rng_draw(orc_rng* g, int slot) {
    g->n_draws++;
    if (g->mode == ORC_RNG_PHILOX) return (double)(g->out[slot] >> 1) / 2147483648.0;
    return rand() / (RAND_MAX + 1.0);
}
/* philox mode only: the draw a saved word stands for */
static inline double rng_draw_word(orc_rng* g, uint32_t word) {
    g->n_draws++;
    return (double)(word >> 1) / 2147483648.0;
}
/* random_double(MIN, MAX), B1/rtweekend.h:37-40 */
static inline double rng_draw_range(orc_rng* g, int slot, double lo, double hi) {
    return lo + (hi - lo) * rng_draw(g, slot);
}

/* ------------------------------------------------------------------ vec3 (B1/vec3.h) */
static inline v3 V(double x, double y, double z) { v3 r = {{x, y, z}}; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.e[0] + b.e[0], a.e[1] + b.e[1], a.e[2] + b.e[2]); }
static inline v3 vsub(v3 a, v3 b) { return V(a.e[0] - b.e[0], a.e[1] - b.e[1], a.e[2] - b.e[2]); }
static inline v3 vmul(v3 a, v3 b) { return V(a.e[0] * b.e[0], a.e[1] * b.e[1], a.e[2] * b.e[2]); }
static inline v3 vscale(double t, v3 a) { return V(t * a.e[0], t * a.e[1], t * a.e[2]); }
static inline v3 vneg(v3 a) { return V(-a.e[0], -a.e[1], -a.e[2]); }
/* operator/(vec3, double) multiplies by the reciprocal (B1/vec3.h:96-98) */
static inline v3 vdiv(v3 a, double t) { return vscale(1 / t, a); }
static inline double vdot(v3 a, v3 b) { return a.e[0] * b.e[0] + a.e[1] * b.e[1] + a.e[2] * b.e[2]; }
static inline double vlen2(v3 a) { return a.e[0] * a.e[0] + a.e[1] * a.e[1] + a.e[2] * a.e[2]; }
static inline double vlen(v3 a) { return sqrt(vlen2(a)); }
static inline v3 vunit(v3 a) { return vdiv(a, vlen(a)); }
static inline v3 vcross(v3 u, v3 v) {
    return V(u.e[1] * v.e[2] - u.e[2] * v.e[1], u.e[2] * v.e[0] - u.e[0] * v.e[2],
             u.e[0] * v.e[1] - u.e[1] * v.e[0]);
}
static inline int vnear_zero(v3 a) {
    const double s = 1e-8;
    return (fabs(a.e[0]) < s) && (fabs(a.e[1]) < s) && (fabs(a.e[2]) < s);
}

/* random_in_unit_sphere, B1/vec3.h:114-120.  vec3::random(-1,1) is
 * vec3(random_double(..), random_double(..), random_double(..)) (B1/vec3.h:52-54); g++
 * evaluates these arguments last to first, so the FIRST draw lands in z.                 */
/* Philox: attempts 4g, 4g+1, 4g+2 take words 0,1,2 (= x,y,z) of blocks 3g, 3g+1, 3g+2 of the
 * site; attempt 4g+3 takes the three words 3 those blocks left over (x from block 3g, y from
 * 3g+1, z from 3g+2): four attempts per three blocks.                                      */
static v3 random_in_unit_sphere(orc_rng* g, uint32_t site) {
    uint32_t spare[3] = {0, 0, 0};
    for (uint32_t k = 0;; ++k) {
        double x, y, z;
        if (g->mode == ORC_RNG_PHILOX && (k & 3u) == 3u) {
            z = -1 + (1 - -1) * rng_draw_word(g, spare[2]);
            y = -1 + (1 - -1) * rng_draw_word(g, spare[1]);
            x = -1 + (1 - -1) * rng_draw_word(g, spare[0]);
        } else {
            rng_call(g, site, 3u * (k >> 2) + (k & 3u));
            if (g->mode == ORC_RNG_PHILOX) spare[k & 3u] = g->out[3];
            z = rng_draw_range(g, 2, -1, 1);
            y = rng_draw_range(g, 1, -1, 1);
            x = rng_draw_range(g, 0, -1, 1);
        }
        v3 p = V(x, y, z);
        if (vlen2(p) >= 1) continue;
        return p;
    }
}
/* reflect / refract, B1/vec3.h:134-143 */
static inline v3 reflect(v3 v, v3 n) { return vsub(v, vscale(2 * vdot(v, n), n)); }
static inline v3 refract(v3 uv, v3 n, double etai_over_etat) {
    double cos_theta = fmin(vdot(vneg(uv), n), 1.0);
    v3 r_out_perp = vscale(etai_over_etat, vadd(uv, vscale(cos_theta, n)));
    v3 r_out_parallel = vscale(-sqrt(fabs(1.0 - vlen2(r_out_perp))), n);
    return vadd(r_out_perp, r_out_parallel);
}

typedef struct { v3 orig, dir; double tm; } ray_t;
static inline v3 ray_at(const ray_t* r, double t) { return vadd(r->orig, vscale(t, r->dir)); }

typedef struct {
    v3 p, normal;
    int mat;
    double t, u, v;
    int front_face;
} hit_record;

static inline void set_face_normal(hit_record* rec, const ray_t* r, v3 outward_normal) {
    rec->front_face = vdot(r->dir, outward_normal) < 0;
    rec->normal = rec->front_face ? outward_normal : vneg(outward_normal);
}

typedef struct {
    const rt_scene_desc* s;
    orc_rng* rng;
    uint32_t site; /* site of the segment being intersected */
} hit_ctx;

/* ------------------------------------------------------------------ hittables */
static int hit_node(hit_ctx* cx, int idx, const ray_t* r, double t_min, double t_max, hit_record* rec);

static const double ORC_PI = 3.14159265358979323846; /* == acos(-1.0) in double (B1/rtweekend.h:17) */

/* sphere::hit, B1/sphere.h:23-50 and B2/sphere.h:41-69 (the latter adds u,v) */
static int hit_sphere_at(v3 center, double radius, int mat, int want_uv, const ray_t* r, double t_min,
                         double t_max, hit_record* rec) {
    v3 oc = vsub(r->orig, center);
    double a = vlen2(r->dir);
    double half_b = vdot(oc, r->dir);
    double c = vlen2(oc) - radius * radius;
    double discriminant = half_b * half_b - a * c;
    if (discriminant < 0) return 0;
    double sqrtd = sqrt(discriminant);
    double root = (-half_b - sqrtd) / a;
    if (root < t_min || root > t_max) {
        root = (-half_b + sqrtd) / a;
        if (root < t_min || root > t_max) return 0;
    }
    rec->t = root;
    rec->p = ray_at(r, rec->t);
    v3 outward_normal = vdiv(vsub(rec->p, center), radius);
    set_face_normal(rec, r, outward_normal);
    if (want_uv) { /* get_sphere_uv, B2/sphere.h:25-38 */
        double theta = acos(-outward_normal.e[1]);
        double phi = atan2(-outward_normal.e[2], outward_normal.e[0]) + ORC_PI;
        rec->u = phi / (2 * ORC_PI);
        rec->v = theta / ORC_PI;
    }
    rec->mat = mat;
    return 1;
}

/* aabb::hit, B2/aabb.h:34-47 (1.0f / double is a double division) */
static int hit_aabb(const double* mn, const double* mx, const ray_t* r, double t_min, double t_max) {
    for (int a = 0; a < 3; ++a) {
        double invD = 1.0f / r->dir.e[a];
        double t0 = (mn[a] - r->orig.e[a]) * invD;
        double t1 = (mx[a] - r->orig.e[a]) * invD;
        if (invD < 0.0f) { double tmp = t0; t0 = t1; t1 = tmp; }
        t_min = t0 > t_min ? t0 : t_min;
        t_max = t1 < t_max ? t1 : t_max;
        if (t_max <= t_min) return 0;
    }
    return 1;
}

/* xy_rect / xz_rect / yz_rect ::hit, B2/aarect.h:28-46,70-88,112-130.
 * axis k: the constant axis; (a, b) the two in-plane axes in the reference's u, v order. */
static int hit_rect(const rt_node* n, int ka, int aa, int ba, const ray_t* r, double t_min, double t_max,
                    hit_record* rec) {
    double a0 = n->p[0], a1 = n->p[1], b0 = n->p[2], b1 = n->p[3], k = n->p[4];
    double t = (k - r->orig.e[ka]) / r->dir.e[ka];
    if (t < t_min || t > t_max) return 0;
    double a = r->orig.e[aa] + t * r->dir.e[aa];
    double b = r->orig.e[ba] + t * r->dir.e[ba];
    if (a < a0 || a > a1 || b < b0 || b > b1) return 0;
    rec->u = (a - a0) / (a1 - a0);
    rec->v = (b - b0) / (b1 - b0);
    rec->t = t;
    v3 outward_normal = V(0, 0, 0);
    outward_normal.e[ka] = 1;
    set_face_normal(rec, r, outward_normal);
    rec->mat = n->material;
    rec->p = ray_at(r, t);
    return 1;
}

/* hittable_list::hit, B1/hittable_list.h:26-39 */
static int hit_list(hit_ctx* cx, const int32_t* kids, int n, const ray_t* r, double t_min, double t_max,
                    hit_record* rec) {
    hit_record temp_rec = *rec; /* the reference's temp_rec starts uninitialised; carry u,v */
    int hit_anything = 0;
    double closest_so_far = t_max;
    for (int i = 0; i < n; ++i) {
        if (hit_node(cx, kids[i], r, t_min, closest_so_far, &temp_rec)) {
            hit_anything = 1;
            closest_so_far = temp_rec.t;
            *rec = temp_rec;
        }
    }
    return hit_anything;
}

static int hit_node(hit_ctx* cx, int idx, const ray_t* r, double t_min, double t_max, hit_record* rec) {
    const rt_scene_desc* s = cx->s;
    const rt_node* n = &s->nodes[idx];
    switch (n->kind) {
    case RT_NODE_SPHERE:
        return hit_sphere_at(V(n->p[0], n->p[1], n->p[2]), n->p[3], n->material, s->n_textures > 0, r, t_min,
                             t_max, rec);
    case RT_NODE_MOVING_SPHERE: { /* B2/moving_sphere.h:27-57.  The reference leaves rec.u, rec.v as they were:
                                   * indeterminate for a first hit, stale after another object's.  No reference
                                   * world reads them (its moving spheres have solid colours); pinned to 0 here
                                   * and in the kernels so that such worlds are well defined. */
        v3 c0 = V(n->p[0], n->p[1], n->p[2]), c1 = V(n->p[3], n->p[4], n->p[5]);
        double time0 = n->p[6], time1 = n->p[7];
        v3 center = vadd(c0, vscale((r->tm - time0) / (time1 - time0), vsub(c1, c0)));
        if (!hit_sphere_at(center, n->p[8], n->material, 0, r, t_min, t_max, rec)) return 0;
        rec->u = rec->v = 0.0;
        return 1;
    }
    case RT_NODE_XY_RECT: return hit_rect(n, 2, 0, 1, r, t_min, t_max, rec);
    case RT_NODE_XZ_RECT: return hit_rect(n, 1, 0, 2, r, t_min, t_max, rec);
    case RT_NODE_YZ_RECT: return hit_rect(n, 0, 1, 2, r, t_min, t_max, rec);
    case RT_NODE_BOX: /* B2/box.h:41-43 */
    case RT_NODE_LIST:
        return hit_list(cx, s->children + n->child0, n->child1, r, t_min, t_max, rec);
    case RT_NODE_BVH: { /* B2/bvh.h:96-103 */
        if (!hit_aabb(&n->p[0], &n->p[3], r, t_min, t_max)) return 0;
        int hit_left = hit_node(cx, n->child0, r, t_min, t_max, rec);
        int hit_right = hit_node(cx, n->child1, r, t_min, hit_left ? rec->t : t_max, rec);
        return hit_left || hit_right;
    }
    case RT_NODE_TRANSLATE: { /* B2/hittable.h:43-52 */
        v3 offset = V(n->p[0], n->p[1], n->p[2]);
        ray_t moved = {vsub(r->orig, offset), r->dir, r->tm};
        if (!hit_node(cx, n->child0, &moved, t_min, t_max, rec)) return 0;
        rec->p = vadd(rec->p, offset);
        set_face_normal(rec, &moved, rec->normal);
        return 1;
    }
    case RT_NODE_ROTATE_Y: { /* B2/hittable.h:113-140 */
        double sin_theta = n->p[0], cos_theta = n->p[1];
        v3 origin = r->orig, direction = r->dir;
        origin.e[0] = cos_theta * r->orig.e[0] - sin_theta * r->orig.e[2];
        origin.e[2] = sin_theta * r->orig.e[0] + cos_theta * r->orig.e[2];
        direction.e[0] = cos_theta * r->dir.e[0] - sin_theta * r->dir.e[2];
        direction.e[2] = sin_theta * r->dir.e[0] + cos_theta * r->dir.e[2];
        ray_t rotated = {origin, direction, r->tm};
        if (!hit_node(cx, n->child0, &rotated, t_min, t_max, rec)) return 0;
        v3 p = rec->p, normal = rec->normal;
        p.e[0] = cos_theta * rec->p.e[0] + sin_theta * rec->p.e[2];
        p.e[2] = -sin_theta * rec->p.e[0] + cos_theta * rec->p.e[2];
        normal.e[0] = cos_theta * rec->normal.e[0] + sin_theta * rec->normal.e[2];
        normal.e[2] = -sin_theta * rec->normal.e[0] + cos_theta * rec->normal.e[2];
        rec->p = p;
        set_face_normal(rec, &rotated, normal);
        return 1;
    }
    case RT_NODE_FLIP_FACE: /* flip_face::hit of book 3 (public text): the child's record, front_face negated */
        if (!hit_node(cx, n->child0, r, t_min, t_max, rec)) return 0;
        rec->front_face = !rec->front_face;
        return 1;
    case RT_NODE_MEDIUM: { /* B2/constant_medium.h:30-66 */
        hit_record rec1 = *rec, rec2 = *rec;
        if (!hit_node(cx, n->child0, r, -INFINITY, INFINITY, &rec1)) return 0;
        if (!hit_node(cx, n->child0, r, rec1.t + 0.0001, INFINITY, &rec2)) return 0;
        if (rec1.t < t_min) rec1.t = t_min;
        if (rec2.t > t_max) rec2.t = t_max;
        if (rec1.t >= rec2.t) return 0;
        if (rec1.t < 0) rec1.t = 0;
        const double ray_length = vlen(r->dir);
        const double distance_inside_boundary = (rec2.t - rec1.t) * ray_length;
        /* Philox: keyed by the medium's node index, so that the draw does not depend on the order in
         * which a traversal meets the media of a segment */
        rng_call(cx->rng, cx->site, ORC_TRY_MEDIUM + (uint32_t)idx);
        cx->rng->medium_draws++;
        const double hit_distance = n->p[0] * log(rng_draw(cx->rng, 0));
        if (hit_distance > distance_inside_boundary) return 0;
        rec->t = rec1.t + hit_distance / ray_length;
        rec->p = ray_at(r, rec->t);
        rec->normal = V(1, 0, 0);
        rec->front_face = 1;
        rec->mat = n->material;
        rec->u = rec->v = 0.0; /* left as they were by the reference, like moving_sphere: pinned to 0 */
        return 1;
    }
    default:
        return 0;
    }
}

/* ------------------------------------------------------------------ textures (B2/texture.h, B2/perlin.h) */
static double perlin_noise(const rt_perlin* pn, v3 p) { /* B2/perlin.h:24-48,86-104 */
    double u = p.e[0] - floor(p.e[0]);
    double v = p.e[1] - floor(p.e[1]);
    double w = p.e[2] - floor(p.e[2]);
    int i = (int)floor(p.e[0]);
    int j = (int)floor(p.e[1]);
    int k = (int)floor(p.e[2]);
    double uu = u * u * (3 - 2 * u);
    double vv = v * v * (3 - 2 * v);
    double ww = w * w * (3 - 2 * w);
    double accum = 0.0;
    for (int di = 0; di < 2; ++di)
        for (int dj = 0; dj < 2; ++dj)
            for (int dk = 0; dk < 2; ++dk) {
                const double* c =
                    pn->ranvec[pn->perm_x[(i + di) & 255] ^ pn->perm_y[(j + dj) & 255] ^ pn->perm_z[(k + dk) & 255]];
                v3 weight_v = V(u - di, v - dj, w - dk);
                accum += (di * uu + (1 - di) * (1 - uu)) * (dj * vv + (1 - dj) * (1 - vv)) *
                         (dk * ww + (1 - dk) * (1 - ww)) * vdot(V(c[0], c[1], c[2]), weight_v);
            }
    return accum;
}
static double perlin_turb(const rt_perlin* pn, v3 p) { /* B2/perlin.h:50-61, depth = 7 */
    double accum = 0.0;
    v3 temp_p = p;
    double weight = 1.0;
    for (int i = 0; i < 7; ++i) {
        accum += weight * perlin_noise(pn, temp_p);
        weight *= 0.5;
        temp_p = V(temp_p.e[0] * 2, temp_p.e[1] * 2, temp_p.e[2] * 2);
    }
    return fabs(accum);
}
static double clampd(double x, double lo, double hi) { /* B1/rtweekend.h:42-46 */
    if (x < lo) return lo;
    if (x > hi) return hi;
    return x;
}
static v3 texture_value(const rt_scene_desc* s, int ti, double u, double v, v3 p) {
    const rt_texture* t = &s->textures[ti];
    switch (t->kind) {
    case RT_TEX_SOLID: return V(t->p[0], t->p[1], t->p[2]);
    case RT_TEX_CHECKER: { /* B2/texture.h:37-41 */
        double sines = sin(10 * p.e[0]) * sin(10 * p.e[1]) * sin(10 * p.e[2]);
        return texture_value(s, sines < 0 ? t->b : t->a, u, v, p);
    }
    case RT_TEX_NOISE: { /* B2/texture.h:53-55 */
        double f = 1 + sin(t->p[0] * p.e[2] + 10 * perlin_turb(&s->perlins[t->a], p));
        return vscale(f, vscale(0.5, V(1, 1, 1)));
    }
    case RT_TEX_IMAGE: { /* B2/texture.h:85-104 */
        if (t->a < 0 || s->images[t->a].rgb == NULL) return V(0, 1, 1);
        const rt_image* im = &s->images[t->a];
        u = clampd(u, 0.0, 1.0);
        v = 1.0 - clampd(v, 0.0, 1.0);
        int i = (int)(u * im->width);
        int j = (int)(v * im->height);
        if (i >= im->width) i = im->width - 1;
        if (j >= im->height) j = im->height - 1;
        const double color_scale = 1.0 / 255.0;
        const uint8_t* px = im->rgb + (size_t)j * 3 * (size_t)im->width + (size_t)i * 3;
        return V(color_scale * px[0], color_scale * px[1], color_scale * px[2]);
    }
    default: return V(0, 0, 0);
    }
}

/* ------------------------------------------------------------------ materials */
static double reflectance(double cosine, double ref_idx) { /* B1/material.h:83-88 */
    double r0 = (1 - ref_idx) / (1 + ref_idx);
    r0 = r0 * r0;
    return r0 + (1 - r0) * pow((1 - cosine), 5);
}

/* material::scatter of the five classes (B1/material.h:19-77, B2/material.h:25-131) */
static int scatter(const rt_scene_desc* s, orc_rng* g, uint32_t site, const ray_t* r_in, const hit_record* rec,
                   v3* attenuation, ray_t* scattered) {
    const rt_material* m = &s->materials[rec->mat];
    switch (m->kind) {
    case RT_MAT_LAMBERTIAN: {
        v3 scatter_direction = vadd(rec->normal, vunit(random_in_unit_sphere(g, site)));
        if (vnear_zero(scatter_direction)) scatter_direction = rec->normal;
        scattered->orig = rec->p;
        scattered->dir = scatter_direction;
        scattered->tm = r_in->tm;
        *attenuation = texture_value(s, m->texture, rec->u, rec->v, rec->p);
        return 1;
    }
    case RT_MAT_METAL: {
        v3 reflected = reflect(vunit(r_in->dir), rec->normal);
        scattered->orig = rec->p;
        scattered->dir = vadd(reflected, vscale(m->param, random_in_unit_sphere(g, site)));
        scattered->tm = r_in->tm;
        *attenuation = V(m->albedo[0], m->albedo[1], m->albedo[2]);
        return vdot(scattered->dir, rec->normal) > 0;
    }
    case RT_MAT_DIELECTRIC: {
        *attenuation = V(1.0, 1.0, 1.0);
        double refraction_ratio = rec->front_face ? (1.0 / m->param) : m->param;
        v3 unit_direction = vunit(r_in->dir);
        double cos_theta = fmin(vdot(vneg(unit_direction), rec->normal), 1.0);
        double sin_theta = sqrt(1.0 - cos_theta * cos_theta);
        int cannot_refract = refraction_ratio * sin_theta > 1.0;
        v3 direction;
        int do_reflect = cannot_refract;
        if (!do_reflect) { /* the draw happens only when refraction is possible (short-circuit ||) */
            rng_call(g, site, 0);
            do_reflect = reflectance(cos_theta, refraction_ratio) > rng_draw(g, 0);
        }
        if (do_reflect) direction = reflect(unit_direction, rec->normal);
        else direction = refract(unit_direction, rec->normal, refraction_ratio);
        scattered->orig = rec->p;
        scattered->dir = direction;
        scattered->tm = r_in->tm;
        return 1;
    }
    case RT_MAT_DIFFUSE_LIGHT: return 0;
    case RT_MAT_ISOTROPIC: {
        scattered->orig = rec->p;
        scattered->dir = random_in_unit_sphere(g, site);
        scattered->tm = r_in->tm;
        *attenuation = texture_value(s, m->texture, rec->u, rec->v, rec->p);
        return 1;
    }
    default: return 0;
    }
}
static v3 emitted(const rt_scene_desc* s, const hit_record* rec) { /* B2/material.h:15-17,108-110 */
    const rt_material* m = &s->materials[rec->mat];
    if (m->kind == RT_MAT_DIFFUSE_LIGHT) return texture_value(s, m->texture, rec->u, rec->v, rec->p);
    return V(0, 0, 0);
}

/* ------------------------------------------------------------------ ray_color
 * B1/RayTracing.cpp:11-31 and B2/RayTracing.cpp:17-36, unrolled into a loop.  The
 * recursion computes  e1 + a1*(e2 + a2*( ... L_end)); ORC_ASSOC_REFERENCE replays exactly
 * that from a stack, ORC_ASSOC_FORWARD keeps L += T*e, T *= a.                           */
static v3 ray_color(const rt_scene_desc* s, orc_rng* g, ray_t r, int depth, int assoc, uint64_t* rays) {
    static _Thread_local v3 att_stack[ORC_MAX_DEPTH], emi_stack[ORC_MAX_DEPTH];
    int n = 0;
    v3 L_end = V(0, 0, 0);
    v3 T = V(1, 1, 1), L = V(0, 0, 0);
    /* the sky mode only selects what a miss returns (book 1's gradient or book 2's background); lights emit under
     * either.  A world without lights (every book-1 world) skips the "+ emitted" of B2/RayTracing.cpp:35, which
     * would add +0.0 and could turn a -0.0 of book 1's product chain into +0.0 */
    int emissive_scene = s->sky_mode == RT_SKY_BACKGROUND;
    for (int k = 0; k < s->n_materials && !emissive_scene; ++k) emissive_scene = s->materials[k].kind == RT_MAT_DIFFUSE_LIGHT;
    for (uint32_t bounce = 0;; ++bounce) {
        if (depth <= 0) { L_end = V(0, 0, 0); break; }
        hit_record rec;
        memset(&rec, 0, sizeof(rec));
        hit_ctx cx = {s, g, 1 + bounce};
        g->medium_draws = 0;
        ++*rays;
        if (!hit_node(&cx, s->root, &r, 0.001, INFINITY, &rec)) {
            if (s->sky_mode == RT_SKY_GRADIENT) {
                v3 unit_direction = vunit(r.dir);
                double t = 0.5 * (unit_direction.e[1] + 1.0);
                L_end = vadd(vscale(1.0 - t, V(1.0, 1.0, 1.0)), vscale(t, V(0.5, 0.7, 1.0)));
            } else {
                L_end = V(s->background[0], s->background[1], s->background[2]);
            }
            break;
        }
        ray_t scattered;
        v3 attenuation;
        v3 e = emissive_scene ? emitted(s, &rec) : V(0, 0, 0);
        if (!scatter(s, g, 1 + bounce, &r, &rec, &attenuation, &scattered)) {
            /* book 1: absorbed -> black; book 2: -> emitted */
            L_end = e;
            break;
        }
        att_stack[n] = attenuation;
        emi_stack[n] = e;
        ++n;
        if (assoc == ORC_ASSOC_FORWARD) {
            if (s->materials[rec.mat].kind == RT_MAT_DIFFUSE_LIGHT) L = vadd(L, vmul(T, e));
            T = vmul(T, attenuation);
        }
        r = scattered;
        --depth;
    }
    if (assoc == ORC_ASSOC_FORWARD) {
        /* L_end is the sky/background, an emitter that ended the path, or black */
        return vadd(L, vmul(T, L_end));
    }
    v3 acc = L_end;
    for (int k = n - 1; k >= 0; --k) {
        acc = vmul(att_stack[k], acc);
        if (emissive_scene) acc = vadd(emi_stack[k], acc);
    }
    return acc;
}

/* ------------------------------------------------------------------ "The Rest Of Your Life"
 * There is no book-3 code in the reference repository (its third directory is a copy of book 2), so this
 * part restates the final listing of the PUBLIC BOOK TEXT (Ray Tracing: The Rest of Your Life, v3.2) and
 * its parity is UNPINNED: no output of the reference exists to check it against.
 *
 * Philox words of a lambertian bounce at segment site s: block 0 = [mixture choice, light index,
 * first and second number of the chosen sampler]; metal and dielectric draw as in books 1/2.   */
typedef struct { v3 u, v, w; } onb_t;
static onb_t onb_from_w(v3 n) { /* onb::build_from_w */
    onb_t o;
    o.w = vunit(n);
    v3 a = (fabs(o.w.e[0]) > 0.9) ? V(0, 1, 0) : V(1, 0, 0);
    o.v = vunit(vcross(o.w, a));
    o.u = vcross(o.w, o.v);
    return o;
}
static v3 onb_local(const onb_t* o, v3 a) {
    return vadd(vadd(vscale(a.e[0], o->u), vscale(a.e[1], o->v)), vscale(a.e[2], o->w));
}
/* hittable::pdf_value of xz_rect and sphere */
static double shape_pdf_value(const rt_scene_desc* s, const rt_node* n, v3 origin, v3 v) {
    hit_record rec;
    memset(&rec, 0, sizeof(rec));
    ray_t r = {origin, v, 0.0};
    if (n->kind == RT_NODE_XZ_RECT) {
        if (!hit_rect(n, 1, 0, 2, &r, 0.001, INFINITY, &rec)) return 0;
        double area = (n->p[1] - n->p[0]) * (n->p[3] - n->p[2]);
        double distance_squared = rec.t * rec.t * vlen2(v);
        double cosine = fabs(vdot(v, rec.normal) / vlen(v));
        return distance_squared / (cosine * area);
    }
    if (n->kind == RT_NODE_SPHERE) {
        v3 center = V(n->p[0], n->p[1], n->p[2]);
        if (!hit_sphere_at(center, n->p[3], n->material, s->n_textures > 0, &r, 0.001, INFINITY, &rec)) return 0;
        double cos_theta_max = sqrt(1 - n->p[3] * n->p[3] / vlen2(vsub(center, origin)));
        double solid_angle = 2 * ORC_PI * (1 - cos_theta_max);
        return 1 / solid_angle;
    }
    return 0;
}
/* hittable::random of xz_rect and sphere; r1 = word 2, r2 = word 3 of the current block */
static v3 shape_random(const rt_node* n, v3 origin, orc_rng* g) {
    if (n->kind == RT_NODE_XZ_RECT) { /* point3(random_double(x0,x1), k, random_double(z0,z1)): z is drawn first */
        double z = rng_draw_range(g, 3, n->p[2], n->p[3]);
        double x = rng_draw_range(g, 2, n->p[0], n->p[1]);
        return vsub(V(x, n->p[4], z), origin);
    }
    v3 center = V(n->p[0], n->p[1], n->p[2]);
    v3 direction = vsub(center, origin);
    double distance_squared = vlen2(direction);
    onb_t uvw = onb_from_w(direction);
    double r1 = rng_draw(g, 2), r2 = rng_draw(g, 3); /* random_to_sphere */
    double z = 1 + r2 * (sqrt(1 - n->p[3] * n->p[3] / distance_squared) - 1);
    double phi = 2 * ORC_PI * r1;
    double x = cos(phi) * sqrt(1 - z * z);
    double y = sin(phi) * sqrt(1 - z * z);
    return onb_local(&uvw, V(x, y, z));
}
/* hittable_list::pdf_value over the light shapes */
static double lights_pdf_value(const rt_scene_desc* s, v3 origin, v3 v) {
    const rt_node* L = &s->nodes[s->lights_plus1 - 1];
    double weight = 1.0 / L->child1;
    double sum = 0.0;
    for (int i = 0; i < L->child1; ++i)
        sum += weight * shape_pdf_value(s, &s->nodes[s->children[L->child0 + i]], origin, v);
    return sum;
}

/* ray_color of the book's final listing.  ORC_ASSOC_REFERENCE replays the recursion
 *     emitted + attenuation * scattering_pdf * ray_color(...) / pdf_val      (diffuse)
 *     attenuation * ray_color(...)                                            (specular)
 * from a stack; ORC_ASSOC_FORWARD keeps  L += T*e,  T *= (1/pdf_val) * (scattering_pdf * attenuation). */
static v3 ray_color_pdf(const rt_scene_desc* s, orc_rng* g, ray_t r, int depth, int assoc, uint64_t* rays) {
    static _Thread_local v3 att_stack[ORC_MAX_DEPTH], emi_stack[ORC_MAX_DEPTH];
    static _Thread_local double spdf_stack[ORC_MAX_DEPTH], pdf_stack[ORC_MAX_DEPTH];
    static _Thread_local int spec_stack[ORC_MAX_DEPTH];
    int n = 0;
    v3 L_end = V(0, 0, 0), T = V(1, 1, 1), L = V(0, 0, 0);
    for (uint32_t bounce = 0;; ++bounce) {
        if (depth <= 0) break;
        hit_record rec;
        memset(&rec, 0, sizeof(rec));
        const uint32_t site = 1 + bounce;
        hit_ctx cx = {s, g, site};
        ++*rays;
        if (!hit_node(&cx, s->root, &r, 0.001, INFINITY, &rec)) {
            L_end = V(s->background[0], s->background[1], s->background[2]);
            break;
        }
        const rt_material* m = &s->materials[rec.mat];
        /* diffuse_light::emitted: the front face only */
        v3 e = V(0, 0, 0);
        if (m->kind == RT_MAT_DIFFUSE_LIGHT && rec.front_face) e = texture_value(s, m->texture, rec.u, rec.v, rec.p);
        if (m->kind == RT_MAT_METAL || m->kind == RT_MAT_DIELECTRIC) { /* srec.is_specular */
            v3 attenuation;
            ray_t scattered;
            scatter(s, g, site, &r, &rec, &attenuation, &scattered); /* metal: the ray is followed whatever its side */
            att_stack[n] = attenuation; emi_stack[n] = e; spec_stack[n] = 1; spdf_stack[n] = pdf_stack[n] = 1; ++n;
            T = vmul(T, attenuation);
            r = scattered;
            --depth;
            continue;
        }
        if (m->kind != RT_MAT_LAMBERTIAN) { /* lights (and anything that does not scatter) end the path */
            L_end = e;
            break;
        }
        v3 attenuation = texture_value(s, m->texture, rec.u, rec.v, rec.p);
        onb_t uvw = onb_from_w(rec.normal); /* cosine_pdf(rec.normal) */
        rng_call(g, site, 0);
        v3 direction;
        if (rng_draw(g, 0) < 0.5) { /* mixture_pdf::generate: p[0] = hittable_pdf(lights, rec.p) */
            const rt_node* Ls = &s->nodes[s->lights_plus1 - 1];
            int idx = (int)rng_draw_range(g, 1, 0, (Ls->child1 - 1) + 1); /* random_int(0, size-1) */
            direction = shape_random(&s->nodes[s->children[Ls->child0 + idx]], rec.p, g);
        } else { /* random_cosine_direction */
            double r1 = rng_draw(g, 2), r2 = rng_draw(g, 3);
            double z = sqrt(1 - r2);
            double phi = 2 * ORC_PI * r1;
            double x = cos(phi) * sqrt(r2);
            double y = sin(phi) * sqrt(r2);
            direction = onb_local(&uvw, V(x, y, z));
        }
        ray_t scattered = {rec.p, direction, r.tm};
        double cosine_w = vdot(vunit(direction), uvw.w);
        double cos_pdf = (cosine_w <= 0) ? 0 : cosine_w / ORC_PI;
        double pdf_val = 0.5 * lights_pdf_value(s, rec.p, direction) + 0.5 * cos_pdf;
        double cosine_n = vdot(rec.normal, vunit(direction)); /* lambertian::scattering_pdf */
        double spdf = cosine_n < 0 ? 0 : cosine_n / ORC_PI;
        att_stack[n] = attenuation; emi_stack[n] = e; spec_stack[n] = 0; spdf_stack[n] = spdf; pdf_stack[n] = pdf_val; ++n;
        T = vmul(T, vscale(1 / pdf_val, vscale(spdf, attenuation)));
        r = scattered;
        --depth;
    }
    if (assoc == ORC_ASSOC_FORWARD) return vadd(L, vmul(T, L_end));
    v3 acc = L_end;
    for (int k = n - 1; k >= 0; --k) {
        if (spec_stack[k]) acc = vmul(att_stack[k], acc);
        else acc = vadd(emi_stack[k], vdiv(vmul(vscale(spdf_stack[k], att_stack[k]), acc), pdf_stack[k]));
    }
    return acc;
}

/* camera::get_ray, B1/camera.h:34-39 and B2/camera.h:38-45.
 * random_in_unit_disk (B1/vec3.h:145-151) builds vec3(random_double(-1,1), random_double(-1,1), 0):
 * g++ evaluates the arguments last to first, so the first draw lands in y.
 * Philox: block 0 of site 0 = [jitter u, jitter v, disk x, disk y]; disk attempt k >= 1 uses
 * block (k+1)/2: words 0,1 when k is odd, words 2,3 when k is even.                       */
static ray_t get_ray(const rt_camera_desc* c, orc_rng* g, double s, double t, int has_time) {
    double px, py;
    for (uint32_t k = 0;; ++k) {
        if ((k & 1u) == 1u) rng_call(g, 0, (k + 1u) / 2u);
        int sx = (k & 1u) ? 0 : 2, sy = sx + 1;
        py = rng_draw_range(g, sy, -1, 1);
        px = rng_draw_range(g, sx, -1, 1);
        if (px * px + py * py + 0.0 * 0.0 >= 1) continue;
        break;
    }
    v3 rd = vscale(c->lens_radius, V(px, py, 0));
    v3 cu = V(c->u[0], c->u[1], c->u[2]), cv = V(c->v[0], c->v[1], c->v[2]);
    v3 offset = vadd(vscale(rd.e[0], cu), vscale(rd.e[1], cv));
    v3 origin = V(c->origin[0], c->origin[1], c->origin[2]);
    v3 llc = V(c->lower_left_corner[0], c->lower_left_corner[1], c->lower_left_corner[2]);
    v3 hor = V(c->horizontal[0], c->horizontal[1], c->horizontal[2]);
    v3 ver = V(c->vertical[0], c->vertical[1], c->vertical[2]);
    ray_t r;
    r.orig = vadd(origin, offset);
    r.dir = vsub(vsub(vadd(vadd(llc, vscale(s, hor)), vscale(t, ver)), origin), offset);
    r.tm = 0.0;
    if (has_time) {
        rng_call(g, 0, ORC_TRY_TIME);
        r.tm = rng_draw_range(g, 0, c->time0, c->time1);
    }
    return r;
}

/* camera::camera, B1/camera.h:17-31 / B2/camera.h:19-35 */
void orc_camera_init(rt_camera_desc* out, const double lookfrom[3], const double lookat[3], const double vup[3],
                     double vfov, double aspect_ratio, double aperture, double focus_dist, double time0,
                     double time1) {
    const double pi = acos(-1.0);
    double theta = vfov * pi / 180.0;
    double h = tan(theta / 2);
    double viewport_height = 2.0 * h;
    double viewport_width = aspect_ratio * viewport_height;
    v3 from = V(lookfrom[0], lookfrom[1], lookfrom[2]), at = V(lookat[0], lookat[1], lookat[2]);
    v3 w = vunit(vsub(from, at));
    v3 u = vunit(vcross(V(vup[0], vup[1], vup[2]), w));
    v3 v = vcross(w, u);
    v3 horizontal = vscale(focus_dist * viewport_width, u);
    v3 vertical = vscale(focus_dist * viewport_height, v);
    v3 llc = vsub(vsub(vsub(from, vdiv(horizontal, 2)), vdiv(vertical, 2)), vscale(focus_dist, w));
    for (int c = 0; c < 3; ++c) {
        out->origin[c] = from.e[c];
        out->lower_left_corner[c] = llc.e[c];
        out->horizontal[c] = horizontal.e[c];
        out->vertical[c] = vertical.e[c];
        out->u[c] = u.e[c];
        out->v[c] = v.e[c];
        out->w[c] = w.e[c];
    }
    out->lens_radius = aperture / 2;
    out->time0 = time0;
    out->time1 = time1;
}

/* ------------------------------------------------------------------ the render loop */
typedef struct {
    const rt_scene_desc* s;
    const rt_camera_desc* cam;
    const rt_render_params* prm;
    int rng_mode, assoc, has_time;
    int row_begin, row_end; /* image rows, 0 = top */
    double* accum;
    int64_t* fixed;
    uint64_t rays, nan_samples, draws;
} render_job;

static int64_t to_fixed(double x) { return llrint(x * (double)(1ull << RT_ACCUM_FRAC_BITS)); }

static void render_rows(render_job* jb) {
    const rt_render_params* prm = jb->prm;
    const int W = prm->width, H = prm->height;
    orc_rng g;
    memset(&g, 0, sizeof(g));
    g.mode = jb->rng_mode;
    g.key[0] = (uint32_t)prm->seed;
    g.key[1] = (uint32_t)(prm->seed >> 32);
    for (int row = jb->row_begin; row < jb->row_end; ++row) {
        const int j = H - 1 - row; /* B1/RayTracing.cpp:115: top row first */
        for (int i = 0; i < W; ++i) {
            v3 pixel_color = V(0, 0, 0);
            int64_t fx[3] = {0, 0, 0};
            for (int sidx = prm->sample_begin; sidx < prm->sample_begin + prm->sample_count; ++sidx) {
                g.pixel = (uint32_t)(row * W + i);
                g.sample = (uint32_t)sidx;
                rng_call(&g, 0, 0);
                double u = (i + rng_draw(&g, 0)) / (W - 1);
                double v = (j + rng_draw(&g, 1)) / (H - 1);
                ray_t r = get_ray(jb->cam, &g, u, v, jb->has_time);
                v3 c = (prm->flags & RT_RENDER_IMPORTANCE)
                           ? ray_color_pdf(jb->s, &g, r, prm->max_depth, jb->assoc, &jb->rays)
                           : ray_color(jb->s, &g, r, prm->max_depth, jb->assoc, &jb->rays);
                pixel_color = vadd(pixel_color, c);
                if (jb->fixed) {
                    /* the accumulation format's range (raytracing_amd.h): NaN, Inf and huge samples are dropped */
                    if (fabs(c.e[0]) < RT_ACCUM_SAMPLE_LIMIT && fabs(c.e[1]) < RT_ACCUM_SAMPLE_LIMIT &&
                        fabs(c.e[2]) < RT_ACCUM_SAMPLE_LIMIT) {
                        for (int k = 0; k < 3; ++k) fx[k] += to_fixed(c.e[k]);
                    } else {
                        jb->nan_samples++;
                    }
                }
            }
            size_t o = 3 * ((size_t)row * W + i);
            if (jb->accum) for (int k = 0; k < 3; ++k) jb->accum[o + k] = pixel_color.e[k];
            if (jb->fixed) for (int k = 0; k < 3; ++k) jb->fixed[o + k] = fx[k];
        }
    }
    jb->draws = g.n_draws;
}

static void* render_thread(void* p) {
    render_rows((render_job*)p);
    return NULL;
}

/* has_time: 1 when the scene comes from book 2/3 (its camera always draws a shutter time,
 * B2/camera.h:44); 0 for book 1.
 * accum (double sums in pixel order, as the reference adds them) and fixed (the
 * order-independent fixed-point sums the GPU produces) may each be NULL.
 * n_threads > 1 is allowed only with ORC_RNG_PHILOX (rand() is one global stream).
 * stats_out[0] = rays, [1] = non-finite samples, [2] = draws.                            */
int orc_render(const rt_scene_desc* s, const rt_camera_desc* cam, const rt_render_params* prm, int rng_mode,
               int assoc, int has_time, int n_threads, double* accum, int64_t* fixed, uint64_t* stats_out) {
    if (!s || !cam || !prm || prm->max_depth > ORC_MAX_DEPTH) return RT_ERR_INVALID_ARGUMENT;
    if ((prm->flags & RT_RENDER_IMPORTANCE) && s->lights_plus1 <= 0) return RT_ERR_INVALID_ARGUMENT;
    if (n_threads < 1) n_threads = 1;
    if (rng_mode == ORC_RNG_RAND) n_threads = 1;
    if (n_threads > prm->height) n_threads = prm->height > 0 ? prm->height : 1;
    render_job* jobs = (render_job*)calloc((size_t)n_threads, sizeof(render_job));
    pthread_t* th = (pthread_t*)calloc((size_t)n_threads, sizeof(pthread_t));
    /* interleave small row blocks over threads: image rows differ a lot in cost */
    const int H = prm->height;
    for (int t = 0; t < n_threads; ++t) {
        jobs[t].s = s; jobs[t].cam = cam; jobs[t].prm = prm;
        jobs[t].rng_mode = rng_mode; jobs[t].assoc = assoc; jobs[t].has_time = has_time;
        jobs[t].accum = accum; jobs[t].fixed = fixed;
        jobs[t].row_begin = (int)((int64_t)H * t / n_threads);
        jobs[t].row_end = (int)((int64_t)H * (t + 1) / n_threads);
    }
    if (n_threads == 1) {
        render_rows(&jobs[0]);
    } else {
        for (int t = 0; t < n_threads; ++t) pthread_create(&th[t], NULL, render_thread, &jobs[t]);
        for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
    }
    if (stats_out) {
        stats_out[0] = stats_out[1] = stats_out[2] = 0;
        for (int t = 0; t < n_threads; ++t) {
            stats_out[0] += jobs[t].rays;
            stats_out[1] += jobs[t].nan_samples;
            stats_out[2] += jobs[t].draws;
        }
    }
    free(jobs);
    free(th);
    return RT_OK;
}

/* Philox / forward-association fixed-point sums of image rows [row_begin, row_end) of the frame
 * described by prm; fixed has (row_end-row_begin)*width*3 entries.  Rows are independent because
 * every sample's random block is keyed by its pixel index.                                  */
int orc_render_rows(const rt_scene_desc* s, const rt_camera_desc* cam, const rt_render_params* prm, int row_begin,
                    int row_end, int has_time, int64_t* fixed) {
    if (!s || !cam || !prm || !fixed || row_begin < 0 || row_end > prm->height || row_begin > row_end)
        return RT_ERR_INVALID_ARGUMENT;
    render_job jb;
    memset(&jb, 0, sizeof(jb));
    jb.s = s; jb.cam = cam; jb.prm = prm;
    jb.rng_mode = ORC_RNG_PHILOX; jb.assoc = ORC_ASSOC_FORWARD; jb.has_time = has_time;
    jb.row_begin = row_begin; jb.row_end = row_end;
    jb.accum = NULL;
    /* render_rows indexes the output by absolute row: bias the pointer */
    jb.fixed = fixed - (size_t)3 * (size_t)row_begin * (size_t)prm->width;
    render_rows(&jb);
    return RT_OK;
}

/* Pixel-centre closest-hit ids (definition: rt_first_hit in raytracing_amd.h). */
int orc_first_hit(const rt_scene_desc* s, const rt_camera_desc* c, int W, int H, int32_t* ids) {
    if (!s || !c || !ids || s->nodes[s->root].kind != RT_NODE_LIST) return RT_ERR_INVALID_ARGUMENT;
    orc_rng g;
    memset(&g, 0, sizeof(g));
    g.mode = ORC_RNG_PHILOX;
    const rt_node* root = &s->nodes[s->root];
    v3 cu = V(c->u[0], c->u[1], c->u[2]), cv = V(c->v[0], c->v[1], c->v[2]);
    v3 origin = V(c->origin[0], c->origin[1], c->origin[2]);
    v3 llc = V(c->lower_left_corner[0], c->lower_left_corner[1], c->lower_left_corner[2]);
    v3 hor = V(c->horizontal[0], c->horizontal[1], c->horizontal[2]);
    v3 ver = V(c->vertical[0], c->vertical[1], c->vertical[2]);
    for (int row = 0; row < H; ++row) {
        int j = H - 1 - row;
        for (int i = 0; i < W; ++i) {
            double sx = (i + 0.5) / (W - 1);
            double ty = (j + 0.5) / (H - 1);
            v3 offset = vadd(vscale(0.0, cu), vscale(0.0, cv));
            ray_t r;
            r.orig = vadd(origin, offset);
            r.dir = vsub(vsub(vadd(vadd(llc, vscale(sx, hor)), vscale(ty, ver)), origin), offset);
            r.tm = c->time0;
            hit_record tmp;
            memset(&tmp, 0, sizeof(tmp));
            double closest = INFINITY;
            int32_t id = -1;
            g.pixel = (uint32_t)(row * W + i);
            g.medium_draws = 0;
            hit_ctx cx = {s, &g, 1};
            for (int o = 0; o < root->child1; ++o) {
                if (hit_node(&cx, s->children[root->child0 + o], &r, 0.001, closest, &tmp)) {
                    closest = tmp.t;
                    id = o;
                }
            }
            ids[(size_t)row * W + i] = id;
        }
    }
    return RT_OK;
}

/* write_color, B1/color.h:8-23, for a whole image */
void orc_write_color(const double* accum, int W, int H, int spp, uint8_t* rgb) {
    double scale = 1.0 / spp;
    for (size_t k = 0; k < (size_t)3 * W * H; ++k) {
        double c = sqrt(scale * accum[k]);
        rgb[k] = (uint8_t)(int)(256 * clampd(c, 0.0, 0.999));
    }
}

void orc_srand(unsigned seed) { srand(seed); }
int orc_rand(void) { return rand(); }
