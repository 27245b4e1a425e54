/*
 * rt_scene_io.h -- the on-disk form of an rt_scene_desc ("RTSCENE1", little-endian).
 *
 * Scenes of the reference are rolled with the C library's rand() inside function-argument
 * lists ("Ray Tracing In a Weekend/code/RayTracing.cpp":42,49,54), so the same source yields
 * different scenes under different compilers.  A scene therefore travels as DATA: it is
 * dumped once from the reference program and loaded from this file everywhere else.
 *
 * Layout (all integers int32, all reals IEEE double, no padding beyond the structs' own):
 *   char   magic[8]  = "RTSCENE1"
 *   int32  root, n_nodes, n_children, n_materials, n_textures, n_perlins, n_images, sky_mode
 *          (the last word also carries lights_plus1 in bits 8..31; files without lights have 0 there)
 *   double background[3]
 *   rt_node     nodes[n_nodes]
 *   int32       children[n_children]
 *   rt_material materials[n_materials]
 *   rt_texture  textures[n_textures]
 *   rt_perlin   perlins[n_perlins]
 *   n_images x { int32 width, height; uint8 rgb[3*width*height] }
 *
 * Header-only, C and C++; depends on <stdio.h>/<stdlib.h>/<string.h> only.
 */
#ifndef RT_SCENE_IO_H
#define RT_SCENE_IO_H

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "raytracing_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* An rt_scene_desc whose arrays are owned (malloc'ed) by this object. */
typedef struct rt_scene_file {
    rt_scene_desc desc;
    rt_node* nodes;
    int32_t* children;
    rt_material* materials;
    rt_texture* textures;
    rt_perlin* perlins;
    rt_image* images;
    uint8_t** image_data;
} rt_scene_file;

static inline void rt_scene_file_free(rt_scene_file* f) {
    if (!f) return;
    free(f->nodes);
    free(f->children);
    free(f->materials);
    free(f->textures);
    free(f->perlins);
    if (f->image_data) {
        for (int i = 0; i < f->desc.n_images; ++i) free(f->image_data[i]);
    }
    free(f->image_data);
    free(f->images);
    memset(f, 0, sizeof(*f));
}

/* fwrite of n elements; an empty array may be a null pointer, which fwrite must not be handed */
static inline int rt_scene_file_write_array_(FILE* fp, const void* a, size_t elem, int32_t n) {
    if (n < 0 || (n > 0 && !a)) return 0;
    return n == 0 || fwrite(a, elem, (size_t)n, fp) == (size_t)n;
}

static inline int rt_scene_file_write(const char* path, const rt_scene_desc* s) {
    FILE* fp = fopen(path, "wb");
    if (!fp) return RT_ERR_INVALID_ARGUMENT;
    int32_t head[8] = {s->root,       s->n_nodes,   s->n_children, s->n_materials,
                       s->n_textures, s->n_perlins, s->n_images,   (int32_t)((s->sky_mode & 0xFF) | (s->lights_plus1 << 8))};
    int ok = fwrite("RTSCENE1", 1, 8, fp) == 8;
    ok = ok && fwrite(head, sizeof(int32_t), 8, fp) == 8;
    ok = ok && fwrite(s->background, sizeof(double), 3, fp) == 3;
    ok = ok && rt_scene_file_write_array_(fp, s->nodes, sizeof(rt_node), s->n_nodes);
    ok = ok && rt_scene_file_write_array_(fp, s->children, sizeof(int32_t), s->n_children);
    ok = ok && rt_scene_file_write_array_(fp, s->materials, sizeof(rt_material), s->n_materials);
    ok = ok && rt_scene_file_write_array_(fp, s->textures, sizeof(rt_texture), s->n_textures);
    ok = ok && rt_scene_file_write_array_(fp, s->perlins, sizeof(rt_perlin), s->n_perlins);
    for (int i = 0; ok && i < s->n_images; ++i) {
        int32_t wh[2] = {s->images[i].width, s->images[i].height};
        size_t n = (size_t)3 * (size_t)wh[0] * (size_t)wh[1];
        ok = fwrite(wh, sizeof(int32_t), 2, fp) == 2;
        ok = ok && (n == 0 || fwrite(s->images[i].rgb, 1, n, fp) == n);
    }
    ok = (fclose(fp) == 0) && ok;
    return ok ? RT_OK : RT_ERR_INVALID_ARGUMENT;
}

static inline int rt_scene_file_read_array_(FILE* fp, void** out, size_t elem, int32_t n) {
    *out = NULL;
    if (n < 0) return 0;
    if (n == 0) return 1;
    *out = malloc(elem * (size_t)n);
    if (!*out) return 0;
    return fread(*out, elem, (size_t)n, fp) == (size_t)n;
}

static inline int rt_scene_file_read(const char* path, rt_scene_file* f) {
    memset(f, 0, sizeof(*f));
    FILE* fp = fopen(path, "rb");
    if (!fp) return RT_ERR_INVALID_ARGUMENT;
    char magic[8];
    int32_t head[8];
    int ok = fread(magic, 1, 8, fp) == 8 && memcmp(magic, "RTSCENE1", 8) == 0;
    ok = ok && fread(head, sizeof(int32_t), 8, fp) == 8;
    if (ok) {
        f->desc.abi_version = RT_ABI_VERSION;
        f->desc.root = head[0];
        f->desc.n_nodes = head[1];
        f->desc.n_children = head[2];
        f->desc.n_materials = head[3];
        f->desc.n_textures = head[4];
        f->desc.n_perlins = head[5];
        f->desc.n_images = head[6];
        f->desc.sky_mode = head[7] & 0xFF;
        f->desc.lights_plus1 = (int32_t)((uint32_t)head[7] >> 8);
    }
    ok = ok && fread(f->desc.background, sizeof(double), 3, fp) == 3;
    ok = ok && rt_scene_file_read_array_(fp, (void**)&f->nodes, sizeof(rt_node), f->desc.n_nodes);
    ok = ok && rt_scene_file_read_array_(fp, (void**)&f->children, sizeof(int32_t), f->desc.n_children);
    ok = ok && rt_scene_file_read_array_(fp, (void**)&f->materials, sizeof(rt_material), f->desc.n_materials);
    ok = ok && rt_scene_file_read_array_(fp, (void**)&f->textures, sizeof(rt_texture), f->desc.n_textures);
    ok = ok && rt_scene_file_read_array_(fp, (void**)&f->perlins, sizeof(rt_perlin), f->desc.n_perlins);
    if (ok && f->desc.n_images > 0) {
        f->images = (rt_image*)calloc((size_t)f->desc.n_images, sizeof(rt_image));
        f->image_data = (uint8_t**)calloc((size_t)f->desc.n_images, sizeof(uint8_t*));
        ok = f->images && f->image_data;
        for (int i = 0; ok && i < f->desc.n_images; ++i) {
            int32_t wh[2];
            ok = fread(wh, sizeof(int32_t), 2, fp) == 2 && wh[0] >= 0 && wh[1] >= 0;
            if (!ok) break;
            size_t n = (size_t)3 * (size_t)wh[0] * (size_t)wh[1];
            if (n) {
                f->image_data[i] = (uint8_t*)malloc(n);
                ok = f->image_data[i] && fread(f->image_data[i], 1, n, fp) == n;
            }
            f->images[i].width = wh[0];
            f->images[i].height = wh[1];
            f->images[i].rgb = f->image_data[i];
        }
    }
    fclose(fp);
    if (!ok) {
        rt_scene_file_free(f);
        return RT_ERR_INVALID_ARGUMENT;
    }
    f->desc.nodes = f->nodes;
    f->desc.children = f->children;
    f->desc.materials = f->materials;
    f->desc.textures = f->textures;
    f->desc.perlins = f->perlins;
    f->desc.images = f->images;
    return RT_OK;
}

#ifdef __cplusplus
}
#endif
#endif /* RT_SCENE_IO_H */
