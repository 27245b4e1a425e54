/* TEST INFRASTRUCTURE -- not part of the product.
 *
 * Runs the oracle (rt_oracle.c, compiled into this program with -fsanitize=address,undefined) over the scene files
 * given on the command line: a small render in each random-number mode, the pixel-centre first hits, and -- for worlds
 * with light shapes -- a render with importance sampling.  Prints one line per scene with a checksum of the fixed-point
 * sums, which tests/test_sanitizers.py compares with the unsanitized library's.  Any sanitizer report makes the exit
 * status non-zero (-fno-sanitize-recover).
 *
 *   make -C oracle sanitize && oracle/_san/sanitize_check tests/golden/scene_c1.rtsc ...
 */
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "raytracing_amd.h"
#include "rt_scene_io.h"

/* entry points of rt_oracle.c */
void orc_camera_init(rt_camera_desc* out, const double lookfrom[3], const double lookat[3], const double vup[3], double vfov,
                     double aspect, double aperture, double focus, double t0, double t1);
void orc_srand(unsigned seed);
int orc_render(const rt_scene_desc* sc, const rt_camera_desc* cam, const rt_render_params* prm, int rng_mode, int assoc,
               int has_time, int threads, double* accum, int64_t* fixed, uint64_t* stats);
int orc_first_hit(const rt_scene_desc* sc, const rt_camera_desc* cam, int W, int H, int32_t* ids);

static uint64_t fnv(const void* p, size_t n, uint64_t h) {
    const unsigned char* b = (const unsigned char*)p;
    for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 0x100000001b3ull;
    return h;
}

int main(int argc, char** argv) {
    enum { W = 24, H = 16, SPP = 3 };
    for (int a = 1; a < argc; ++a) {
        rt_scene_file f;
        if (rt_scene_file_read(argv[a], &f) != RT_OK) {
            fprintf(stderr, "cannot read %s\n", argv[a]);
            return 2;
        }
        const rt_scene_desc* sc = &f.desc;
        const double from[3] = {13, 2.5, 4}, at[3] = {0, 0.5, 0}, up[3] = {0, 1, 0};
        const double from_box[3] = {278, 278, -800}, at_box[3] = {278, 278, 0};
        const int big = strstr(argv[a], "cornell") != NULL || strstr(argv[a], "final") != NULL || strstr(argv[a], "b3") != NULL;
        rt_camera_desc cam;
        orc_camera_init(&cam, big ? from_box : from, big ? at_box : at, up, big ? 40.0 : 25.0, (double)W / H, 0.05, 10.0, 0.0, 1.0);
        rt_render_params prm;
        memset(&prm, 0, sizeof(prm));
        prm.width = W;
        prm.height = H;
        prm.sample_begin = 2;
        prm.sample_count = SPP;
        prm.max_depth = 50;
        prm.seed = 7;
        double* accum = (double*)malloc(sizeof(double) * 3 * W * H);
        int64_t* fixed = (int64_t*)malloc(sizeof(int64_t) * 3 * W * H);
        int32_t* ids = (int32_t*)malloc(sizeof(int32_t) * W * H);
        uint64_t stats[4], h = 0xcbf29ce484222325ull;
        for (int rng = 0; rng < 2; ++rng) {  /* 0: rand() in the reference's order, 1: Philox blocks */
            orc_srand(1u);
            if (orc_render(sc, &cam, &prm, rng, 0, 1, 2, accum, fixed, stats) != 0) return 3;
            h = fnv(fixed, sizeof(int64_t) * 3 * W * H, h);
            h = fnv(stats, sizeof(uint64_t) * 2, h);
        }
        if (sc->lights_plus1 != 0) {
            prm.flags = RT_RENDER_IMPORTANCE;
            if (orc_render(sc, &cam, &prm, 1, 0, 1, 2, accum, fixed, stats) != 0) return 3;
            h = fnv(fixed, sizeof(int64_t) * 3 * W * H, h);
        }
        if (orc_first_hit(sc, &cam, W, H, ids) != 0) return 4;
        h = fnv(ids, sizeof(int32_t) * W * H, h);
        printf("%s %016" PRIx64 "\n", argv[a], h);
        free(accum);
        free(fixed);
        free(ids);
        rt_scene_file_free(&f);
    }
    return 0;
}
