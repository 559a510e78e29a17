/*
 * dvs_scene.h — deterministic synthetic scenes for the rasterizer hot path (host-side C-ABI).
 *
 * The reference's data pipeline (COLMAP / image loading inside the closed `gstrain` plugin,
 * load_train_data at application/diverseshot-cli/source/gs_train.cpp:108-122) is out of scope
 * (SURVEY.md §8(b), §8(f)); BASELINE.json's configs are all "random splats / synthetic cams".
 * This generator is the one source of those scenes for bench.py, the tests and libgstrain's
 * `synthetic:` loader, following SURVEY.md §8(d) "Synthetic inputs".
 */
#ifndef DVS_SCENE_H
#define DVS_SCENE_H
#include <stdint.h>
#include "dvs_raster.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct dvs_scene_spec {
    int32_t n;             /* splats */
    int32_t width, height;
    int32_t sh_degree;     /* higher bands beyond this degree are generated as zero */
    int32_t n_cams;        /* camera 0 at the origin looking +Z; 1..n_cams-1 on a 0.5-radius ring, all look at (0,0,7) */
    uint64_t seed;         /* 1 for the BASELINE configs */
    float fov_x_deg;       /* 60 */
    float scale_log_offset;/* added to the log-scale mean (C5 uses -ln 2) */
} dvs_scene_spec;

/* Fill HOST arrays in the A0 layout (pos[n*3] sh0[n*3] shN[n*45] opacity[n] scale[n*3] rot[n*4]). */
int dvs_synth_splats(const dvs_scene_spec* spec, float* pos, float* sh0, float* shN, float* opacity, float* scale, float* rot);
/* Camera `index` of the spec (A1 block, background black). */
int dvs_synth_camera(const dvs_scene_spec* spec, int index, dvs_camera* out);
/* Target image for the L2 upstream gradient: U[0,1), [3,H,W] planar, seeded by (seed+1, index). */
int dvs_synth_target(const dvs_scene_spec* spec, int index, float* target);
/* Build a camera from pose (world->camera rotation R row-major [9], translation t[3]) and pinhole fov. */
int dvs_make_camera(const float* R, const float* t, float fov_x_deg, int width, int height, dvs_camera* out);

#ifdef __cplusplus
}
#endif
#endif
