/*
 * pt_oracle.h — CPU restatement of the reference's render() hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load liboracle.so; the product
 * (path_tracer_amd/) never does.  See oracle/README.md for the pin status:
 * the RNG is pinned against the reference's own xorshift.hpp compiled from
 * /root/reference (oracle/_ref/); everything that touches sycl::float3 is
 * "parity unpinned" because triSYCL is absent and stand-in headers are not
 * allowed — those functions follow the reference source line by line and are
 * anchored on the KATs SURVEY.md §8a/a13 recorded from the reference headers.
 *
 * Takes the same PtSceneDesc / PtCamera / PtRenderParams tables as the product's
 * C ABI (include/pt_render.h), so one scene dump feeds both sides.
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include "../include/pt_render.h"

#ifdef __cplusplus
extern "C" {
#endif

/* 0 = platform libm (the reference's semantics on this host, default);
 * 1 = ptm_portable.h (bit-comparable with the GPU kernels).                    */
void orc_set_math(int portable);
int orc_get_math(void);

/* xorshift<32>::operator()  xorshift.hpp:64-75 — advances *state, returns it.  */
uint32_t orc_xorshift32(uint32_t* state);
/* LocalPseudoRNG  rtweekend.hpp:33-92 */
float orc_float_t(uint32_t* state);
void orc_unit_vec(uint32_t* state, float out[3]);
void orc_in_unit_ball(uint32_t* state, float out[3]);
void orc_in_unit_disk(uint32_t* state, float out[3]);

/* camera::camera  camera.hpp:67-87 */
void orc_camera_init(PtCamera* cam, const float look_from[3], const float look_at[3],
                     const float vup[3], float vfov_deg, float aspect_ratio, float aperture,
                     float focus_dist, float time0, float time1);

/* Event counters of one orc_render call (the algorithmic work
 * per sample comes from these).                                                */
typedef struct OrcCounters {
  uint64_t samples;
  uint64_t rays;              /* hit_world calls                         render.hpp:60 */
  uint64_t rng_draws;
  uint64_t tests[PT_HIT_KIND_COUNT];    /* top-level hit() calls by kind */
  uint64_t accepts[PT_HIT_KIND_COUNT];  /* calls that returned true      */
  uint64_t rect_tests;        /* incl. the 6 sides of every box and box-boundary tests */
  uint64_t sphere_tests;      /* incl. medium boundary tests */
  uint64_t scatters[5];       /* scatter() calls by material kind */
  uint64_t end_sky, end_emit, end_depth;
  /* Exit points, so that the algorithmic work can be priced per exit as SURVEY.md §8(d) does (appended: the
   * fields above keep their offsets).                                                                          */
  uint64_t rect_exit[3];      /* rectangle.hpp:36 t-reject | :40 bounds-reject | accept  (box sides included)   */
  uint64_t tri_exit[5];       /* triangle.hpp:71 |a|<eps | :81 u | :86 v | :91 t-range | accept                 */
  uint64_t sphere_exit[3];    /* sphere.hpp:74 discriminant <= 0 | both roots outside (min,max) | accept        */
  uint64_t sphere_moving;     /* sphere tests that evaluated center(time) with time0 != time1 (sphere.hpp:52)   */
  uint64_t tex_evals[3];      /* texture value() calls by kind: checker, solid, image (texture.hpp:154 order)   */
} OrcCounters;

/* render<W,H,S>() render.hpp:25-160 on host cores (OpenMP over rows).
 * fb layout as pt_render (full frame or shard tiles).  counters may be NULL.
 * Returns 0, or PT_ERR_* for a malformed scene.                                */
int orc_render(const PtSceneDesc* scene, const PtCamera* cam, const PtRenderParams* p,
               float* fb, OrcCounters* counters);

/* Only pixels y in [y0,y1): used by the bounded CPU-baseline timing.           */
int orc_render_rows(const PtSceneDesc* scene, const PtCamera* cam, const PtRenderParams* p,
                    int32_t y0, int32_t y1, float* fb_rows, OrcCounters* counters);

/* Arbitrary pixels (full spp each) of the frame: xy = [n][2] -> out [n][3].    */
int orc_render_pixels(const PtSceneDesc* scene, const PtCamera* cam, const PtRenderParams* p,
                      const int32_t* xy, int32_t n, float* out);

/* Same, plus rays[n] = rays traced per pixel (its sequential chain length).     */
int orc_render_pixels_rays(const PtSceneDesc* scene, const PtCamera* cam, const PtRenderParams* p,
                           const int32_t* xy, int32_t n, float* out, uint64_t* rays);

int orc_bounce(const PtSceneDesc* scene, const PtBounceIn* in, PtBounceOut* out, int32_t n,
               int32_t depth_unused);

int orc_camera_rays(const PtCamera* cam, int32_t width, int32_t height, const int32_t* xy,
                    const uint32_t* rng_in, PtCameraRay* out, int32_t n);

/* op codes as pt_debug_math; uses the current orc_set_math() mode.             */
int orc_math(int32_t op, const float* a, const float* b, float* out, int64_t n);

/* main.cpp:33-59 output stage. rgb8 is [height][width][3], row 0 = top.        */
void orc_tonemap_rgb8(const float* fb, int32_t width, int32_t height, uint8_t* rgb8);

int orc_max_threads(void);
void orc_set_threads(int n); /* omp_set_num_threads */

#ifdef __cplusplus
}
#endif
#endif
