/*
 * rt_oracle.h -- CPU restatement of the reference renderer's per-pixel hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the checker the GPU path is compared
 * against; it is never linked into, loaded by or shipped with the product
 * library.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may use it.
 *
 * Pinned: every function here is checked, bit for bit, against outputs of the
 * reference itself (oracle/_ref, built from /root/reference/src by
 * oracle/Makefile) committed under tests/golden/ by oracle/make_golden.py.
 *
 * Scene, camera and render parameters use the POD structs of the public ABI
 * header (include/rt_mi355x.h), which are the flattened form of the
 * reference's Scene/Camera objects.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include "rt_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

/* PCG32 stream (reference: class RNG, src/rng.h:29-99) */
typedef struct orc_rng { uint64_t state, inc; } orc_rng;
void orc_rng_set_sequence(orc_rng *r, uint64_t seq, uint64_t seed); /* rng.h:39-46 */
void orc_rng_set_sequence1(orc_rng *r, uint64_t seq);               /* rng.h:38, seed = MixBits(seq) */
uint32_t orc_rng_u32(orc_rng *r);                                   /* rng.h:49-57 */
float orc_rng_float(orc_rng *r);                                    /* rng.h:60-65 */
void orc_rng_advance(orc_rng *r, int64_t delta);                    /* rng.h:68-82 */
float orc_halton(int index, int base);                              /* rng.h:103-112 */

/* camera constants main() derives before rendering (src/main.cpp:404-414):
 * out[0..8] = cameraToWorld (column-major 3x3), out[9] = half width,
 * out[10] = half height, out[11] = pixel size */
void orc_camera_derive(const rt_camera *cam, float out[12]);

/* Renderer::TraceRay / TraceShadowRay (src/main.cpp:16-125) */
typedef struct orc_stats { uint64_t closest, shadow, paths, box_tests, tri_tests, node_visits, gathers, photon_visits; } orc_stats;
int orc_trace(const rt_scene_desc *sc, const float ray6[6], rt_hit *hit, orc_stats *st);
int orc_trace_shadow(const rt_scene_desc *sc, const float ray6[6], float t_max, orc_stats *st);

/* samplers (src/lights.h:208-288, src/materials.h:87-303).
 * surf = P xyz, N xyz (normalised), V xyz (normalised view vector), front flag */
typedef struct orc_surface { float p[3], n[3], v[3]; int front; } orc_surface;
typedef struct orc_sample { float dir[3], mult[3], prob, dist; int lobe; } orc_sample;
int orc_light_generate(const rt_light *l, const orc_surface *s, orc_rng *rng, orc_sample *out);
float orc_light_pdf(const rt_light *l, const float p[3], const float dir[3]);
int orc_mtl_generate(const rt_material *m, const orc_surface *s, orc_rng *rng, orc_sample *out);
float orc_mtl_pdf(const rt_material *m, const orc_surface *s, const float dir[3]);

/* tracePath (src/main.cpp:160-268) */
void orc_trace_path(const rt_scene_desc *sc, const float ray6[6], orc_rng *rng,
                    const rt_render_params *prm, float radiance[3], orc_stats *st);

/* threadRenderTiles (src/main.cpp:271-385) with per-pixel / per-sample seeding.
 * n_threads <= 0 means one thread. Renders rows [y0,y1) (y1 <= 0: all rows) of
 * the tile shard given in prm; out pointers are full-frame host planes. */
int orc_render(const rt_scene_desc *sc, const rt_camera *cam, const rt_render_params *prm,
               rt_frame_out *out, int n_threads, int y0, int y1, orc_stats *st);

/* Photon maps (src/photonmap.h:38-414) and the emission loops of the project-12 snapshot (its main.cpp:292-470),
 * with one PCG stream per photon path (see rt_photon_params in the ABI header).
 * orc_photons_emit runs paths [path_begin, path_begin+path_count) in order and stops early when an append finds
 * `capacity` photons stored already; *paths_run = paths started. */
int orc_photons_emit(const rt_scene_desc *sc, const rt_photon_params *pp, int which, uint64_t path_begin, uint64_t path_count,
                     rt_photon *out, uint32_t capacity, uint64_t *n_stored, uint64_t *paths_run, orc_stats *st);
/* PointLight::RandomPhoton (src/lights.h:92-124) and the snapshot's MtlBlinn::GenerateSample (p12 materials.h:82-193) */
void orc_light_random_photon(const rt_light *l, orc_rng *rng, float ray6[6], float c3[3]);
int orc_p12_mtl_generate(const rt_material *m, const orc_surface *s, orc_rng *rng, orc_sample *out);
int orc_photon_build(const rt_photon *in, uint32_t n, rt_photon *tree);
uint64_t orc_photon_irradiance(const rt_photon *tree, uint32_t n, const float pos[3], const float normal[3], float radius,
                               float irrad[3]);
/* the maps orc_render's Whitted-style shader reads when prm->photon_flags != 0 (not copied) */
void orc_set_photon_map(int which, const rt_photon *tree, uint32_t n);

/* TextureMap::Eval of texture `tex` (1 + index into sc->textures) at uvw (src/scene.h:346, src/texture.cpp:100-134) */
void orc_texture_eval(const rt_scene_desc *sc, int tex, const float uvw[3], float rgb[3]);
/* TexturedColor::EvalEnvironment of the scene's environment for a direction (src/scene.h:390-397) */
void orc_environment_eval(const rt_scene_desc *sc, const float dir[3], float rgb[3]);

/* sRGB + 8-bit quantisation (cyColor.h:98,324-325) */
void orc_resolve_pixel(const float linear[3], int srgb, uint8_t rgb[3]);

#ifdef __cplusplus
}
#endif
#endif
