/*
 * rt_mi355x.h -- C ABI of the MI355X (gfx950) rendering backend.
 *
 * The library replaces the per-pixel hot path of the reference ray tracer:
 * the tile/pixel/sample loop, camera-ray generation, scene traversal,
 * Sphere/Plane/TriObj(+BVH) intersection, the path integrator with next-event
 * estimation and MIS, the Blinn material and spherical-light samplers, and
 * the sRGB/8-bit resolve.  Scene loading, the Node/Object/Material/Light
 * object model and PNG output stay on the host side of this boundary.
 *
 * Conventions
 *   - every entry point is extern "C", takes plain pointers and sizes, returns
 *     an rt_status (0 = OK, negative = error) and never throws or aborts;
 *     rt_last_error() returns the text of the last failure on a context;
 *   - matrices are 3x4, column-major, 12 floats: cell[0..2] = first column,
 *     cell[9..11] = translation (the reference's Matrix34f, cyMatrix.h:1089-1100);
 *   - all arithmetic is IEEE binary32 without FMA contraction, so primary-hit
 *     depth and object ids are bit-identical to a CPU build of the reference
 *     made with -ffp-contract=off;
 *   - one rt_ctx is bound to one GPU and may be used by one host thread at a time.
 *
 * Reference interfaces replaced (paths relative to the reference checkout):
 *   class Renderer                      src/renderer.h:205-238
 *   threadRenderTiles / tracePath       src/main.cpp:160-385
 *   Renderer::TraceRay/TraceShadowRay   src/main.cpp:115-125
 */
#ifndef RT_MI355X_H
#define RT_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 6

typedef struct rt_ctx rt_ctx;

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID = -1,     /* bad argument (null pointer, negative size, index out of range) */
    RT_ERR_NO_DEVICE = -2,   /* no usable GPU / HIP runtime failed to initialise */
    RT_ERR_HIP = -3,         /* a HIP call failed; see rt_last_error */
    RT_ERR_UNSUPPORTED = -4, /* scene uses a feature outside the supported set */
    RT_ERR_IO = -5,          /* file could not be read or parsed */
    RT_ERR_STATE = -6,       /* call order violated (e.g. render before scene upload) */
    RT_ERR_CANCELLED = -7    /* rt_cancel stopped the call; pixels finished in earlier sampling rounds are in the planes, the others are undefined */
} rt_status;

/* ---- scene description (flattened form of the reference's object model) ---- */

typedef enum rt_prim { RT_PRIM_NONE = 0, RT_PRIM_SPHERE = 1, RT_PRIM_PLANE = 2, RT_PRIM_MESH = 3 } rt_prim;
typedef enum rt_light_type { RT_LIGHT_AMBIENT = 0, RT_LIGHT_DIRECT = 1, RT_LIGHT_POINT = 2 } rt_light_type;

/* One scene-graph node (reference: class Node + Transformation, src/scene.h:155-195,414-454).
 * Nodes are stored in depth-first pre-order; node 0 is the root; parent < index. */
typedef struct rt_node {
    float tm[12];     /* node-to-parent transform */
    float itm[12];    /* its inverse */
    int32_t parent;   /* -1 for the root */
    int32_t prim;     /* rt_prim */
    int32_t mesh;     /* index into rt_scene_desc.meshes when prim == RT_PRIM_MESH, else -1 */
    int32_t material; /* index into rt_scene_desc.materials, -1 = none */
} rt_node;

/* BVH node as the reference lays it out (cyBVH.h:187-200): 6 floats of bounds
 * (min xyz, max xyz) + one data word (bit 31 = leaf; leaf: bits 28..30 = count-1,
 * bits 0..27 = first element; inner: index of the first child, second = first+1).
 * Node 0 is unused, node 1 is the root. */
typedef struct rt_bvh_node {
    float bounds[6];
    uint32_t data;
} rt_bvh_node;

/* Triangle mesh (reference: TriObj = cyTriMesh + BVHTriMesh, src/objects.h:50-71).
 * All index arrays hold 3 indices per face. vn/fn and vt/ft are mandatory: the
 * reference dereferences both for every triangle hit (src/TriObj.cpp:126-134). */
typedef struct rt_mesh {
    const float *v;      uint32_t nv;   /* positions,  3 floats each */
    const float *vn;     uint32_t nvn;  /* normals,    3 floats each */
    const float *vt;     uint32_t nvt;  /* tex coords, 3 floats each */
    const uint32_t *f;                  /* position indices  [nf*3] */
    const uint32_t *fn;                 /* normal indices    [nf*3] */
    const uint32_t *ft;                 /* tex-coord indices [nf*3] */
    uint32_t nf;
    /* The reference's BVH (cyBVH::Build, host side).  bvh_nodes == NULL: the library builds a BVH of its own on the GPU
     * at rt_scene_upload (Morton order + one thread per node) - for every mesh of the scene if any mesh comes without
     * one, or when RT_GPU_BVH=1 is set.  Depths and object ids of closest hits do not depend on which BVH is walked
     * (same triangles, same triangle test); which of two triangles hit at exactly the same distance is reported, and
     * the box / triangle test counters, do. */
    const rt_bvh_node *bvh_nodes; uint32_t n_bvh_nodes; /* including the unused node 0 */
    const uint32_t *bvh_elements;       /* face ids referenced by the leaves [nf] */
    uint32_t bvh_depth;                 /* depth of the deepest leaf, root = 0 */
    /* Sub-materials of an OBJ file that names materials (`usemtl`): the reference turns them into one MultiMtl
     * (src/xmlload.cpp:298-322, src/materials.h:350-382) whose members are picked by HitInfo::mtlID.  Here the
     * members are n_sub_materials consecutive entries of rt_scene_desc.materials starting at the node's material,
     * and face_material[face] (< n_sub_materials) is the member the face names (cyTriMesh::GetMaterialIndex). */
    const uint32_t *face_material;      /* [nf], or NULL: the mesh has no sub-materials */
    uint32_t n_sub_materials;           /* 0 when face_material is NULL */
} rt_mesh;

/* Blinn material constants (reference: MtlBasePhongBlinn, src/materials.h:52-60). */
typedef struct rt_material {
    float diffuse[3];
    float specular[3];
    float glossiness;
    float emission[3];
    float reflection[3];
    float refraction[3];
    float absorption[3];
    float ior;
    int32_t diffuse_texture; /* 1 + index into rt_scene_desc.textures, 0 = none. The only material channel either
                                renderer evaluates with its texture (the snapshot's Shade: diffuse.Eval(UVW),
                                project12.zip src/materials.cpp:22; everything else reads GetValue()) */
} rt_material;

/* A TextureMap with its Texture (src/scene.h:294-362, src/texture.h, src/texture.cpp:100-134): uvw is taken into
 * texture space by itm (TransformTo), tiled into [0,1), then either a checkerboard of two colours or an 8-bit RGB
 * image with bilinear filtering is evaluated.  The owning TexturedColor multiplies the result by its value. */
typedef enum rt_texture_type { RT_TEX_CHECKER = 1, RT_TEX_IMAGE = 2 } rt_texture_type;
typedef struct rt_texture {
    int32_t type;          /* rt_texture_type */
    float itm[12];         /* inverse of the texture transformation, 3x4 column-major */
    float color1[3];       /* checkerboard: colour where exactly one of u, v is <= 0.5 is color2, else color1 */
    float color2[3];
    const uint8_t *pixels; /* image: [height*width*3], row 0 first (Color24) */
    int32_t width, height;
} rt_texture;

/* Light (reference: AmbientLight/DirectLight/PointLight, src/lights.h). */
typedef struct rt_light {
    int32_t type; /* rt_light_type */
    float intensity[3];
    float position[3];
    float direction[3];
    float size;
    float attenuation;
} rt_light;

typedef struct rt_scene_desc {
    const rt_node *nodes;         uint32_t n_nodes;
    const rt_mesh *meshes;        uint32_t n_meshes;
    const rt_material *materials; uint32_t n_materials;
    const rt_light *lights;       uint32_t n_lights;
    float background[3];  /* background colour (src/scene.h:502): what a camera ray sees that hits nothing */
    float environment[3]; /* environment colour (src/scene.h:503): what a secondary ray of the Whitted-style shader sees then */
    const rt_texture *textures;  uint32_t n_textures;
    int32_t background_texture;  /* 1 + index into textures, 0 = none: colour * texture. The path tracer evaluates it at
                                    uvw = ray direction (src/main.cpp:175), the Whitted-style shader at the pixel's
                                    (u, v, 1) (project12.zip src/main.cpp:224-226) */
    int32_t environment_texture; /* likewise, evaluated through EvalEnvironment (src/scene.h:390-397) */
    int32_t face_materials;      /* 0: as the reference renders - no intersector ever sets HitInfo::mtlID (it stays 0,
                                    src/scene.h:92, src/TriObj.cpp:12-137), so every face of a mesh with sub-materials
                                    is shaded with the first one.  1: a face is shaded with the sub-material it names
                                    (node material + rt_mesh.face_material[face]) */
} rt_scene_desc;

/* Camera as the reference holds it after Camera::Load (src/scene.h:458-480,
 * src/xmlload.cpp:104-120): dir is normalised target-pos, up is re-orthogonalised. */
typedef struct rt_camera {
    float pos[3], dir[3], up[3];
    float fov, focaldist, dof;
    int32_t width, height;
    int32_t srgb;
} rt_camera;

/* RT_MODE_PATH:    the path tracer of the reference at HEAD (src/main.cpp:160-268).
 * RT_MODE_WHITTED: the Whitted-style shader of the reference's project-12 snapshot (project12.zip,
 *                  src/materials.cpp:14-202, src/lights.h:69-108, src/main.cpp:132-270) without photon
 *                  maps: Blinn-Phong direct light with the adaptive 4..64-ray soft shadows of
 *                  PointLight::Illuminate, then - while bounce < max_bounces (the snapshot compiles in 5) -
 *                  one Monte-Carlo indirect ray on the first two bounces, glossy reflection and refraction. */
typedef enum rt_mode { RT_MODE_PATH = 0, RT_MODE_WHITTED = 1 } rt_mode;

/* The four photon-map switches of the snapshot (project12.zip src/renderer.cpp:4-7), as bits.
 * They select how the photon maps are filled (src/main.cpp:292-470 there) and what the Whitted-style
 * shader takes from them (src/materials.cpp:24-72,75 there). */
enum {
    RT_PM_DIRECT = 1,      /* doingDirectWithPhotonMapping: direct light from the global map, no light loop   */
    RT_PM_INDIRECT = 2,    /* doingIndirectWithPhotonMapping: indirect light from the global map, no Monte-Carlo ray */
    RT_PM_CAUSTICS = 4,    /* doingCaustics: caustics map                                                     */
    RT_PM_MONTE_CARLO = 8  /* monteCarloWithPhoton: global map holds first hits only                          */
};

/* Everything the reference fixes at compile time (src/main.cpp:93,107,164,
 * 220-221,262,273-274,369) plus the seeding convention and the tile shard. */
typedef struct rt_render_params {
    int32_t mode;          /* rt_mode */
    int32_t min_spp;       /* reference: 128 */
    int32_t max_spp;       /* reference: 128 */
    float delta_max;       /* reference: 0.01 */
    int32_t max_bounces;   /* reference: 50 */
    float bias;            /* reference: 0.002 */
    int32_t halton_period; /* reference: 16 (1..64) */
    int32_t tile_size;     /* reference: 16 */
    int32_t tile_rank;     /* render tiles t with t % tile_world == tile_rank */
    int32_t tile_world;    /* 1 = whole image */
    uint64_t seed;         /* frame seed; pixel p uses seed + p*0x9E3779B97F4A7C15,
                              PCG stream 0 for the four per-pixel offsets and
                              stream k+1 for sample k */
    int32_t photon_flags;  /* RT_MODE_WHITTED: RT_PM_* bits; 0 = no photon maps. The maps must have been built
                              (rt_photons_fill or rt_photons_build) with the same bits. */
    float gather_radius;   /* RT_MODE_WHITTED with photon maps: search radius of the 128-photon irradiance
                              estimate; reference: 3.0 */
} rt_render_params;

/* Output planes, row-major, row 0 = top (reference: RenderImage, src/renderer.h:38-106).
 * Any pointer may be null. Pixels outside the tile shard are left untouched.
 * With device_memory != 0 the pointers are device addresses on the context's GPU. */
typedef struct rt_frame_out {
    uint8_t *rgb;          /* [H*W*3]  8-bit colour (Color24)                         */
    float *z;              /* [H*W]    ray parameter t of the first sample's primary hit, FLT_MAX = miss */
    int32_t *sample_count; /* [H*W]    samples taken                                   */
    int32_t *object_id;    /* [H*W]    pre-order node index; -1 = miss; -2-l = light l */
    float *linear;         /* [H*W*3]  mean linear radiance before sRGB/quantisation   */
    int32_t device_memory;
} rt_frame_out;

/* Result of one closest-hit query (reference: HitInfo, src/scene.h:78-93). */
typedef struct rt_hit {
    float z;         /* ray parameter, FLT_MAX on a miss */
    float p[3];      /* world-space hit point */
    float n[3];      /* world-space shading normal (not normalised) */
    float uvw[3];
    int32_t node;    /* pre-order node index; -1 miss; -2-l light l */
    int32_t front;   /* 1 = front face */
} rt_hit;

typedef struct rt_counters {
    uint64_t closest_rays;  /* TraceRay call equivalents       */
    uint64_t shadow_rays;   /* TraceShadowRay call equivalents */
    uint64_t camera_paths;
    uint64_t box_tests;     /* BVH slab tests     */
    uint64_t tri_tests;     /* triangle tests     */
    uint64_t gathers;       /* irradiance estimates (photon-map lookups) */
    uint64_t photon_visits; /* kd-tree nodes visited by them (counting variant only) */
    double kernel_ms;       /* GPU time of the last rt_render, HIP events on the render stream */
    double trace_kernel_ms; /* of which: the dominant kernel's launches (k_paths / k_whitted; the ray-list kernel k_rays in the wavefront pipeline) */
    uint32_t trace_launches; /* how many such launches */
    uint32_t trace_pipeline; /* after rt_render: 0 = lock-step kernel (one launch per sampling round), 1 = wavefront pipeline (one k_rays launch per bounce); after rt_trace_brief: 0 = the reference's binary BVH walk, 2 = the four-wide walk of the frames' rays (RT_QUERY_WIDE=1 in the environment at rt_init: a test gate) */
    uint64_t phase_cycles[16]; /* diagnostic builds only (RT_PROFILE_PHASES): per kernel phase wave clocks [0..3], executions [4..7], active lanes [8..11] */
    double bvh_build_ms;    /* GPU time the last rt_scene_upload spent building BVHs (0: the scene brought them) */
} rt_counters;

/* ---- context ---- */
int rt_abi_version(void);
int rt_init(int device, rt_ctx **out);
void rt_shutdown(rt_ctx *ctx);
const char *rt_last_error(const rt_ctx *ctx); /* ctx may be null: last error of a failed rt_init */

/* ---- scene / camera: replaces the reference's in-memory Scene/Camera walk ---- */
int rt_scene_upload(rt_ctx *ctx, const rt_scene_desc *scene);
int rt_camera_set(rt_ctx *ctx, const rt_camera *cam);

/* ---- frame: replaces the thread pool + threadRenderTiles (src/main.cpp:271-385,599-606) ---- */
int rt_render(rt_ctx *ctx, const rt_render_params *params, rt_frame_out *out);

/* ---- frame exchange between GPUs (no counterpart in the reference, which is one process): a rank's finished pixels
 * as one array of 16-byte records (rgb, z, sample count, object id) in tile order, ready for ONE all-gather, and the
 * way back.  Every tile takes tile_size^2 records, so all ranks' arrays have rt_frame_records() entries.  `planes`
 * are DEVICE pointers (rgb, z, sample_count and object_id all non-null); both calls return when the GPU is done. ---- */
uint32_t rt_frame_records(int width, int height, int tile_size, int world);
int rt_frame_pack(rt_ctx *ctx, const rt_frame_out *planes, int tile_size, int rank, int world, void *packed);
/* packed_all: the world ranks' arrays in rank order (what the all-gather delivers); skip_rank's pixels are left alone */
int rt_frame_unpack(rt_ctx *ctx, const void *packed_all, int tile_size, int skip_rank, int world, const rt_frame_out *planes);

/* ---- stopping and watching a frame: Renderer::StopRender (src/renderer.h:229, "should wait till rendering threads
 * stop") and RenderImage::GetNumRenderedPixels / IsRenderDone (src/renderer.h:72-75), which the viewer polls from its
 * idle loop while the render threads run (src/viewport.cpp:568-600, 625, 766-787).  These two are the only entry
 * points that may be called from ANOTHER thread while rt_render or rt_photons_fill runs on the context. ---- */
/* Ask the call in progress to stop: no new work is handed to the GPU (kernels in flight finish the paths they hold:
 * milliseconds), rt_render / rt_photons_fill return RT_ERR_CANCELLED.  Cleared when the next such call starts. */
int rt_cancel(rt_ctx *ctx);
typedef struct rt_progress_info {
    uint64_t pixels_total;     /* pixels of this context's tile shard */
    uint64_t pixels_done;      /* of which finished: released by the stop test or at max_spp, planes written */
    uint64_t round_items;      /* (pixel, sample) camera paths of the sampling round in progress */
    uint64_t round_items_done; /* of which handed to the GPU and finished, approximately (read while the kernels run) */
    int32_t round;             /* sampling rounds started so far, 0 before the first */
    int32_t rendering;         /* 1 while rt_render is inside its rounds */
} rt_progress_info;
int rt_progress(rt_ctx *ctx, rt_progress_info *out);

/* ---- batch queries: Renderer::TraceRay / TraceShadowRay (src/main.cpp:115-125)
 * and tracePath (src/main.cpp:160-268) for arrays of rays. rays6 = origin xyz,
 * direction xyz (direction need not be normalised). Host pointers. ---- */
int rt_trace(rt_ctx *ctx, const float *rays6, int n, rt_hit *out);
int rt_trace_shadow(rt_ctx *ctx, const float *rays6, const float *t_max, int n, int32_t *occluded);
/* path i is traced with PCG stream i+1 of the given seed */
int rt_trace_paths(rt_ctx *ctx, const float *rays6, int n, uint64_t seed,
                   const rt_render_params *params, float *radiance3);

/* The same two queries answered by the persistent ray-list kernel (a lane that finishes a ray
 * takes the next one at once): t_max == NULL asks for closest hits, otherwise for any hit
 * below t_max[i]. Results are brief - no hit point, normal or uvw:
 *   closest: t (FLT_MAX = none), node, face, front, u, v;   any-hit: node = 1 if occluded else 0.
 * rt_get_counters().kernel_ms afterwards holds the kernel's own duration. */
typedef struct rt_hit_brief {
    float t;
    int32_t node;
    uint32_t face;
    int32_t front;
    float u, v;
} rt_hit_brief;
int rt_trace_brief(rt_ctx *ctx, const float *rays6, const float *t_max, int n, rt_hit_brief *out);

/* rt_trace_brief for callers that cannot allocate device memory (tight serving loops, harnesses that forbid
 * hipMalloc, code that runs between the capture and the replay of HIP graphs).  No counterpart in the reference.
 * All three arrays are DEVICE memory (t_max may be null); the answers are those of rt_trace_brief, bit for bit.
 * A negative t_max entry lets nothing occlude its ray (the walk compares t < t_max): the answer is "not
 * occluded"; the answer for a NaN entry is undefined.  rt_trace_brief refuses both on the host; this call cannot
 * look at device memory there.  The call
 * makes no hipMalloc / hipFree: the kernel's work queue, counters and the BVH-stack levels beyond LDS live in a
 * static 32 MiB device array of the library, one per GPU, shared by every context and call on that GPU - calls
 * that overlap in time on one GPU race.  At most RT_NOALLOC_MAX_RAYS rays per call (RT_ERR_INVALID beyond);
 * RT_ERR_UNSUPPORTED if the scene's BVH needs more stack than the array holds for one workgroup.  It enqueues a
 * memset and a kernel on the context's own stream and returns after the answers are complete (it synchronises
 * that stream, so it cannot itself be called while a stream is being captured); only kernel_ms of
 * rt_get_counters is updated. */
#define RT_NOALLOC_MAX_RAYS (1 << 24)
int rt_trace_brief_noalloc(rt_ctx *ctx, const float *d_rays6, const float *d_t_max, int n, rt_hit_brief *d_out);

/* ---- known-answer queries for tests: the samplers and random streams the kernels use, one evaluation per query, so
 * that fixtures made from the reference's own samplers can be compared with the device code value for value.
 *   RT_SAMPLER_LIGHT            PointLight::GenerateSample at p and GetSampleInfo for `dir` (src/lights.h:208-288), lights[0]
 *   RT_SAMPLER_MATERIAL         MtlBlinn::GenerateSample and GetSampleInfo for `dir` (src/materials.h:87-303)
 *   RT_SAMPLER_PHOTON_MATERIAL  MtlBlinn::GenerateSample of the project-12 snapshot, which scatters photons (project12.zip src/materials.h:82-193)
 *   RT_SAMPLER_PHOTON_EMIT      PointLight::RandomPhoton (src/lights.h:92-124): origin, dir, mult = the photon's colour
 * Every query draws from PCG stream (seq, seed) (RNG::SetSequence, src/rng.h:39-46); next_u32 is the stream's next
 * number after the call, which pins how many it drew.  Host pointers. ---- */
enum { RT_SAMPLER_LIGHT = 0, RT_SAMPLER_MATERIAL = 1, RT_SAMPLER_PHOTON_MATERIAL = 2, RT_SAMPLER_PHOTON_EMIT = 3 };
typedef struct rt_sampler_query {
    float p[3], n[3], v[3]; /* surface point, unit shading normal, unit direction towards the viewer */
    int32_t front;
    float dir[3];           /* direction whose probability GetSampleInfo is asked for */
    int32_t material;
    uint64_t seq, seed;
} rt_sampler_query;
typedef struct rt_sampler_answer {
    float dir[3], mult[3], prob, dist;
    int32_t ok, lobe;       /* GenerateSample's return value and lobe */
    float pdf;              /* GetSampleInfo(dir).prob */
    uint32_t next_u32;
    float origin[3];        /* RT_SAMPLER_PHOTON_EMIT: where the photon starts */
    float pad;
} rt_sampler_answer;
int rt_debug_samplers(rt_ctx *ctx, int which, const rt_sampler_query *queries, int n, rt_sampler_answer *out);
/* n_draws numbers of each of n_streams PCG streams: RNG::RandomUInt (words) and, from the stream seeded again,
 * RNG::RandomFloat (floats) (src/rng.h:48-65); both arrays are [n_streams][n_draws].  Needs no scene. */
int rt_debug_rng(rt_ctx *ctx, const uint64_t *seq, const uint64_t *seed, int n_streams, int n_draws, uint32_t *words, float *floats);

/* ---- photon maps: replaces class PhotonMap (src/photonmap.h:38-414) and the emission loops of the
 * project-12 snapshot's main() (project12.zip src/main.cpp:262-470) ---- */

/* One photon, bit-compatible with PhotonMap::PhotonData (src/photonmap.h:42-59): 32 bytes. */
typedef struct rt_photon {
    float pos[3];
    float power;    /* largest colour channel (scaled by 1/stored once the map is built) */
    float dir[3];   /* direction of travel */
    uint8_t rgb[3]; /* colour / power, 8 bits per channel (Color24) */
    uint8_t plane;  /* kd-tree split axis 0..2 of a node that has children */
} rt_photon;

enum { RT_MAP_GLOBAL = 0, RT_MAP_CAUSTICS = 1 };
#define RT_GATHER_MAX_PHOTONS 128 /* EstimateIrradiance<128> (project12.zip src/materials.cpp:54,65) */

typedef struct rt_photon_params {
    int32_t flags;        /* RT_PM_* bits */
    uint32_t n_global;    /* capacity of the global map   (photonMap.Resize(n));   reference: 100000 */
    uint32_t n_caustic;   /* capacity of the caustics map (causticsMap.Resize(n)); reference: 100000 */
    int32_t max_segments; /* bound on the TraceRay calls of one photon path; the reference has none
                             (a path ends by absorption), 64 loses less than 1e-6 of the photons of its scene */
    uint64_t seed;        /* path p of the global map draws from PCG stream p of seed ^ 0x50484F54,
                             path p of the caustics map from stream p of seed ^ 0x43415553 */
} rt_photon_params;

/* Trace photon paths [path_begin, path_begin + path_count) of one map and return the photons they
 * store, in path order (within a path in the order stored) - the order in which the reference's
 * serial loop would have added them.  At most `capacity` photons are written (the first ones in that
 * order); *n_stored receives the number the paths stored in total, which may exceed capacity.
 * Powers are unscaled.  `out` is host memory, or device memory if device_memory != 0. */
int rt_photons_emit(rt_ctx *ctx, const rt_photon_params *params, int which, uint64_t path_begin, uint64_t path_count,
                    rt_photon *out, uint32_t capacity, uint64_t *n_stored, int device_memory);

/* Make `photons[0..n)` (emission order, unscaled) the context's map `which`: powers are scaled by 1/n
 * (ScalePhotonPowers, src/photonmap.h:82), the left-balanced kd-tree is built exactly as
 * PrepareForIrradianceEstimation / BalanceSegment do (src/photonmap.h:176-264) and uploaded. n = 0 clears the map. */
int rt_photons_build(rt_ctx *ctx, int which, const rt_photon *photons, uint32_t n, int device_memory);

/* One GPU: emit rounds of paths until each map selected by params->flags holds its capacity, then build. */
int rt_photons_fill(rt_ctx *ctx, const rt_photon_params *params);

/* The balanced map as PhotonMap::GetPhotons() exposes it (heap order, node i at index i-1). dst may be null to query *n. */
int rt_photons_download(rt_ctx *ctx, int which, rt_photon *dst, uint32_t *n);

/* EstimateIrradiance<128>(irrad, dir, radius, pos, normal, 1) (src/photonmap.h:268-414) for n points.
 * pos3 / normal3 / irradiance3 are host arrays of 3 floats per point. */
int rt_photons_irradiance(rt_ctx *ctx, int which, const float *pos3, const float *normal3, int n, float radius,
                          float *irradiance3);

int rt_get_counters(rt_ctx *ctx, rt_counters *out);
/* box_tests / tri_tests are only counted while counting is on (a separately
 * compiled kernel variant; the default variant carries no per-test counters) */
int rt_set_counting(rt_ctx *ctx, int on);

/* ---- host-side helpers (no GPU needed): our own readers for the reference's
 * scene.xml schema (src/xmlload.cpp:80-270) and OBJ meshes (cyTriMesh.h:264-497),
 * and the reference's BVH build (cyBVH.h:122-328) ---- */
typedef struct rt_scene_host rt_scene_host; /* owns all arrays a desc points into */
int rt_scene_load_xml(const char *path, rt_scene_host **out, char *err, size_t err_len);
int rt_scene_from_xml_string(const char *xml, const char *base_dir, rt_scene_host **out, char *err, size_t err_len);
const rt_scene_desc *rt_scene_host_desc(const rt_scene_host *s);
const rt_camera *rt_scene_host_camera(const rt_scene_host *s);
void rt_scene_host_free(rt_scene_host *s);
/* Append an in-memory triangle mesh (positions + tex coords, one index set for
 * both, normals computed as the reference's ComputeNormals does) to a loaded
 * scene and bind it to the node called node_name. */
int rt_scene_host_set_mesh(rt_scene_host *s, const char *node_name, const float *v, const float *vt,
                           uint32_t nv, const uint32_t *f, uint32_t nf, char *err, size_t err_len);

/* on != 0: the scene's meshes go to rt_scene_upload without a BVH (the BVHs built so far are dropped, later
 * rt_scene_host_set_mesh calls build none), so the library builds its own on the GPU; 0: back to the reference's
 * BVH, built on the host. */
int rt_scene_host_set_gpu_bvh(rt_scene_host *s, int on);
/* rt_scene_desc.face_materials of a loaded scene: 0 (as loaded) shades every face of an OBJ object that brings its own
 * materials with the first of them, as the reference does; 1 shades a face with the material its `usemtl` group names. */
int rt_scene_host_set_face_materials(rt_scene_host *s, int on);
/* Warnings of the loader (things the reference reports with a printf and goes on): n-th text, or NULL past the last. */
const char *rt_scene_host_warning(const rt_scene_host *s, uint32_t n);

/* A PNG file in memory -> 8-bit RGB rows, top row first (what the reference asks lodepng for when it loads a texture,
 * src/texture.cpp:77-87).  *rgb is freed with rt_free. */
int rt_png_decode_rgb8(const uint8_t *file, size_t size, int32_t *width, int32_t *height, uint8_t **rgb, char *err, size_t err_len);

/* BVH build over nf triangles; returns arrays the caller frees with rt_free. */
int rt_bvh_build(const float *v, uint32_t nv, const uint32_t *f, uint32_t nf, uint32_t max_leaf,
                 rt_bvh_node **nodes, uint32_t *n_nodes, uint32_t **elements, uint32_t *depth);
void rt_free(void *p);

#ifdef __cplusplus
}
#endif
#endif /* RT_MI355X_H */
