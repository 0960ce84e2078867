// Device-resident form of a scene.  Built once per rt_scene_upload; everything a
// ray needs is reachable from one DScene passed to the kernels by value.
#pragma once
#include "rt_mi355x.h"
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtd {

constexpr int kMaxSceneDepth = 6; // levels of the node hierarchy below and including the root

// Scene-graph node, depth-first order.  Read with wave-uniform indices, so the
// loads are scalar.
struct DNode {
    float itm[12];
    float tm[12];
    int32_t parent;
    int32_t prim;
    int32_t mesh;
    int32_t material;
    int32_t depth;        // root = 0
    int32_t has_children;
    int32_t chain[kMaxSceneDepth]; // ancestors root-first, chain[depth] = this node
    float wbox[6];        // world-space box around the node's own object, padded: min xyz, max xyz (-/+FLT_MAX: unknown)
    int32_t pad[2];
}; // 176 bytes

// Scene-graph node as the ray-list kernel reads it: one 128-byte record, fetched with a wave-uniform index through
// the scalar cache.  A mesh node carries the root of its BVH (box and data word), so a ray enters the walk without
// another dependent fetch.
struct TNode {
    float itm[12];
    uint32_t meta;    // prim | depth << 8 | has_children << 16
    uint32_t root;    // mesh: data word of the BVH root (first child / leaf range, pool numbering)
    uint32_t root4;   // mesh: the same for the four-wide form of the BVH (record index in bvh4_pool / leaf range)
    float box[6];     // mesh: root box as the slab test wants it: min.x, max.x, min.y, max.y, min.z, max.z
    float wbox[6];    // world-space box around the node's own object, padded: min xyz, max xyz (-/+FLT_MAX: unknown)
    uint32_t pad[5];
};
static_assert(sizeof(TNode) == 128, "one TNode per 128-byte line");

// Hierarchy over the padded world-space boxes of the object nodes (the reference computes such boxes, src/scene.h:434-444,
// and never looks at them: its rays are tested against every object, src/main.cpp:16-89).  Depth-first order with a skip
// link: a walk goes to the next record when a box is reached and to `skip` when it is not - no stack, and with a
// wave-uniform cursor every record arrives through the scalar cache.
struct TlasNode {
    float box[6];   // min xyz, max xyz
    int32_t object; // scene node of a leaf, -1 for an inner record
    int32_t skip;   // record to go on with when the box is not reached (= the end of this record's subtree)
};
static_assert(sizeof(TlasNode) == 32, "two records per 64-byte line");

// One triangle mesh.  The BVH nodes and triangle records of all meshes of a scene live in two pools (one
// allocation each); node ids and triangle offsets inside the BVH data words are pool indices.
//   bvh  : the node pool.  32-byte nodes, two float4 each: (min.x, max.x, min.y, max.y) (min.z, max.z, data, 0);
//          the two children of a node are adjacent, the pair starts on a 64-byte line
//   tris : the triangle pool.  48-byte records in BVH-leaf order (indexed by element slot, not face id):
//          (v0.xyz, v1.x) (v1.yz, v2.xy) (v2.z, face id, 0, 0) - the three position
//          reads and the two index reads of the reference collapse into one dependent fetch
struct DMesh {
    const float4 *bvh;
    const float4 *tris;
    uint32_t root;      // pool id of this mesh's root node
    uint32_t pad_;
    const uint32_t *fn; // normal indices per face
    const uint32_t *ft; // tex-coord indices per face
    const float *vn;
    const float *vt;
    uint32_t n_nodes;
    uint32_t depth;
    const uint32_t *fmtl; // sub-material per face (rt_mesh::face_material), null unless the scene asks for per-face materials
    // What shading a hit needs of its face, in one 64-byte record (one memory line instead of four: the face's normal
    // indices and the three normals they name are 12-byte reads at unrelated addresses): four float4 per face =
    // n0.xyz n1.x | n1.yz n2.xy | n2.z, sub-material, -, - | unused
    const float4 *face_rec;
};

// One texture map (rt_texture with its pixels in device memory).
struct DTexture {
    int32_t type;
    float itm[12];
    float color1[3], color2[3];
    const uint8_t *pixels;
    int32_t width, height;
};

struct DScene {
    const DNode *nodes;
    const TNode *tnodes;       // the same nodes, compact (ray-list kernel)
    const TlasNode *tlas;      // hierarchy over the object nodes' world boxes; null for scenes of a few objects
    int32_t n_tlas;
    const float4 *bvh_pool;    // BVH nodes of all meshes
    const float4 *tri_pool;    // triangle records of all meshes
    // Four-wide BVHs over the same triangles, which the frames' extension and shadow rays walk: 64-byte records of up
    // to four children with 8-bit boxes (lbvh.h: wide_record_pack), four float4 per record.  The data word of an inner
    // child is the index of its own record; a slot without a child holds 0x7FFFFFFF.
    const float4 *bvh4_pool;
    const float4 *tri4_pool;   // triangle records in the leaf order of the four-wide BVH (the same array as tri_pool when both forms are one tree)
    const DMesh *meshes;
    const rt_material *materials;
    const rt_light *lights;
    int32_t n_nodes;
    int32_t n_lights;
    int32_t face_materials; // some mesh carries fmtl
    float background[3];
    float environment[3]; // what a secondary ray of the Whitted-style shader sees when it leaves the scene
    int32_t stack_levels; // BVH stack entries a lane may need (deepest mesh + 2), 0 without meshes
    int32_t lds_levels;   // of which in LDS; the rest in stack_overflow
    uint2 *gather_heaps;   // [photon-lookup slots of the largest grid][66]: last level of the slots' heaps of nearest photons
    uint32_t *gather_ids;  // [slots][128]: the photons found
    uint2 *stack_overflow; // [(stack_levels - lds_levels)][threads of the largest grid]
    const rt_photon *pmap[2]; // balanced photon maps (global, caustics), node 1 first; null when not built
    uint32_t pmap_n[2];
    const DTexture *textures;
    int32_t background_texture, environment_texture; // 1 + index, 0 = none
    // two words of pinned host memory, read and written in place (system scope):
    const uint32_t *cancel;       // non-zero: hand out no more work (rt_cancel); read by the persistent kernels when they draw from their queues
    unsigned long long *progress; // items of the running lock-step kernel's queue handed out so far (rt_progress)
};

// Frame constants (camera + sampling), by value in the kernel arguments.
struct DFrame {
    float cam_pos[3];
    float cam2world[9];
    float half_w, half_h, pixel_size, focaldist, dof;
    int32_t width, height, srgb;
    int32_t min_spp, max_spp;
    float delta_max;
    int32_t max_bounces;
    float bias;
    int32_t halton_period;
    uint64_t seed;
    int32_t photon_flags; // RT_PM_* bits the Whitted-style shader honours
    float gather_radius;
    uint32_t chunk;       // items a wave of the path kernel draws from the work queue at a time
    uint32_t pad_;
};

struct DCounters {
    unsigned long long closest, shadow, paths, box_tests, tri_tests, gathers, photon_visits;
    unsigned long long phase[16]; // diagnostic: wave clock cycles per phase (RT_PROFILE_PHASES builds)
};

} // namespace rtd
