// k_rays: closest-hit / any-hit queries for a list of rays, as a persistent kernel
// with ray-level regeneration: a lane that finishes its ray takes the next one from
// the list at once, so the lanes of a wave stay busy although rays need anything from
// 3 to 60 BVH node visits (src/main.cpp:16-125, src/TriObj.cpp:12-229).  The lane
// carries nothing but the ray and its walk, so the kernel runs at high occupancy.
//
// Where rays come from and where answers go is a policy (IO): a plain list for the batch
// query entry point (RayListIO below), the path pool of the wavefront renderer (wavefront.h).
//   bool fetch(e, o, d, any_hit, t_max)   entry e of the list; false = no ray there
//   void put_closest(e, t, node, face, front, u, v)   t = FLT_MAX, node = -1: nothing hit
//   void put_shadow(e, occluded)
#pragma once
#include "kernels.h"

namespace rtd {

// ray_o = origin xyz, w = t_max for an any-hit (shadow) ray, negative for a closest-hit ray; ray_d = direction
// hit = closest: (t, node id bits, face | front << 31, u), hit_v = v;  any-hit: (1 or 0, 0, 0, 0)
struct RayListIO {
    const float4 *ray_o, *ray_d;
    float4 *hit;
    float *hit_v;
    RT_DEV bool fetch(uint32_t e, float4 &o, float4 &d, bool &any_hit, float &t_max) const
    {
        o = ray_o[e]; d = ray_d[e];
        any_hit = !(o.w < 0.0f); t_max = o.w;
        return true;
    }
    RT_DEV void put_closest(uint32_t e, float t, int node, uint32_t face, int front, float u, float v) const
    {
        hit[e] = make_float4(t, __uint_as_float((uint32_t)node), __uint_as_float(face | ((uint32_t)front << 31)), u);
        hit_v[e] = v;
    }
    RT_DEV void put_shadow(uint32_t e, bool occluded) const { hit[e] = make_float4(occluded ? 1.0f : 0.0f, 0.0f, 0.0f, 0.0f); }
};

// The same list read and answered in the caller's own formats (six floats per ray, an optional t_max array,
// rt_hit_brief records): nothing to convert on the host, so an entry point built on it needs no staging buffers.
struct BriefIO {
    const float *rays6, *t_max; // t_max null: closest hits
    rt_hit_brief *out;
    RT_DEV bool fetch(uint32_t e, float4 &o, float4 &d, bool &any_hit, float &tm) const
    {
        const float *r = rays6 + (size_t)e * 6;
        any_hit = t_max != nullptr;
        tm = any_hit ? t_max[e] : -1.0f;
        o = make_float4(r[0], r[1], r[2], tm); d = make_float4(r[3], r[4], r[5], 0.0f);
        return true;
    }
    RT_DEV void put_closest(uint32_t e, float t, int node, uint32_t face, int front, float u, float v) const
    {
        out[e] = rt_hit_brief{t, node, face & 0x7FFFFFFFu, front, u, v};
    }
    RT_DEV void put_shadow(uint32_t e, bool occluded) const { out[e] = rt_hit_brief{0.0f, occluded ? 1 : 0, 0u, 0, 0.0f, 0.0f}; }
};

#ifndef RT_RAYS_MIN_WAVES
#define RT_RAYS_MIN_WAVES 4 // 120 registers, nothing spilled: faster than 6 waves with the walk's temporaries in scratch
#endif

#ifndef RT_RAYS_BVH_MIN_LANES
#define RT_RAYS_BVH_MIN_LANES 32 // keep taking BVH steps while at least this many lanes are inside a mesh
#endif
#ifndef RT_RAYS_LEAF_MIN_LANES
#define RT_RAYS_LEAF_MIN_LANES 16 // run triangle tests once this many lanes are parked at a leaf
#endif
#ifndef RT_RAYS_SCENE_MIN_LANES
#define RT_RAYS_SCENE_MIN_LANES 16 // run a pass over the scene nodes once this many lanes wait outside a mesh
#endif
#ifndef RT_RAYS_REFILL_MIN_LANES
#define RT_RAYS_REFILL_MIN_LANES 24 // fetch new rays once this many lanes are idle
#endif

constexpr uint32_t kRayChunk = 256; // rays a wave reserves per queue atomic unless the launch says otherwise
constexpr uint32_t kNoNode = 0x7FFFFFFFu; // not a data word: an inner node's first child index is far smaller

// One lane's ray and its walk, and the steps a wave takes with them.  Used by k_rays below and by
// the path kernel (pathsm.h); the members live in registers.
template <bool COUNT, int LEVELS, class IO> struct RayWalker {
    bool active = false, any_hit = false, in_bvh = false;
    uint32_t ray_i = 0;
    Ray wray{mk(0, 0, 0), mk(0, 0, 1)}; // world space
    float t_max = 0.0f;
    int node_i = 0;                      // scene node being visited
    RayLevels<LEVELS> lv;                // local rays of the hierarchy levels above it
    float best_t = FLT_MAX; int best_node = -1, best_front = 1; uint32_t best_face = 0; float best_u = 0, best_v = 0;
    Ray lr{mk(0, 0, 0), mk(0, 0, 1)}; V3 inv = mk(0, 0, 0); bool exact = false;
    GF4 bvh = nullptr, tris = nullptr;
    int sp = 0; float t_limit = 0.0f;
    bool m_hit = false; uint32_t m_face = 0; float m_u = 0, m_v = 0, m_det = 0;
    bool at_leaf = false; uint32_t leaf_data = 0, leaf_i = 0;
    uint32_t cur = kNoNode; // node to visit next (its data word), kNoNode: take one from the stack



    RT_DEV void start(uint32_t e, float4 o, float4 d, bool shadow, float tmax, PathTally &pt)
    {
        ray_i = e;
        wray.p = mk(o.x, o.y, o.z); wray.d = mk(d.x, d.y, d.z);
        any_hit = shadow; t_max = tmax;
        if (any_hit) ++pt.shadow; else ++pt.closest;
        node_i = 0; in_bvh = false; active = true;
        best_t = FLT_MAX; best_node = -1; best_front = 1; best_face = 0; best_u = 0; best_v = 0;
    }
    RT_DEV void shadow_done(const IO &io, bool occluded)
    {
        io.put_shadow(ray_i, occluded);
        active = false;
    }

    // The scene pass, if enough lanes wait for it, then BVH steps while enough lanes are inside a mesh.
    // Called by the whole wave; every call makes progress if any lane has a ray.
    RT_DEV void phases(const DScene &S, const IO &io, const LaneStack &st, PathTally &pt, Tally<COUNT> &tl)
    {
        const float epsilon = 1e-6f;
    // ================= scene-graph steps of rays that are not inside a mesh (main.cpp:16-89) =================
    // One pass over the node list in step: the node index is wave-uniform, so node matrices arrive
    // through scalar loads; a lane takes part from the node it has got to, and stays in the pass
    // until it enters a mesh or runs off the end of the list.
    const uint64_t stepping = __ballot(active && !in_bvh);
    if (stepping != 0 && ((uint32_t)__popcll(stepping) >= (uint32_t)RT_RAYS_SCENE_MIN_LANES || !__any(in_bvh))) {
        pt.tock(12);
        for (int n = 0; n < S.n_nodes; ++n) {
            if (!__any(active && !in_bvh && node_i == n)) continue;
            const DNode &nd = S.nodes[n];
            if (active && !in_bvh && node_i == n) {
                const int d = nd.depth;
                const Ray parent = (d == 0) ? wray : lv.get(d - 1);
                lr = to_node(nd, parent);
                if (nd.has_children) lv.set(d, lr);
                const int prim = nd.prim;
                bool advance = true;
                if (prim == RT_PRIM_SPHERE || prim == RT_PRIM_PLANE) {
                    if (any_hit) {
                        const bool occ = prim == RT_PRIM_SPHERE ? sphere_shadow(lr, t_max) : plane_shadow(lr, t_max);
                        if (occ) { shadow_done(io, true); advance = false; }
                    } else {
                        float t = 0; int front = 1;
                        const bool ok = prim == RT_PRIM_SPHERE ? sphere_t(lr, t, front) : plane_t(lr, t, front);
                        if (ok && t < best_t) { best_t = t; best_node = n; best_front = front; best_face = 0; best_u = 0; best_v = 0; }
                    }
                } else if (prim == RT_PRIM_MESH) {
                    const DMesh &M = S.meshes[nd.mesh];
                    bvh = (GF4)M.bvh; tris = (GF4)M.tris;
                    inv = mk(1.0f / lr.d.x, 1.0f / lr.d.y, 1.0f / lr.d.z);
                    exact = needs_exact_slabs(lr.p, inv);
                    t_limit = any_hit ? t_max : FLT_MAX;
                    m_hit = false;
                    const f4 r0 = bvh[2], r1 = bvh[3];
                    float dist;
                    tl.boxes(1);
                    if (bvh_box(exact, lr.p, inv, t_limit, r0, r1, dist)) {
                        cur = __float_as_uint(r1.z); // the root is the node to visit next; the stack is empty
                        sp = 0; in_bvh = true; advance = false;
                    }
                }
                if (advance) node_i = n + 1;
            }
        }
        if (active && !in_bvh && node_i >= S.n_nodes) {
            if (any_hit) shadow_done(io, false); // nothing in the way
            else {
                // closest ray: the spherical lights last (main.cpp:52-70, lights.h:129-156), front side only
                for (int l = 0; l < S.n_lights; ++l) {
                    const rt_light &L = S.lights[l];
                    if (L.type != RT_LIGHT_POINT) continue;
                    Ray r;
                    r.p = (wray.p - ld3(L.position)) / L.size;
                    r.d = wray.d / L.size;
                    const float a = dot(r.d, r.d);
                    const float b = 2 * dot(r.d, r.p);
                    const float c = dot(r.p, r.p) - 1.0f;
                    const float disc = b * b - 4 * a * c;
                    if (disc < 0.0f) continue;
                    const float s = fsqrt(disc);
                    const float inv2a = 1.0f / (2.0f * a);
                    const float t1 = (-b - s) * inv2a;
                    if (t1 <= 0.0f) continue;
                    if (t1 < best_t) { best_t = t1; best_node = -2 - l; best_front = 1; break; }
                }
                io.put_closest(ray_i, best_t, best_node, best_face, best_front, best_u, best_v);
                active = false;
            }
        }
        pt.ran(1, stepping);
    }

    // ================= BVH steps (TriObj.cpp:30-122 / 149-226) =================
    // The node to visit next is kept in a register (cur): of two children that are hit only the
    // farther one goes to the stack, the nearer one is visited in the next trip - exactly the
    // order of the reference, which pushes both and pops the nearer one at once (its entry
    // distance was just tested against the current closest hit, so the pop-time test of the
    // reference cannot fail for it and is not repeated).  The stack is read only when a visit
    // leaves nothing to go on with.
    // A lane whose next node is a leaf parks there: triangle tests are long and only a few
    // lanes reach a leaf in any one trip, so they run for many lanes at once.
    while (__any(in_bvh)) {
        pt.tock(12);
        const uint64_t walkers = __ballot(in_bvh && !at_leaf);
        if (in_bvh && !at_leaf) {
            if (cur == kNoNode && sp > 0) {
                --sp;
                uint32_t data; float nodeT;
                st.pop(sp, data, nodeT);
                if (!(t_limit < nodeT)) cur = data; // else: entered farther than the best hit so far
            }
            if (cur != kNoNode) {
                if (cur & kLeafBit) { at_leaf = true; leaf_data = cur; leaf_i = 0; cur = kNoNode; }
                else {
                    const uint32_t c1 = cur & 0x7FFFFFFFu;
                    const f4 a0 = bvh[c1 * 2], a1 = bvh[c1 * 2 + 1], b0 = bvh[c1 * 2 + 2], b1 = bvh[c1 * 2 + 3];
                    float d1, d2;
                    tl.boxes(2);
                    bool h1 = bvh_box(exact, lr.p, inv, t_limit, a0, a1, d1);
                    h1 = h1 && d1 < t_limit;
                    bool h2 = bvh_box(exact, lr.p, inv, t_limit, b0, b1, d2);
                    h2 = h2 && d2 < t_limit;
                    const uint32_t k1 = __float_as_uint(a1.z), k2 = __float_as_uint(b1.z);
                    const bool firstNear = d1 < d2; // on a tie the second child is visited first
                    if (h1 && h2) { st.push(sp, firstNear ? k2 : k1, firstNear ? d2 : d1); ++sp; }
                    cur = h1 ? ((h2 && !firstNear) ? k2 : k1) : (h2 ? k2 : kNoNode);
                }
            }
        }
        pt.ran(2, walkers);
        const uint32_t n_leaf = (uint32_t)__popcll(__ballot(at_leaf));
        const uint32_t n_walk = (uint32_t)__popcll(__ballot(in_bvh && !at_leaf));
        if (n_leaf != 0 && (n_leaf >= (uint32_t)RT_RAYS_LEAF_MIN_LANES || n_leaf >= n_walk)) {
            const uint64_t parked = __ballot(at_leaf);
            if (at_leaf) { // two triangles of the leaf per visit: six loads in flight together
                const uint32_t off = leaf_data & 0x0FFFFFFFu;
                const uint32_t cnt = ((leaf_data >> 28) & 7u) + 1;
                const GF4 ra = &tris[(off + leaf_i) * 3];
                const bool two = leaf_i + 1 < cnt;
                const GF4 rb = two ? ra + 3 : ra;
                const f4 a0 = ra[0], a1 = ra[1], a2 = ra[2], b0 = rb[0], b1 = rb[1], b2 = rb[2];
                float t, u, v, det; uint32_t face;
                bool stop = false;
                tl.tris(1);
                if (tri_test(a0, a1, a2, lr, t, u, v, det, face) && !(t <= epsilon || t > t_limit)) {
                    m_hit = true; m_det = det; m_u = u; m_v = v; m_face = face; t_limit = t;
                    stop = any_hit; // first hit ends a shadow ray
                }
                if (two && !stop) {
                    tl.tris(1);
                    if (tri_test(b0, b1, b2, lr, t, u, v, det, face) && !(t <= epsilon || t > t_limit)) {
                        m_hit = true; m_det = det; m_u = u; m_v = v; m_face = face; t_limit = t;
                        stop = any_hit;
                    }
                }
                leaf_i += 2;
                if (stop) { sp = 0; at_leaf = false; } // cur is kNoNode while parked
                else if (leaf_i >= cnt) at_leaf = false;
            }
            pt.ran(3, parked);
        }
        if (in_bvh && !at_leaf && cur == kNoNode && sp == 0) { // mesh done: fold its result into the ray's (main.cpp:38-48 / 80-85)
            in_bvh = false;
            if (any_hit) {
                if (m_hit) shadow_done(io, true);
                else ++node_i;
            } else {
                if (m_hit && t_limit < best_t) {
                    best_t = t_limit; best_node = node_i; best_front = m_det > 0.0f; best_face = m_face; best_u = m_u; best_v = m_v;
                }
                ++node_i;
            }
        }
        if ((uint32_t)__popcll(__ballot(in_bvh)) < (uint32_t)RT_RAYS_BVH_MIN_LANES) break;
    }
    }
};

template <bool COUNT, int LEVELS, class IO>
__global__ __launch_bounds__(kBlock, RT_RAYS_MIN_WAVES) void k_rays(DScene S, IO io, uint32_t n_rays, DCounters *ctr,
                                                                  unsigned long long *queue, uint32_t ray_chunk)
{
    extern __shared__ uint32_t lds[];
    const LaneStack st = lane_stack(lds, S);
    const uint32_t lane = threadIdx.x & 63u;
    PathTally pt;
    Tally<COUNT> tl;
    RayWalker<COUNT, LEVELS, IO> w;
    w.lv.init(w.wray);

    // wave-uniform window of the ray list
    uint32_t chunk_next = 0, chunk_end = 0;
    bool exhausted = false;

    // A lane is idle, stepping through the scene nodes, walking a BVH, or parked at a leaf.  The wave
    // runs each kind of step for all lanes that wait for it at once, and lets lanes wait until enough
    // of them do (thresholds above) or nothing else can run.  The schedule decides only when a lane
    // takes its next step, never what the step computes.
    pt.tick();
    for (;;) {
        // ================= refill idle lanes from the list =================
        const uint64_t idle = __ballot(!w.active);
        pt.tock(12);
        if (idle != 0 && !exhausted && ((uint32_t)__popcll(idle) >= (uint32_t)RT_RAYS_REFILL_MIN_LANES || !__any(w.in_bvh))) {
            if (chunk_next >= chunk_end) {
                unsigned long long b = 0;
                if (lane == 0) b = atomicAdd(queue, (unsigned long long)ray_chunk);
                b = __shfl(b, 0, 64);
                chunk_next = b < n_rays ? (uint32_t)b : n_rays;
                chunk_end = b + ray_chunk < n_rays ? (uint32_t)(b + ray_chunk) : n_rays;
                if (chunk_next >= chunk_end) exhausted = true;
            }
            const uint32_t avail = chunk_end - chunk_next;
            const uint32_t want = (uint32_t)__popcll(idle);
            const uint32_t take = avail < want ? avail : want;
            const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
            if (!w.active && rank < take) {
                const uint32_t e = chunk_next + rank;
                float4 o, d; bool shadow; float tmax;
                if (io.fetch(e, o, d, shadow, tmax)) w.start(e, o, d, shadow, tmax, pt); // a list may have holes
            }
            chunk_next += take;
            pt.ran(0, idle);
        }
        if (!__any(w.active)) {
            if (exhausted) break; // every wave gets here: the queue word only grows
            continue;
        }
        w.phases(S, io, st, pt, tl);
    }
    flush_counters<COUNT>(ctr, pt, tl, 0);
}

} // namespace rtd
