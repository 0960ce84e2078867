// k_rays: closest-hit / any-hit queries for a list of rays, as a persistent kernel
// with ray-level regeneration: a lane that finishes its ray takes the next one from
// the list at once, so the lanes of a wave stay busy although rays need anything from
// 3 to 60 BVH node visits (src/main.cpp:16-125, src/TriObj.cpp:12-229).  The lane
// carries its ray (world space, and the space of the mesh it is in), its walk and the best
// hit so far: about forty registers of state, four waves per SIMD.
//
// Where rays come from and where answers go is a policy (IO): a plain list for the batch
// query entry points (RayListIO, BriefIO below), the path state of the wavefront renderer (wavefront.h).
//   bool fetch(e, split, o, d, any_hit, t_max, tag)   entry e of the list; false = no ray there.  tag is what
//                                              the lane keeps of the entry (whatever the other calls need)
//   void put_closest(tag, t, node, face, front, u, v)   t = FLT_MAX, node = -1: nothing hit
//   void put_shadow(tag, occluded)
//   kDeferShadow, Pending, begin_put_shadow(tag, pending), end_put_shadow(tag, pending): an IO that has to read
//   something before it can take the answer "not occluded" splits the hand-over into its loads and its stores; the
//   kernel issues the loads, then the loads of the lane's next ray, and only then asks for the stores
//   uint32_t setup(n, split)                   called once per thread before anything else: the length of the list (an IO
//                                              whose lists have their lengths in device memory reads them here); split
//                                              is a word the kernel keeps for the IO and hands to every fetch()
#pragma once
#include "kernels.h"

namespace rtd {

// ray_o = origin xyz, w = t_max for an any-hit (shadow) ray, negative for a closest-hit ray; ray_d = direction
// hit = closest: (t, node id bits, face | front << 31, u), hit_v = v;  any-hit: (1 or 0, 0, 0, 0)
struct NoPending {};
struct RayListIO {
    static constexpr bool kWide = false; // the reference's binary walk
    static constexpr bool kDeferShadow = false;
    typedef NoPending Pending;
    RT_DEV uint32_t setup(uint32_t n, uint32_t &) const { return n; }
    RT_DEV void begin_put_shadow(uint32_t, Pending &) const {}
    RT_DEV void end_put_shadow(uint32_t, const Pending &) const {}
    const float4 *ray_o, *ray_d;
    float4 *hit;
    float *hit_v;
    RT_DEV bool fetch(uint32_t e, uint32_t, float4 &o, float4 &d, bool &any_hit, float &t_max, uint32_t &tag) const
    {
        o = ray_o[e]; d = ray_d[e];
        any_hit = !(o.w < 0.0f); t_max = o.w; tag = e;
        return true;
    }
    RT_DEV void put_closest(uint32_t e, float t, int node, uint32_t face, int front, float u, float v) const
    {
        hit[e] = make_float4(t, __uint_as_float((uint32_t)node), __uint_as_float(face | ((uint32_t)front << 31)), u);
        hit_v[e] = v;
    }
    RT_DEV void put_shadow(uint32_t e, bool occluded) const { hit[e] = make_float4(occluded ? 1.0f : 0.0f, 0.0f, 0.0f, 0.0f); }
};

// The plain list through the four-wide walk, which the frames' extension and shadow rays take (WfIO, LaneIO): lets a
// test send known rays through that walk and compare with the reference's answers (RT_QUERY_WIDE=1 makes
// rt_trace_brief use it).  Same members as RayListIO, so the two are interchangeable as kernel arguments.
struct RayListWideIO : RayListIO {
    static constexpr bool kWide = true;
};

// The same list read and answered in the caller's own formats (six floats per ray, an optional t_max array,
// rt_hit_brief records): nothing to convert on the host, so an entry point built on it needs no staging buffers.
struct BriefIO {
    static constexpr bool kWide = false;
    static constexpr bool kDeferShadow = false;
    typedef NoPending Pending;
    RT_DEV uint32_t setup(uint32_t n, uint32_t &) const { return n; }
    RT_DEV void begin_put_shadow(uint32_t, Pending &) const {}
    RT_DEV void end_put_shadow(uint32_t, const Pending &) const {}
    const float *rays6, *t_max; // t_max null: closest hits
    rt_hit_brief *out;
    RT_DEV void refetch(uint32_t e, bool, float4 &o, float4 &d) const
    {
        const float *r = rays6 + (size_t)e * 6;
        o = make_float4(r[0], r[1], r[2], 0.0f); d = make_float4(r[3], r[4], r[5], 0.0f);
    }
    RT_DEV bool fetch(uint32_t e, uint32_t, float4 &o, float4 &d, bool &any_hit, float &tm, uint32_t &tag) const
    {
        any_hit = t_max != nullptr;
        tm = any_hit ? t_max[e] : -1.0f;
        refetch(e, any_hit, o, d);
        tag = e;
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
#define RT_RAYS_BVH_MIN_LANES 24 // keep taking BVH steps while at least this many lanes are inside a mesh
#endif
#ifndef RT_RAYS_LEAF_MIN_LANES
#define RT_RAYS_LEAF_MIN_LANES 16 // run triangle tests once this many lanes are parked at a leaf
#endif
#ifndef RT_RAYS_SCENE_MIN_LANES
#define RT_RAYS_SCENE_MIN_LANES 12 // run a pass over the scene nodes once this many lanes wait outside a mesh
#endif
#ifndef RT_RAYS_REFILL_MIN_LANES
#define RT_RAYS_REFILL_MIN_LANES 24 // fetch new rays once this many lanes are idle
#endif
constexpr uint32_t kRayChunk = 256; // rays a wave reserves per queue atomic unless the launch says otherwise
constexpr uint32_t kNoNode = 0x7FFFFFFFu; // not a data word: an inner node's first child index is far smaller

// One lane's ray and its walk, and the steps a wave takes with them; the members live in registers.
template <bool COUNT, int LEVELS, class IO> struct RayWalker {
    // The walk goes through the four-wide BVH (DScene::bvh4_pool): one 64-byte record gives the boxes of up to four
    // nodes two or three levels further down, the ones the ray enters are taken nearest first.  Half the dependent
    // fetches of a binary walk, the same triangles: a leaf is reached if and only if the ray enters every box above it,
    // and the record's 8-bit boxes contain the real ones.  What can differ from the reference's order is which of several triangles hit at exactly the
    // same distance is reported; t and the node never do.  The counting variant walks the binary tree, as the reference
    // does: its box and triangle counts are the reference's.  So do the batch query entry points, whose answers are the
    // reference's to the last bit: with two hits an ulp apart the order of the walk can decide which box is still
    // entered "before the closest hit so far" (one camera ray in twelve million of the 1M-triangle scene came back
    // one ulp farther through the wide walk).
    static constexpr bool kWide = IO::kWide && !COUNT;
    bool active = false, any_hit = false, in_bvh = false;
    bool unoccluded = false;             // a finished shadow ray whose answer the IO has not been told yet (retire())
    bool finished = false;               // a closest-hit ray through with every object: the lights and the answer are due (finish_closest())
    int mesh_node = 0;                   // scene node of the mesh the ray is in
    uint32_t tag = 0;                    // the list entry as the IO names it
    Ray wray{mk(0, 0, 0), mk(0, 0, 1)};  // world space
    float t_max = 0.0f;
    int node_i = 0;                      // scene node being visited
    float best_t = FLT_MAX; int best_node = -1, best_front = 1; uint32_t best_face = 0; float best_u = 0, best_v = 0;
    Ray lr{mk(0, 0, 0), mk(0, 0, 1)}; V3 inv = mk(0, 0, 0); bool exact = false; // the ray in the space of the mesh it is in
    int sp = 0; float t_limit = 0.0f;
    bool m_hit = false; uint32_t m_face = 0; float m_u = 0, m_v = 0, m_det = 0;
    bool at_leaf = false; uint32_t leaf_data = 0, leaf_i = 0;
    uint32_t cur = kNoNode; // node to visit next (its data word), kNoNode: take one from the stack

    RT_DEV void start(uint32_t tg, float4 o, float4 d, bool shadow, float tmax, PathTally &pt)
    {
        tag = tg;
        wray.p = mk(o.x, o.y, o.z); wray.d = mk(d.x, d.y, d.z);
        any_hit = shadow; t_max = tmax;
        if (any_hit) ++pt.shadow; else ++pt.closest;
        node_i = 0; in_bvh = false; active = true; finished = false;
        best_t = FLT_MAX; best_node = -1; best_front = 1; best_face = 0; best_u = 0; best_v = 0;
    }
    // A shadow ray's answer is handed over when the lane takes its next ray (retire(), called by the refill step): what
    // the IO has to read for it then travels together with the next ray's data instead of holding the wave up here.
    RT_DEV void shadow_done(const IO &io, bool occluded)
    {
        if (IO::kDeferShadow) unoccluded = !occluded;
        else io.put_shadow(tag, occluded);
        active = false;
    }

    // A closest-hit ray that is through with the objects: the spherical lights last (main.cpp:52-70, lights.h:129-156),
    // front side only, then the answer.  Called by the refill step, where the lane takes its next ray.
    RT_DEV void finish_closest(const DScene &S, const IO &io)
    {
        for (int l = 0; l < S.n_lights; ++l) {
            const rt_light L = load_uniform(&S.lights[l]);
            if (L.type != RT_LIGHT_POINT) continue;
            Ray r;
            r.p = wray.p - ld3(L.position);
            r.d = wray.d;
            if (L.size != 1.0f) { r.p = r.p / L.size; r.d = r.d / L.size; } // x / 1 is x
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
        io.put_closest(tag, best_t, best_node, best_face, best_front, best_u, best_v);
        finished = false;
    }
    // A ray that enters a mesh does not leave the scene pass for it: it notes the entry, goes on through the remaining
    // objects (spheres and planes are a handful of instructions each, and a shadow ray one of them stops never walks
    // the mesh) and starts the walk when the pass is over, so that a ray of a scene with one mesh takes part in one
    // pass, not two.  The reference takes the objects in list order and keeps the first of two hits at exactly the same
    // distance; the mesh's hit is therefore folded in with its place in the list as the tie-break.  A second mesh met
    // while one is noted ends the lane's pass there.  (Not in the counting variant: a shadow ray stopped by a later
    // sphere would skip box and triangle tests the reference makes.)
    static constexpr bool kDeferMesh = !COUNT;

    // Scenes of many objects (DScene::tlas: sixteen and more) are not walked object by object: the wave moves one cursor
    // through the hierarchy over the objects' padded world boxes (depth-first records with skip links, so no stack),
    // enters a record if any lane's ray reaches its box before that lane's closest hit so far, and only the lanes that
    // reach a leaf's box take its object into account - the ray in the object's space rebuilt through the object's
    // ancestors, the same products in the same order as the pass over the flat list forms them.  Objects come up in the
    // hierarchy's order, not the list's: of two hits at exactly the same distance the object that is first in the list
    // wins, as in the reference (strict <, objects in list order, src/main.cpp:38-48).  node_i is the lane's place among
    // the records: a lane that comes back from a mesh (it had met a second one) goes on from there - on the way down
    // to its place it follows the records that contain it and looks at nothing else.  (Not the counting variant: the
    // reference tests every object, and the counters are the reference's.)
    // The two passes are separate instantiations (LEVELS == 0: the hierarchy): both in one kernel cost the walker the
    // registers it has to spare (spills, 12 % on the 1M-triangle frame, which has five objects and never takes it).
    static constexpr bool kObjectHierarchy = LEVELS == 0;
    static_assert(!(COUNT && LEVELS == 0), "the counting variant walks the flat list");
    RT_DEV void scene_pass_hierarchy(const DScene &S, const IO &io, Tally<COUNT> &tl)
    {
        const bool mine = active && !in_bvh;
        const V3 winv = cull_inverse(wray.d);
        bool noted = false, stopped = false; // a mesh entry waits for the end of the pass; the lane has met a second mesh
        int cursor = 0;
        while (cursor < S.n_tlas) { // every lane of the wave stays in this loop to its end: the votes need them all
            const int c = __builtin_amdgcn_readfirstlane(cursor);
            const TlasNode tn = load_uniform(&S.tlas[c]);
            const int skip = tn.skip, obj = tn.object;
            const bool here = mine && active && !in_bvh && !stopped;
            const bool behind = c < node_i; // the lane has been past this record already
            const bool reach = here && (behind ? skip > node_i : reaches_box(tn.box, wray.p, winv, any_hit ? t_max : best_t));
            if (!__any(reach)) { cursor = skip; continue; }
            cursor = c + 1;
            if (obj < 0 || !reach || behind) continue;
            const DNode &nd = S.nodes[obj];
            Ray l = wray;
            for (int k = 0; k <= nd.depth; ++k) l = to_node(S.nodes[nd.chain[k]], l);
            const int prim = nd.prim;
            if (prim == RT_PRIM_SPHERE || prim == RT_PRIM_PLANE) {
                if (any_hit) {
                    const bool occ = prim == RT_PRIM_SPHERE ? sphere_shadow(l, t_max) : plane_shadow(l, t_max);
                    if (occ) { shadow_done(io, true); noted = false; }
                } else {
                    float t = 0; int front = 1;
                    const bool ok = prim == RT_PRIM_SPHERE ? sphere_t(l, t, front) : plane_t(l, t, front);
                    if (ok && (t < best_t || (t == best_t && obj < best_node))) { best_t = t; best_node = obj; best_front = front; best_face = 0; best_u = 0; best_v = 0; }
                }
            } else if (prim == RT_PRIM_MESH) {
                const TNode tm = load_uniform(&S.tnodes[obj]);
                const V3 iv = mk(1.0f / l.d.x, 1.0f / l.d.y, 1.0f / l.d.z);
                const bool ex = needs_exact_slabs(l.p, iv);
                const float lim = any_hit ? t_max : FLT_MAX;
                const f4 r0 = {tm.box[0], tm.box[1], tm.box[2], tm.box[3]}, r1 = {tm.box[4], tm.box[5], 0.0f, 0.0f};
                float dist;
                tl.boxes(1);
                if (!bvh_box(ex, l.p, iv, lim, r0, r1, dist)) {}
                else if (noted) { stopped = true; node_i = c; } // one walk at a time: come back to this record after it
                else {
                    lr = l; inv = iv; exact = ex; t_limit = lim; m_hit = false;
                    cur = kWide ? tm.root4 : tm.root; // the root is the node to visit next; the stack is empty
                    sp = 0; mesh_node = obj;
                    noted = true;
                }
            }
        }
        if (mine && active && !stopped) node_i = S.n_tlas; // through with the objects
        if (mine && active && noted) in_bvh = true; // the walk of the mesh noted on the way
        else if (mine && active && !in_bvh) { // nothing left to look at
            if (any_hit) shadow_done(io, false);
            else { finished = true; active = false; }
        }
    }

    // The scene pass, if enough lanes wait for it, then BVH steps while enough lanes are inside a mesh.
    // Called by the whole wave; every call makes progress if any lane has a ray.
    RT_DEV void phases(const DScene &S, const IO &io, const LaneStack &st, PathTally &pt, Tally<COUNT> &tl)
    {
        const float epsilon = 1e-6f;
        const GF4 bvh = (GF4)S.bvh_pool, tris = (GF4)(kWide ? S.tri4_pool : S.tri_pool);
    // ================= scene-graph steps of rays that are not inside a mesh (main.cpp:16-89) =================
    // One pass over the node list in step: the node index is wave-uniform, so a node's record arrives through one
    // scalar fetch; a lane takes part from the node it has got to (the local rays of the hierarchy levels above it
    // are rebuilt on the way there), and stays in the pass until it enters a mesh or runs off the end of the list.
    const uint64_t stepping = __ballot(active && !in_bvh);
    const bool pass_now = stepping != 0 && ((uint32_t)__popcll(stepping) >= (uint32_t)RT_RAYS_SCENE_MIN_LANES || !__any(in_bvh));
    if constexpr (kObjectHierarchy) {
        if (pass_now) {
            pt.tock(12);
            scene_pass_hierarchy(S, io, tl);
            pt.ran(1, stepping);
        }
    } else if (pass_now) {
        pt.tock(12);
        const bool mine = active && !in_bvh;
        const int first = node_i;
        RayLevels<LEVELS> lv;
        lv.init(wray);
        // An object whose padded world-space box the ray does not reach before its closest hit so far (t is the same
        // in every node's space) cannot change the answer: the lane skips it, and a node no lane of the wave has
        // business with costs the wave nothing but this test.  The reference tests every object (it keeps the boxes,
        // src/scene.h:434-444, but never looks at them).
        const V3 winv = cull_inverse(wray.d);
        bool noted = false;   // a mesh entry waits for the end of the pass
        bool stopped = false; // ... and the lane has met a second mesh
        for (int n = 0; n < S.n_nodes; ++n) {
            const TNode nd = load_uniform(&S.tnodes[n]);
            const uint32_t meta = nd.meta;
            const int prim = (int)(meta & 0xFFu), d = (int)((meta >> 8) & 0xFFu);
            const bool has_children = (meta >> 16) != 0;
            bool here = mine && active && !in_bvh && !stopped; // still in the pass
            // (the counting variant takes every object, as the reference does: its box and triangle tests are the reference's)
            const bool wanted = here && n >= first && prim != RT_PRIM_NONE && (COUNT || reaches_box(nd.wbox, wray.p, winv, any_hit ? t_max : best_t));
            if (here && n >= first && !wanted) node_i = n + 1; // nothing to do at this node
            if (!has_children) here = wanted;
            if (!__any(here)) continue;
            if (here) {
                const Ray parent = (d == 0) ? wray : lv.get(d - 1);
                Ray l;
                l.p = m34_mul(nd.itm, parent.p);
                l.d = m3_mul(nd.itm, parent.d);
                if (has_children) lv.set(d, l);
                if (wanted) {
                    bool advance = true;
                    if (prim == RT_PRIM_SPHERE || prim == RT_PRIM_PLANE) {
                        if (any_hit) {
                            const bool occ = prim == RT_PRIM_SPHERE ? sphere_shadow(l, t_max) : plane_shadow(l, t_max);
                            if (occ) { shadow_done(io, true); advance = false; noted = false; }
                        } else {
                            float t = 0; int front = 1;
                            const bool ok = prim == RT_PRIM_SPHERE ? sphere_t(l, t, front) : plane_t(l, t, front);
                            if (ok && t < best_t) { best_t = t; best_node = n; best_front = front; best_face = 0; best_u = 0; best_v = 0; }
                        }
                    } else if (prim == RT_PRIM_MESH) {
                        const V3 iv = mk(1.0f / l.d.x, 1.0f / l.d.y, 1.0f / l.d.z);
                        const bool ex = needs_exact_slabs(l.p, iv);
                        const float lim = any_hit ? t_max : FLT_MAX;
                        const f4 r0 = {nd.box[0], nd.box[1], nd.box[2], nd.box[3]}, r1 = {nd.box[4], nd.box[5], 0.0f, 0.0f};
                        float dist;
                        tl.boxes(1);
                        if (!bvh_box(ex, l.p, iv, lim, r0, r1, dist)) {}
                        else if (noted) { stopped = true; advance = false; } // one walk at a time: come back to this node after it
                        else {
                            lr = l; inv = iv; exact = ex; t_limit = lim; m_hit = false;
                            cur = kWide ? nd.root4 : nd.root; // the root is the node to visit next; the stack is empty
                            sp = 0; mesh_node = n;
                            if (kDeferMesh) noted = true;
                            else { in_bvh = true; advance = false; }
                        }
                    }
                    if (advance) node_i = n + 1;
                }
            }
        }
        if (mine && active && noted) in_bvh = true; // the walk of the mesh noted on the way
        else if (mine && active && !in_bvh) { // ran off the end of the node list
            if (any_hit) shadow_done(io, false); // nothing in the way
            else { finished = true; active = false; }
        }
        pt.ran(1, stepping);
    }

    // ================= BVH steps (TriObj.cpp:30-122 / 149-226) =================
    // The node to visit next is kept in a register (cur): of two children that are hit only the
    // farther one goes to the stack, the nearer one is visited in the next trip.  In the binary
    // walk that is exactly the order of the reference, which pushes both and pops the nearer one
    // at once (its entry distance was just tested against the current closest hit, so the
    // pop-time test of the reference cannot fail for it and is not repeated).  The four-wide walk
    // (kWide) visits the children of two levels in distance order instead: same closest hit, but
    // pruned against a different "best so far", so t can differ in the last bit now and then -
    // which is why the depth / id planes and the query entry points use the binary walk.  The
    // stack is read only when a visit leaves nothing to go on with.
    // A lane whose next node is a leaf parks there: triangle tests are long and only a few
    // lanes reach a leaf in any one trip, so they run for many lanes at once.
    while (__any(in_bvh)) {
        pt.tock(12);
        const uint64_t walkers = __ballot(in_bvh && !at_leaf);
        if (in_bvh && !at_leaf) {
            if (cur == kNoNode && sp > 0) {
                --sp;
                uint32_t data; float nodeT;
                st.pop_named(sp, data, nodeT);
                if (!(t_limit < nodeT)) cur = data; // else: entered farther than the best hit so far
            }
            if (cur != kNoNode) {
                if (cur & kLeafBit) { at_leaf = true; leaf_data = cur; leaf_i = 0; cur = kNoNode; }
                else if (kWide) {
                    // the record (device/lbvh.h: wide_record_pack): three and a half 16-byte loads
                    const GF4 rec = (GF4)S.bvh4_pool + (size_t)cur * 4;
                    const f4 r0 = rec[0], r1 = rec[1], r2 = rec[2];
                    const f2 r3 = *(const RT_GLOBAL_AS f2 *)(rec + 3);
                    const uint32_t eb = __float_as_uint(r0.w);
                    const V3 step = mk(__uint_as_float((eb & 0xFFu) << 23), __uint_as_float(((eb >> 8) & 0xFFu) << 23), __uint_as_float(((eb >> 16) & 0xFFu) << 23));
                    const uint32_t lox = __float_as_uint(r2.x), hix = __float_as_uint(r2.y), loy = __float_as_uint(r2.z), hiy = __float_as_uint(r2.w);
                    const uint32_t loz = __float_as_uint(r3.x), hiz = __float_as_uint(r3.y);
                    uint32_t k0 = __float_as_uint(r1.x), k1 = __float_as_uint(r1.y), k2 = __float_as_uint(r1.z), k3 = __float_as_uint(r1.w);
                    float d0, d1, d2, d3;
                    bool h0, h1, h2, h3;
                    if (__builtin_expect(exact, 0)) { // (one branch around the four tests, not one in each) the boxes as floats, then the reference's test
                        const auto child = [&](int k, float &dist) {
                            const auto q = [k](uint32_t plane) { return (float)((plane >> (8 * k)) & 0xFFu); };
                            const f4 b0 = {r0.x + q(lox) * step.x, r0.x + q(hix) * step.x, r0.y + q(loy) * step.y, r0.y + q(hiy) * step.y};
                            const f4 b1 = {r0.z + q(loz) * step.z, r0.z + q(hiz) * step.z, 0.0f, 0.0f};
                            return bvh_box_exact(lr.p, inv, t_limit, b0, b1, dist);
                        };
                        h0 = child(0, d0); h1 = child(1, d1); h2 = child(2, d2); h3 = child(3, d3);
                    } else {
                        // t at a child's plane q: ((corner + q step) - p) inv = (corner - p) inv + q (step inv)
                        const V3 at = mk((r0.x - lr.p.x) * inv.x, (r0.y - lr.p.y) * inv.y, (r0.z - lr.p.z) * inv.z);
                        const V3 per = mk(step.x * inv.x, step.y * inv.y, step.z * inv.z);
                        h0 = wide_box_fast<0>(at, per, t_limit, lox, hix, loy, hiy, loz, hiz, d0);
                        h1 = wide_box_fast<1>(at, per, t_limit, lox, hix, loy, hiy, loz, hiz, d1);
                        h2 = wide_box_fast<2>(at, per, t_limit, lox, hix, loy, hiy, loz, hiz, d2);
                        h3 = wide_box_fast<3>(at, per, t_limit, lox, hix, loy, hiy, loz, hiz, d3);
                    }
                    // (a box is entered before its far side, and the far side is cut at t_limit: no further test against
                    // t_limit) boxes the ray does not enter, and slots without a child, sort to the end
                    d0 = h0 && k0 != kNoNode ? d0 : INFINITY; d1 = h1 && k1 != kNoNode ? d1 : INFINITY;
                    d2 = h2 && k2 != kNoNode ? d2 : INFINITY; d3 = h3 && k3 != kNoNode ? d3 : INFINITY;
                    const auto order = [](float &da, uint32_t &ka, float &db, uint32_t &kb) {
                        const bool sw = db < da;
                        const float td = sw ? db : da; const uint32_t tk = sw ? kb : ka;
                        db = sw ? da : db; kb = sw ? ka : kb;
                        da = td; ka = tk;
                    };
                    order(d0, k0, d1, k1); order(d2, k2, d3, k3); order(d0, k0, d2, k2); order(d1, k1, d3, k3); order(d1, k1, d2, k2);
                    // farthest first onto the stack, the nearest is visited next.  While three more entries fit the LDS
                    // levels all three are written, without a branch each, and the stack pointer moves past those that
                    // count (the sort put them first, so what an entry that does not count leaves behind is written
                    // over by the next one or lies above the top).
                    if (__builtin_expect((uint32_t)sp + 3u <= st.lds_levels, 1)) {
                        st.put_lds(sp, k3, d3); sp += d3 < INFINITY ? 1 : 0;
                        st.put_lds(sp, k2, d2); sp += d2 < INFINITY ? 1 : 0;
                        st.put_lds(sp, k1, d1); sp += d1 < INFINITY ? 1 : 0;
                    } else {
                        if (d3 < INFINITY) { st.push_named(sp, k3, d3); ++sp; }
                        if (d2 < INFINITY) { st.push_named(sp, k2, d2); ++sp; }
                        if (d1 < INFINITY) { st.push_named(sp, k1, d1); ++sp; }
                    }
                    cur = d0 < INFINITY ? k0 : kNoNode;
                } else {
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
                    if (h1 && h2) { st.push_named(sp, firstNear ? k2 : k1, firstNear ? d2 : d1); ++sp; }
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
            if (!kDeferMesh) node_i = mesh_node + 1; // (deferred entry: the pass has moved node_i on already)
            const bool through = node_i >= (kObjectHierarchy ? S.n_tlas : S.n_nodes);
            if (any_hit) {
                if (m_hit) shadow_done(io, true);
                else if (through) shadow_done(io, false);
            } else {
                // the nearer hit wins, of two equally near ones the object that comes first in the list
                if (m_hit && (t_limit < best_t || (kDeferMesh && t_limit == best_t && best_node > mesh_node))) {
                    best_t = t_limit; best_node = mesh_node; best_front = m_det > 0.0f; best_face = m_face; best_u = m_u; best_v = m_v;
                }
                if (through) { finished = true; active = false; }
            }
        }
        if ((uint32_t)__popcll(__ballot(in_bvh)) < (uint32_t)RT_RAYS_BVH_MIN_LANES) break;
    }
    }
};

// The list is handed out in chunks of ray_chunk entries, one atomic per chunk and wave.  (Cutting the list into
// eight parts with a queue word each, so that each XCD's L2 sees the rays of one eighth of the image, was measured:
// 88.6 against 89.0 ms of ray launches per 1080p frame, not worth the index arithmetic in the refill step.)
#ifndef RT_PORTION_DIV
#define RT_PORTION_DIV 8
#endif
#ifndef RT_PORTION_MAX
#define RT_PORTION_MAX 1024
#endif
// rays a wave reserves per queue atomic: few atomics on the queue word in large launches, enough portions for every
// wave (and an even end) in small ones - about an eighth of a wave's share of the list
RT_HD uint32_t ray_portion(uint64_t n_rays, unsigned grid_blocks)
{
    const uint64_t per_wave = n_rays / ((uint64_t)grid_blocks * (kBlock / 64)) / RT_PORTION_DIV;
    return per_wave >= RT_PORTION_MAX ? (uint32_t)RT_PORTION_MAX : per_wave >= 4096 ? 4096u : per_wave >= 2048 ? 2048u : per_wave >= 1024 ? 1024u : per_wave >= 512 ? 512u : per_wave >= 256 ? 256u : per_wave >= 128 ? 128u : 64u;
}

// ray_chunk 0: sized here from the length of the list (ray_portion)
// lane 0's value in every lane, as a scalar (the whole wave is active where this is called)
RT_DEV uint32_t first_lane(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
RT_DEV unsigned long long first_lane(unsigned long long v)
{
    return ((unsigned long long)first_lane((uint32_t)(v >> 32)) << 32) | first_lane((uint32_t)v);
}

template <bool COUNT, int LEVELS, class IO>
__global__ __launch_bounds__(kBlock, RT_RAYS_MIN_WAVES) void k_rays(DScene S, IO io, uint32_t n_rays_arg, DCounters *ctr,
                                                                  unsigned long long *queue, uint32_t ray_chunk)
{
    uint32_t list_split = 0;
    const uint32_t n_rays = io.setup(n_rays_arg, list_split);
    if (ray_chunk == 0) ray_chunk = ray_portion(n_rays, gridDim.x);
#ifdef RT_PROFILE_TAIL
    // diagnostic build: when do the waves of one launch (the one launched with n_rays_arg == RT_PROFILE_TAIL) start,
    // run out of list, and end?  phase[0] earliest start, [1] latest end, [2] sum of starts, [3] sum of the moments the
    // list was found empty, [4] sum of ends, [5] waves, [6] latest "list empty", [7] earliest "list empty" (100 MHz clock)
    const unsigned long long tail_t0 = wall_clock64();
    unsigned long long tail_tx = 0;
#endif
    extern __shared__ uint32_t lds[];
    const LaneStack st = lane_stack(lds, S);
    const uint32_t lane = threadIdx.x & 63u;
    PathTally pt;
    Tally<COUNT> tl;
    RayWalker<COUNT, LEVELS, IO> w;
    // The list is handed out in portions of ray_chunk entries, one global atomic per portion; a portion belongs to the
    // WORKGROUP, whose waves draw their rays from it as their lanes fall idle (one LDS atomic per refill).  A portion
    // is a few neighbouring pixels' rays, and cost is a matter of place (rays along a mesh's rim take fifteen times
    // the average): with a portion per wave, the wave that drew a bad one late in the launch kept the chip waiting for
    // up to four rays per lane, one after the other; shared, it is one ray per lane, and the waves still work on
    // neighbouring rays.  s_win = end << 32 | next: entries [next, end) of the list are the workgroup's to take.
    __shared__ unsigned long long s_win;
    __shared__ uint32_t s_lock, s_done; // one wave at a time fetches the next portion; the list has run out
    if (threadIdx.x == 0) { s_win = 0; s_lock = 0; s_done = 0; }
    __syncthreads();
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
            const uint32_t want = (uint32_t)__popcll(idle);
            uint32_t first = 0, take = 0;
            {
                unsigned long long old = 0;
                if (lane == 0) old = atomicAdd(&s_win, (unsigned long long)want);
                old = first_lane(old);
                const uint32_t nx = (uint32_t)old, en = (uint32_t)(old >> 32);
                if (nx < en) { first = nx; take = en - nx < want ? en - nx : want; }
                else {
                    // the portion is used up: one wave fetches the next (the others come back at their next trip)
                    uint32_t got_lock = 0;
                    if (lane == 0) got_lock = (__hip_atomic_load(&s_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u && atomicCAS(&s_lock, 0u, 1u) == 0u) ? 1u : 0u;
                    got_lock = first_lane(got_lock);
                    if (got_lock) {
                        // (rt_cancel is not looked at here: a ray launch lasts milliseconds and the host checks between launches;
                        // reading the flag, which lives in host memory, at every draw cost the 1080p frame 13 ms)
                        unsigned long long b = 0;
                        if (lane == 0) {
                            // another wave may have fetched a portion between this wave's look at the window and its taking the lock
                            const unsigned long long now = __hip_atomic_load(&s_win, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            if ((uint32_t)now >= (uint32_t)(now >> 32)) {
                                b = atomicAdd(queue, (unsigned long long)ray_chunk);
                                if (b < n_rays) {
                                    const uint32_t e = b + ray_chunk < n_rays ? (uint32_t)(b + ray_chunk) : n_rays;
                                    // this wave's own share comes off the new portion before the others see it
                                    const uint32_t mine = e - (uint32_t)b < want ? e - (uint32_t)b : want;
                                    atomicExch(&s_win, ((unsigned long long)e << 32) | ((uint32_t)b + mine));
                                    b = ((unsigned long long)mine << 32) | (uint32_t)b;
                                } else {
                                    __hip_atomic_store(&s_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                    b = 0;
                                }
                            }
                            __hip_atomic_store(&s_lock, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                        b = first_lane(b);
                        first = (uint32_t)b; take = (uint32_t)(b >> 32);
                    }
                    if (take == 0) {
                        uint32_t done = 0;
                        if (lane == 0) done = __hip_atomic_load(&s_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (first_lane(done)) {
                            // the list has run out - unless a last portion came in just before: look once more
                            unsigned long long now = 0;
                            if (lane == 0) now = __hip_atomic_load(&s_win, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            now = first_lane(now);
                            if ((uint32_t)now >= (uint32_t)(now >> 32)) {
                                exhausted = true;
#ifdef RT_PROFILE_TAIL
                                if (tail_tx == 0) tail_tx = wall_clock64();
#endif
                            }
                        }
                    }
                }
            }
            const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
            const bool taking = !w.active && rank < take;
            typename IO::Pending pend; // what the lane's finished shadow ray still has to hand over: its loads go out first
            const bool owes = IO::kDeferShadow && !w.active && w.unoccluded;
            if (owes) io.begin_put_shadow(w.tag, pend);
            if (!w.active && w.finished) w.finish_closest(S, io);
            if (taking) {
                const uint32_t e = first + rank;
                float4 o, d; bool shadow; float tmax; uint32_t tg;
                const uint32_t old = w.tag;
                const bool got = io.fetch(e, list_split, o, d, shadow, tmax, tg); // a list may have holes
                if (owes) { io.end_put_shadow(old, pend); w.unoccluded = false; }
                if (got) w.start(tg, o, d, shadow, tmax, pt);
            } else if (owes) { io.end_put_shadow(w.tag, pend); w.unoccluded = false; }
            pt.ran(0, idle);
        }
        if (!__any(w.active)) {
            if (exhausted) break; // every wave gets here: the queue word only grows
            continue;
        }
        w.phases(S, io, st, pt, tl);
    }
    if (IO::kDeferShadow && w.unoccluded) { typename IO::Pending pend; io.begin_put_shadow(w.tag, pend); io.end_put_shadow(w.tag, pend); }
    if (w.finished) w.finish_closest(S, io);
#ifdef RT_PROFILE_TAIL
    if (lane == 0 && n_rays_arg == (uint32_t)(RT_PROFILE_TAIL)) {
        const unsigned long long t1 = wall_clock64();
        if (tail_tx == 0) tail_tx = t1;
        atomicMin(&ctr->phase[0], tail_t0); atomicMax(&ctr->phase[1], t1);
        atomicAdd(&ctr->phase[2], tail_t0); atomicAdd(&ctr->phase[3], tail_tx); atomicAdd(&ctr->phase[4], t1);
        atomicAdd(&ctr->phase[5], 1ull);
        atomicMax(&ctr->phase[6], tail_tx); atomicMin(&ctr->phase[7], tail_tx);
    }
#endif
    flush_counters<COUNT>(ctr, pt, tl, 0);
}

} // namespace rtd
