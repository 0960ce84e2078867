// The rendering kernel: a persistent grid in which every lane is a small state
// machine that carries one camera path at a time through
//
//     IDLE --refill--> RAY(closest) --done--> HIT --shade--> RAY(shadow) --done--+
//       ^                  ^                   |                                 |
//       |                  +------- next bounce, if the path goes on ------------+
//       +---- path ended: radiance written to the sample buffer <----------------+
//
// (src/main.cpp:170-265 for the path, 294-323 for the camera ray, 16-125 for the two
// kinds of ray).  What a path computes, its random stream and the order it draws
// from it are exactly those of path_step(); only the schedule differs:
//
//   * BVH steps are the unit of SIMD work.  Closest-hit and any-hit walks share one
//     loop, and a lane whose ray ends goes straight on to its next ray (shadow ray
//     -> next bounce -> fresh path), so the lanes of a wave stay busy although rays
//     of one wave need anything from 3 to 60 node visits.  With one whole trace
//     call per lane in lock-step the wave waits for its longest walk twice per
//     path vertex; measured SIMD utilisation there was below 20 %.
//   * Shading and path start-up ("logic") are batched: lanes wait in HIT / IDLE
//     until enough of them do, then run that code together.
//
// Wave-uniform control flow only decides *when* a lane takes its next step, never
// what the step computes, so results do not depend on the thresholds below.
#pragma once
#include "kernels.h"

namespace rtd {

#ifndef RT_SM_MIN_WAVES
#define RT_SM_MIN_WAVES 4 // waves per SIMD the register allocator must leave room for
#endif
#ifndef RT_SM_BVH_MIN_LANES
#define RT_SM_BVH_MIN_LANES 32 // keep taking BVH steps while at least this many lanes are inside a mesh
#endif
#ifndef RT_SM_LEAF_MIN_LANES
#define RT_SM_LEAF_MIN_LANES 16 // run triangle tests once this many lanes are parked at a leaf
#endif
#ifndef RT_SM_LOGIC_MIN_LANES
#define RT_SM_LOGIC_MIN_LANES 16 // run shading / refill once this many lanes wait for it (or nothing else can run)
#endif

enum : int { L_IDLE = 0, L_RAY = 1, L_HIT = 2, L_DEAD = 3 };

template <bool COUNT, int LEVELS>
__global__ __launch_bounds__(kBlock, RT_SM_MIN_WAVES) void k_path_sm(DScene S, DFrame F, const float *__restrict__ halton,
                                                                    const int32_t *__restrict__ active, uint64_t n_items, int k0,
                                                                    int nk, float4 *__restrict__ sbuf, FramePlanes out,
                                                                    DCounters *ctr, unsigned long long *queue)
{
    extern __shared__ uint32_t lds[];
    const LaneStack st = lane_stack(lds, S);
    const uint32_t lane = threadIdx.x & 63u;
    const float epsilon = 1e-6f;
    const rt_light &light = S.lights[0];
    const bool probe_only = F.max_bounces <= 0; // no bounce loop: camera rays are traced for the depth / id planes alone
    PathTally pt;
    Tally<COUNT> tl;
    uint32_t paths = 0;

    WorkQueue wq(queue, n_items); // wave-uniform

    // ---- per-lane state ----
    int state = L_IDLE;
    uint64_t item = 0;
    int pix = 0, k = 0;
    PathState ps; // ps.ray = the closest-hit ray in flight, or (while the shadow ray flies) the next one
    path_begin(ps, Ray{mk(0, 0, 0), mk(0, 0, 1)});
    ps.rng.state = 0; ps.rng.inc = 1;
    V3 contrib = mk(0, 0, 0);        // next-event estimate waiting for its shadow ray
    bool has_contrib = false, go_on = false;
    // ray in flight
    Ray wray = ps.ray;               // world space
    bool any_hit = false, in_bvh = false;
    float t_max = 0.0f;              // shadow rays
    int node_i = 0;                  // scene node being visited
    RayLevels<LEVELS> lv;            // local rays of the hierarchy levels above it
    lv.init(ps.ray);
    float best_t = FLT_MAX; int best_node = -1, best_front = 1; uint32_t best_face = 0; float best_u = 0, best_v = 0;
    // mesh walk
    Ray lr = ps.ray; V3 inv = mk(0, 0, 0); bool exact = false;
    GF4 bvh = nullptr, tris = nullptr;
    int sp = 0; float t_limit = 0.0f;
    bool m_hit = false; uint32_t m_face = 0; float m_u = 0, m_v = 0, m_det = 0;
    bool at_leaf = false; uint32_t leaf_data = 0, leaf_i = 0; // parked at a leaf, next triangle of it

    auto begin_ray = [&](Ray r, bool shadow, float tmax) {
        wray = r; any_hit = shadow; t_max = tmax;
        node_i = 0; in_bvh = false;
        best_t = FLT_MAX; best_node = -1; best_front = 1; best_face = 0; best_u = 0; best_v = 0;
        state = L_RAY;
    };
    auto end_path = [&]() {
        sbuf[item] = make_float4(ps.result.x, ps.result.y, ps.result.z, 0.0f);
        state = L_IDLE;
    };
    // the shadow ray of the current vertex is back (main.cpp:221-248), then main.cpp:253-264
    auto shadow_done = [&](bool visible) {
        if (visible && has_contrib) ps.result = ps.result + contrib;
        if (go_on) { ++pt.closest; begin_ray(ps.ray, false, 0.0f); }
        else end_path();
    };

    pt.tick();
    for (;;) {
        // ================= logic: shading of finished closest rays, then refill =================
        const uint64_t waiting = __ballot(state == L_HIT || state == L_IDLE);
        const bool flying = __any(state == L_RAY);
        if (waiting != 0 && ((uint32_t)__popcll(waiting) >= (uint32_t)RT_SM_LOGIC_MIN_LANES || !flying)) {
            pt.tock(12);
            if (state == L_HIT) {
                Hit h;
                h.z = best_t; h.node = best_node; h.front = best_front; h.face = best_face; h.u = best_u; h.v = best_v;
                const bool hit = h.node != -1;
                const Ray ray = ps.ray;
                if (ps.bounce == 0 && k == 0) { // depth / id planes: first sample's camera ray
                    if (out.z) out.z[pix] = h.z;
                    if (out.object_id) out.object_id[pix] = hit ? h.node : -1;
                }
                if (probe_only) end_path();
                else if (!hit) {
                    ps.result = ps.result + ld3(S.background) * ps.throughput; // main.cpp:175-176
                    end_path();
                } else if (h.node < 0) { // light (main.cpp:184-210)
                    const V3 rad = light_radiance(light);
                    if (ps.bounce == 0) ps.result = ps.result + rad * ps.throughput;
                    else {
                        float weight = 1.0f;
                        if (ps.lastLobe == LOBE_DIFFUSE) {
                            const float lp = light_pdf(light, ray.p, ray.d);
                            if (lp > 0.0f) weight = (ps.lastProb * ps.lastProb) / (ps.lastProb * ps.lastProb + lp * lp);
                        }
                        ps.result = ps.result + (rad * ps.throughput) * weight;
                    }
                    end_path();
                } else {
                    finish_hit(S, ray, h, false);
                    Surface s; // SamplerInfo::SetHit (renderer.h:145-153)
                    const V3 normal = normalize(h.n);
                    s.p = h.p; s.n = normal; s.v = -normalize(ray.d); s.front = h.front;
                    const rt_material &mtl = S.materials[S.nodes[h.node].material];
                    // next-event estimation (main.cpp:214-251): the shadow ray is always sent; what it
                    // gates is worked out now, while this vertex' throughput is at hand
                    Sample ne;
                    light_generate(light, s.p, ps.rng, ne);
                    const float sgn = h.front ? 1.0f : -1.0f;
                    Ray sr;
                    sr.p = h.p + (normal * F.bias) * sgn;
                    sr.d = ne.dir;
                    has_contrib = false;
                    const float cosS = std_max(0.0f, dot(normal, sr.d));
                    if (cosS > 0.0f && ne.prob > 0.0f) {
                        const float mp = mtl_pdf(mtl, s, ne.dir);
                        float weight = 1.0f;
                        if (mp > 0.0f) weight = (ne.prob * ne.prob) / (ne.prob * ne.prob + mp * mp);
                        // the specular term of main.cpp:241-246 is gated on a field that
                        // GetSampleInfo never writes (it reads as NONE): diffuse only
                        const V3 brdf = ld3(mtl.diffuse) / kPi;
                        contrib = ((((brdf * cosS) * ne.mult) * weight) / ne.prob) * ps.throughput;
                        has_contrib = true;
                    }
                    // indirect bounce (main.cpp:253-264)
                    Sample in;
                    go_on = mtl_generate(mtl, s, ps.rng, in);
                    if (go_on) {
                        ps.lastProb = in.prob;
                        ps.lastLobe = in.lobe;
                        const float sg = (dot(normal, in.dir) > 0.0f) ? 1.0f : -1.0f;
                        ps.ray.p = h.p + (normal * F.bias) * sg;
                        ps.ray.d = in.dir;
                        ps.throughput = ps.throughput * (in.mult / in.prob);
                        ++ps.bounce;
                        if (ps.bounce >= F.max_bounces) go_on = false; // the bounce loop ends here
                    }
                    ++pt.shadow;
                    begin_ray(sr, true, ne.dist - F.bias);
                }
            }
            // refill (main.cpp:294-323); items of a wave are consecutive
            const uint64_t need = __ballot(state == L_IDLE);
            if (need != 0) {
                wq.refill(lane);
                const uint64_t avail = wq.end - wq.next;
                const uint32_t want = (uint32_t)__popcll(need);
                const uint32_t take = avail < want ? (uint32_t)avail : want;
                const uint32_t rank = (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
                if (state == L_IDLE) {
                    if (rank < take) {
                        item = wq.next + rank;
                        const uint32_t a = (uint32_t)(item / (uint32_t)nk);
                        k = k0 + (int)(item - (uint64_t)a * (uint32_t)nk);
                        pix = active[a];
                        const int i = pix % F.width, j = pix / F.width;
                        const uint64_t seed = pixel_seed(F.seed, (uint64_t)pix);
                        rng_set_sequence(ps.rng, 0, seed);
                        float off[4];
                        off[0] = rng_float(ps.rng); off[1] = rng_float(ps.rng); off[2] = rng_float(ps.rng); off[3] = rng_float(ps.rng);
                        rng_set_sequence(ps.rng, (uint64_t)k + 1, seed);
                        path_begin(ps, camera_ray(F, halton, i, j, k, off));
                        ++paths;
                        if (probe_only && k != 0) end_path(); // nothing to trace, radiance 0; the lane takes another item next time
                        else {
                            if (!probe_only) ++pt.closest;
                            begin_ray(ps.ray, false, 0.0f);
                        }
                    } else if (wq.exhausted) state = L_DEAD;
                }
                wq.next += take;
            }
            pt.ran(0, waiting);
        }
        if (!__any(state == L_RAY)) {
            if (!__any(state == L_HIT || state == L_IDLE)) break; // every lane is DEAD: the queue only grows, so every wave gets here
            continue;
        }

        // ================= scene-graph steps of rays that are not inside a mesh (main.cpp:16-89) =================
        for (uint64_t m; (m = __ballot(state == L_RAY && !in_bvh)) != 0; pt.ran(1, m)) {
            pt.tock(12);
            if (state == L_RAY && !in_bvh) {
                if (node_i < S.n_nodes) {
                    const DNode &nd = S.nodes[node_i];
                    const int d = nd.depth;
                    const Ray parent = (d == 0) ? wray : lv.get(d - 1);
                    lr = to_node(nd, parent);
                    if (nd.has_children) lv.set(d, lr);
                    const int prim = nd.prim;
                    bool advance = true;
                    if (prim == RT_PRIM_SPHERE || prim == RT_PRIM_PLANE) {
                        if (any_hit) {
                            const bool occ = prim == RT_PRIM_SPHERE ? sphere_shadow(lr, t_max) : plane_shadow(lr, t_max);
                            if (occ) { shadow_done(false); advance = false; }
                        } else {
                            float t = 0; int front = 1;
                            const bool ok = prim == RT_PRIM_SPHERE ? sphere_t(lr, t, front) : plane_t(lr, t, front);
                            if (ok && t < best_t) { best_t = t; best_node = node_i; best_front = front; best_face = 0; best_u = 0; best_v = 0; }
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
                            st.push(0, __float_as_uint(r1.z), dist);
                            sp = 1; in_bvh = true; advance = false;
                        }
                    }
                    if (advance) ++node_i;
                } else if (any_hit) {
                    shadow_done(true); // nothing in the way
                } else {
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
                    state = L_HIT;
                }
            }
        }

        // ================= BVH steps (TriObj.cpp:30-122 / 149-226) =================
        // A walking lane pops one stack entry per trip and tests the two children of an inner
        // node.  A lane that pops a leaf parks there (at_leaf): triangle tests are long and only
        // a few lanes reach a leaf in any one trip, so they are run for many lanes at once, when
        // enough lanes have parked or the walkers have become fewer than the parked lanes.
        while (__any(in_bvh)) {
            pt.tock(12);
            const uint64_t walkers = __ballot(in_bvh && !at_leaf);
            if (in_bvh && !at_leaf) {
                --sp;
                uint32_t data; float nodeT;
                st.pop(sp, data, nodeT);
                if (!(t_limit < nodeT)) { // else: entered farther than the best hit so far
                    if (data & kLeafBit) { at_leaf = true; leaf_data = data; leaf_i = 0; }
                    else {
                        const uint32_t c1 = data & 0x7FFFFFFFu;
                        const f4 a0 = bvh[c1 * 2], a1 = bvh[c1 * 2 + 1], b0 = bvh[c1 * 2 + 2], b1 = bvh[c1 * 2 + 3];
                        float d1, d2;
                        tl.boxes(2);
                        bool h1 = bvh_box(exact, lr.p, inv, t_limit, a0, a1, d1);
                        h1 = h1 && d1 < t_limit;
                        bool h2 = bvh_box(exact, lr.p, inv, t_limit, b0, b1, d2);
                        h2 = h2 && d2 < t_limit;
                        const uint32_t k1 = __float_as_uint(a1.z), k2 = __float_as_uint(b1.z);
                        if (h1 && h2) {
                            const bool firstNear = d1 < d2; // nearer child on top; on a tie the second child is visited first
                            st.push(sp, firstNear ? k2 : k1, firstNear ? d2 : d1); ++sp;
                            st.push(sp, firstNear ? k1 : k2, firstNear ? d1 : d2); ++sp;
                        } else if (h1 || h2) {
                            st.push(sp, h1 ? k1 : k2, h1 ? d1 : d2); ++sp;
                        }
                    }
                }
            }
            pt.ran(2, walkers);
            const uint32_t n_leaf = (uint32_t)__popcll(__ballot(at_leaf));
            const uint32_t n_walk = (uint32_t)__popcll(__ballot(in_bvh && !at_leaf));
            if (n_leaf != 0 && (n_leaf >= (uint32_t)RT_SM_LEAF_MIN_LANES || n_leaf >= n_walk)) {
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
                    if (stop) { sp = 0; at_leaf = false; }
                    else if (leaf_i >= cnt) at_leaf = false;
                }
                pt.ran(3, parked);
            }
            if (in_bvh && !at_leaf && sp == 0) { // mesh done: fold its result into the ray's (main.cpp:38-48 / 80-85)
                in_bvh = false;
                if (any_hit) {
                    if (m_hit) shadow_done(false);
                    else ++node_i;
                } else {
                    if (m_hit && t_limit < best_t) {
                        best_t = t_limit; best_node = node_i; best_front = m_det > 0.0f; best_face = m_face; best_u = m_u; best_v = m_v;
                    }
                    ++node_i;
                }
            }
            if ((uint32_t)__popcll(__ballot(in_bvh)) < (uint32_t)RT_SM_BVH_MIN_LANES) break;
        }
    }
    flush_counters<COUNT>(ctr, pt, tl, paths);
}

} // namespace rtd
