// Wavefront form of the path tracer: the same per-path arithmetic as the
// megakernel (kernels.h), split so that rays, not paths, are the unit of work.
//
//   k_wf_logic  one lane = one path slot of a fixed pool.  Adds the pending
//               next-event contribution, shades the slot's new hit
//               (src/main.cpp:173-264), emits a shadow ray and/or the next
//               extension ray into the ray queue, and refills finished slots
//               with fresh camera paths (src/main.cpp:294-323).
//   k_wf_trace  persistent kernel over the ray queue.  A lane that finishes
//               its ray takes the next one at once, so a wave stays full
//               although rays of one wave need very different numbers of BVH
//               steps - the loss that dominates incoherent bounces when a
//               whole path is tied to a lane.
//
// The host alternates the two until the item range is exhausted and no slot is
// alive.  Per-path random streams, draw order and every floating-point
// expression are those of path_step(); results are bit-identical to it.
#pragma once
#include "kernels.h"

namespace rtd {

enum : uint32_t { WF_EMPTY = 0, WF_ALIVE = 1, WF_DYING = 2, WF_PROBE = 3 };
constexpr uint32_t kShadowBit = 0x80000000u;

// structure of arrays over the pool's slots; every access is a 16-byte (or 8-byte) lane-contiguous one
struct WfPool {
    float4 *ray_o;   // extension ray origin
    float4 *ray_d;   // extension ray direction
    float4 *thr;     // throughput xyz, w = probability of the last bounce
    float4 *res;     // accumulated radiance xyz
    uint2 *rng;      // PCG state (the increment follows from the sample index)
    uint4 *meta;     // x = item, y = bounce | lobe << 16 | status << 20 | has_contrib << 24, z = pixel, w = sample k
    float4 *hit;     // closest-hit result: t, node, face | front << 31, -
    float2 *hit_uv;  // barycentrics of a triangle hit
    float4 *sh_o;    // shadow ray origin xyz, w = t_max
    float4 *sh_d;    // shadow ray direction xyz
    float4 *contrib; // next-event contribution xyz waiting for its shadow ray
    uint32_t *vis;   // 1 = shadow ray unoccluded
    uint32_t n_slots;
};

struct WfQueue {
    uint32_t *rays;              // slot | kShadowBit
    uint32_t *count;             // entries written by k_wf_logic
    unsigned long long *head;    // next entry k_wf_trace hands out
    unsigned long long *item;    // next work item of the launch
    uint32_t *alive;             // slots still holding a path after k_wf_logic
};

RT_DEV uint32_t pack_meta(int bounce, int lobe, uint32_t status, bool has_contrib)
{
    return ((uint32_t)bounce & 0xFFFFu) | ((uint32_t)lobe << 16) | (status << 20) | ((has_contrib ? 1u : 0u) << 24);
}

// append n entries for this wave with one atomic
RT_DEV uint32_t wave_reserve(uint32_t *counter, bool want, uint32_t &rank)
{
    const uint64_t m = __ballot(want);
    const uint32_t lane = threadIdx.x & 63u;
    rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    uint32_t base = 0;
    if (m != 0) {
        const int leader = __ffsll((unsigned long long)m) - 1;
        if ((int)lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
        base = __shfl(base, leader, 64);
    }
    return base;
}

__global__ __launch_bounds__(kBlock) void k_wf_logic(DScene S, DFrame F, const float *__restrict__ halton,
                                                     const int32_t *__restrict__ active, uint64_t n_items, int k0, int nk,
                                                     float4 *__restrict__ sbuf, FramePlanes out, WfPool P, WfQueue Q,
                                                     DCounters *ctr, int first_call)
{
    const uint32_t slot = blockIdx.x * kBlock + threadIdx.x;
    const bool in_pool = slot < P.n_slots;
    const rt_light &light = S.lights[0];
    uint32_t status = WF_EMPTY;
    uint4 meta = make_uint4(0, 0, 0, 0);
    PathState ps;
    path_begin(ps, Ray{mk(0, 0, 0), mk(0, 0, 1)});
    ps.rng.state = 0; ps.rng.inc = 1;
    bool want_ext = false, want_shadow = false;
    uint32_t new_paths = 0;

    if (in_pool && !first_call) {
        meta = P.meta[slot];
        status = (meta.y >> 20) & 0xFu;
    }
    if (status != WF_EMPTY) {
        const float4 thr = P.thr[slot], res = P.res[slot];
        ps.throughput = mk(thr.x, thr.y, thr.z); ps.lastProb = thr.w;
        ps.result = mk(res.x, res.y, res.z);
        ps.bounce = (int)(meta.y & 0xFFFFu); ps.lastLobe = (int)((meta.y >> 16) & 0xFu);
        // 1. the previous vertex' next-event estimate, now that its shadow ray is back (main.cpp:221-248)
        if ((meta.y >> 24) & 1u) {
            if (P.vis[slot]) { const float4 c = P.contrib[slot]; ps.result = ps.result + mk(c.x, c.y, c.z); }
        }
        bool finished = false;
        if (status == WF_DYING) finished = true;
        else {
            // 2. the new closest hit
            const float4 ro = P.ray_o[slot], rd = P.ray_d[slot];
            const Ray ray{mk(ro.x, ro.y, ro.z), mk(rd.x, rd.y, rd.z)};
            const float4 hr = P.hit[slot];
            Hit h;
            h.z = hr.x; h.node = (int)__float_as_uint(hr.y);
            const uint32_t ff = __float_as_uint(hr.z);
            h.face = ff & 0x7FFFFFFFu; h.front = (int)(ff >> 31);
            const float2 uv = P.hit_uv[slot];
            h.u = uv.x; h.v = uv.y;
            const bool hit = h.node != -1;
            if (ps.bounce == 0 && meta.w == 0) { // depth / id planes: first sample's camera ray
                if (out.z) out.z[meta.z] = h.z;
                if (out.object_id) out.object_id[meta.z] = hit ? h.node : -1;
            }
            if (status == WF_PROBE) finished = true; // zero bounces: the camera ray was traced for the planes only
            else if (!hit) {
                ps.result = ps.result + ld3(S.background) * ps.throughput; // main.cpp:175-176
                finished = true;
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
                finished = true;
            } else {
                const uint2 rs = P.rng[slot];
                ps.rng.state = ((uint64_t)rs.y << 32) | rs.x;
                ps.rng.inc = (((uint64_t)meta.w + 1) << 1u) | 1u;
                finish_hit(S, ray, h, false);
                Surface s; // SamplerInfo::SetHit (renderer.h:145-153)
                const V3 normal = normalize(h.n);
                s.p = h.p; s.n = normal; s.v = -normalize(ray.d); s.front = h.front;
                const rt_material &mtl = S.materials[S.nodes[h.node].material];
                // next-event estimation (main.cpp:214-251): the shadow ray is always sent, the
                // contribution it gates is worked out now with this vertex' throughput
                Sample ne;
                light_generate(light, s.p, ps.rng, ne);
                const float sgn = h.front ? 1.0f : -1.0f;
                const V3 so = h.p + (normal * F.bias) * sgn;
                P.sh_o[slot] = make_float4(so.x, so.y, so.z, ne.dist - F.bias);
                P.sh_d[slot] = make_float4(ne.dir.x, ne.dir.y, ne.dir.z, 0.0f);
                want_shadow = true;
                bool has_contrib = false;
                const float cosS = std_max(0.0f, dot(normal, ne.dir));
                if (cosS > 0.0f && ne.prob > 0.0f) {
                    const float mp = mtl_pdf(mtl, s, ne.dir);
                    float weight = 1.0f;
                    if (mp > 0.0f) weight = (ne.prob * ne.prob) / (ne.prob * ne.prob + mp * mp);
                    const V3 brdf = ld3(mtl.diffuse) / kPi;
                    const V3 c = ((((brdf * cosS) * ne.mult) * weight) / ne.prob) * ps.throughput;
                    P.contrib[slot] = make_float4(c.x, c.y, c.z, 0.0f);
                    has_contrib = true;
                }
                // indirect bounce (main.cpp:253-264)
                Sample in;
                bool go_on = mtl_generate(mtl, s, ps.rng, in);
                if (go_on) {
                    ps.lastProb = in.prob;
                    ps.lastLobe = in.lobe;
                    const float sg = (dot(normal, in.dir) > 0.0f) ? 1.0f : -1.0f;
                    const V3 no = h.p + (normal * F.bias) * sg;
                    ps.throughput = ps.throughput * (in.mult / in.prob);
                    ++ps.bounce;
                    if (ps.bounce >= F.max_bounces) go_on = false; // the bounce loop would end here
                    else {
                        P.ray_o[slot] = make_float4(no.x, no.y, no.z, 0.0f);
                        P.ray_d[slot] = make_float4(in.dir.x, in.dir.y, in.dir.z, 0.0f);
                        want_ext = true;
                    }
                }
                if (!go_on && !has_contrib) finished = true; // nothing left to wait for (the shadow ray is still traced and counted)
                else {
                    status = go_on ? WF_ALIVE : WF_DYING; // a dying path waits for its last shadow ray
                    meta.y = pack_meta(ps.bounce, ps.lastLobe, status, has_contrib);
                    P.meta[slot] = meta;
                    P.thr[slot] = make_float4(ps.throughput.x, ps.throughput.y, ps.throughput.z, ps.lastProb);
                    P.res[slot] = make_float4(ps.result.x, ps.result.y, ps.result.z, 0.0f);
                    P.rng[slot] = make_uint2((uint32_t)ps.rng.state, (uint32_t)(ps.rng.state >> 32));
                }
            }
        }
        if (finished) {
            sbuf[meta.x] = make_float4(ps.result.x, ps.result.y, ps.result.z, 0.0f);
            status = WF_EMPTY;
        }
    }

    // 3. refill: finished and never-used slots take the next work items, consecutive within a wave
    {
        const bool need = in_pool && status == WF_EMPTY;
        const uint64_t m = __ballot(need);
        if (m != 0) {
            const uint32_t lane = threadIdx.x & 63u;
            const uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            const int leader = __ffsll((unsigned long long)m) - 1;
            unsigned long long base = 0;
            if ((int)lane == leader) base = atomicAdd(Q.item, (unsigned long long)__popcll(m));
            base = __shfl(base, leader, 64);
            const uint64_t item = base + rank;
            if (need && item < n_items) {
                const uint32_t a = (uint32_t)(item / (uint32_t)nk);
                const int k = k0 + (int)(item - (uint64_t)a * (uint32_t)nk);
                const int pix = active[a];
                const int i = pix % F.width, j = pix / F.width;
                const uint64_t seed = pixel_seed(F.seed, (uint64_t)pix);
                rng_set_sequence(ps.rng, 0, seed);
                float off[4];
                off[0] = rng_float(ps.rng); off[1] = rng_float(ps.rng); off[2] = rng_float(ps.rng); off[3] = rng_float(ps.rng);
                rng_set_sequence(ps.rng, (uint64_t)k + 1, seed);
                const Ray ray = camera_ray(F, halton, i, j, k, off);
                status = F.max_bounces > 0 ? WF_ALIVE : WF_PROBE;
                if (status == WF_PROBE && k != 0) {
                    sbuf[item] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); // no bounce, no plane to fill: done
                    status = WF_EMPTY;
                } else {
                    P.ray_o[slot] = make_float4(ray.p.x, ray.p.y, ray.p.z, 0.0f);
                    P.ray_d[slot] = make_float4(ray.d.x, ray.d.y, ray.d.z, 0.0f);
                    P.thr[slot] = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
                    P.res[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    P.rng[slot] = make_uint2((uint32_t)ps.rng.state, (uint32_t)(ps.rng.state >> 32));
                    want_ext = true;
                }
                meta = make_uint4((uint32_t)item, pack_meta(0, LOBE_NONE, status, false), (uint32_t)pix, (uint32_t)k);
                P.meta[slot] = meta;
                ++new_paths;
            } else if (need) {
                meta.y = pack_meta(0, 0, WF_EMPTY, false);
                P.meta[slot] = meta;
            }
        }
    }

    // 4. ray queue and liveness
    {
        uint32_t rank;
        // two reservations keep the extension rays of a wave together (the more coherent set)
        uint32_t base = wave_reserve(Q.count, want_ext, rank);
        if (want_ext) Q.rays[base + rank] = slot;
        base = wave_reserve(Q.count, want_shadow, rank);
        if (want_shadow) Q.rays[base + rank] = slot | kShadowBit;
        (void)wave_reserve(Q.alive, status != WF_EMPTY, rank);
    }
    // camera paths started here; closest / shadow rays are counted where they are traced
    const uint64_t p = wave_sum(new_paths);
    if ((threadIdx.x & 63) == 0 && p) atomicAdd(&ctr->paths, (unsigned long long)p);
}

// ---------------------------------------------------------------------------
// ray-queue traversal
// ---------------------------------------------------------------------------
#ifndef RT_WF_MIN_WAVES
#define RT_WF_MIN_WAVES 6
#endif
#ifndef RT_WF_BVH_MIN_LANES
#define RT_WF_BVH_MIN_LANES 40 // leave the BVH loop to refill / advance other lanes below this many walking lanes
#endif

template <bool COUNT, int LEVELS>
__global__ __launch_bounds__(kBlock, RT_WF_MIN_WAVES) void k_wf_trace(DScene S, WfPool P, WfQueue Q, DCounters *ctr)
{
    extern __shared__ uint32_t lds[];
    const LaneStack st = lane_stack(lds, S);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_rays = *Q.count;
    const float epsilon = 1e-6f;
    PathTally pt;
    Tally<COUNT> tl;

    uint64_t chunk_next = 0, chunk_end = 0;
    bool exhausted = false;

    // per-lane ray state
    bool active = false, in_bvh = false, any_hit = false;
    uint32_t slot = 0;
    int node_i = 0;
    float t_max = 0.0f;           // shadow rays
    RayLevels<LEVELS> lv;         // local rays of the hierarchy levels above the current node
    lv.init(Ray{mk(0, 0, 0), mk(0, 0, 1)});
    Ray wray = lv.r0;             // world ray
    // best hit so far (closest rays)
    float best_t = FLT_MAX; int best_node = -1, best_front = 1; uint32_t best_face = 0; float best_u = 0, best_v = 0;
    // mesh walk
    Ray lr = lv.r0; V3 inv = mk(0, 0, 0); bool exact = false;
    GF4 bvh = nullptr, tris = nullptr;
    int sp = 0; float t_limit = 0.0f;
    bool m_hit = false; uint32_t m_face = 0; float m_u = 0, m_v = 0, m_det = 0;

    for (;;) {
        // ---- refill idle lanes from the queue ----
        const uint64_t need = __ballot(!active);
        if (need != 0) {
            if (chunk_next >= chunk_end && !exhausted) {
                unsigned long long b = 0;
                if (lane == 0) b = atomicAdd(Q.head, (unsigned long long)kChunk);
                b = __shfl(b, 0, 64);
                chunk_next = b < n_rays ? b : n_rays;
                chunk_end = b + kChunk < n_rays ? b + kChunk : n_rays;
                if (chunk_next >= chunk_end) exhausted = true;
            }
            const uint64_t avail = chunk_end - chunk_next;
            const uint32_t want = (uint32_t)__popcll(need);
            const uint32_t take = avail < want ? (uint32_t)avail : want;
            const uint32_t rank = (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
            if (!active && rank < take) {
                const uint32_t e = Q.rays[chunk_next + rank];
                slot = e & ~kShadowBit;
                any_hit = (e & kShadowBit) != 0;
                float4 o, d;
                if (any_hit) { o = P.sh_o[slot]; d = P.sh_d[slot]; t_max = o.w; ++pt.shadow; }
                else { o = P.ray_o[slot]; d = P.ray_d[slot]; ++pt.closest; }
                wray.p = mk(o.x, o.y, o.z); wray.d = mk(d.x, d.y, d.z);
                node_i = 0; in_bvh = false; active = true;
                best_t = FLT_MAX; best_node = -1; best_front = 1; best_face = 0; best_u = 0; best_v = 0;
            }
            chunk_next += take;
        }
        if (!__any(active)) {
            if (exhausted) break; // every wave gets here: the queue head only grows
            continue;
        }

        // ---- scene-graph steps for lanes that are not inside a mesh (src/main.cpp:16-89) ----
        while (__any(active && !in_bvh)) {
            if (active && !in_bvh) {
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
                            if (occ) { P.vis[slot] = 0u; active = false; advance = false; }
                        } else {
                            float t = 0; int front = 1;
                            const bool ok = prim == RT_PRIM_SPHERE ? sphere_t(lr, t, front) : plane_t(lr, t, front);
                            if (ok && t < best_t) { best_t = t; best_node = node_i; best_front = front; best_face = 0; best_u = 0; best_v = 0; }
                        }
                    } else if (prim == RT_PRIM_MESH) {
                        const DMesh &M = S.meshes[nd.mesh];
                        bvh = (GF4)M.bvh; tris = (GF4)M.tris;
                        inv = mk(1.0f / lr.d.x, 1.0f / lr.d.y, 1.0f / lr.d.z);
                        exact = needs_exact_slabs(inv);
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
                } else {
                    // all nodes done: unoccluded shadow ray, or closest ray going on to the lights
                    if (any_hit) P.vis[slot] = 1u;
                    else {
                        for (int l = 0; l < S.n_lights; ++l) { // src/main.cpp:52-70, src/lights.h:129-156
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
                        P.hit[slot] = make_float4(best_t, __uint_as_float((uint32_t)best_node),
                                                  __uint_as_float(best_face | ((uint32_t)best_front << 31)), 0.0f);
                        P.hit_uv[slot] = make_float2(best_u, best_v);
                    }
                    active = false;
                }
            }
        }

        // ---- BVH steps (src/TriObj.cpp:30-122 / 149-226), one stack entry per lane and trip ----
        while ((uint32_t)__popcll(__ballot(in_bvh)) >= (uint32_t)RT_WF_BVH_MIN_LANES ||
               (__any(in_bvh) && exhausted && !__any(active && !in_bvh))) {
            if (in_bvh) {
                --sp;
                uint32_t data; float nodeT;
                st.pop(sp, data, nodeT);
                if (!(t_limit < nodeT)) {
                    if (data & kLeafBit) {
                        const uint32_t off = data & 0x0FFFFFFFu;
                        const uint32_t cnt = ((data >> 28) & 7u) + 1;
                        for (uint32_t i = 0; i < cnt; i += 2) {
                            const GF4 ra = &tris[(off + i) * 3];
                            const bool two = i + 1 < cnt;
                            const GF4 rb = two ? ra + 3 : ra;
                            const f4 a0 = ra[0], a1 = ra[1], a2 = ra[2], b0 = rb[0], b1 = rb[1], b2 = rb[2];
                            float t, u, v, det; uint32_t face;
                            tl.tris(1);
                            if (tri_test(a0, a1, a2, lr, t, u, v, det, face) && !(t <= epsilon || t > t_limit)) {
                                m_hit = true; m_det = det; m_u = u; m_v = v; m_face = face; t_limit = t;
                                if (any_hit) { sp = 0; break; }
                            }
                            tl.tris(two ? 1 : 0);
                            if (two && tri_test(b0, b1, b2, lr, t, u, v, det, face) && !(t <= epsilon || t > t_limit)) {
                                m_hit = true; m_det = det; m_u = u; m_v = v; m_face = face; t_limit = t;
                                if (any_hit) { sp = 0; break; }
                            }
                        }
                    } else {
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
                            const bool firstNear = d1 < d2; // nearer child on top; on a tie the second child first
                            st.push(sp, firstNear ? k2 : k1, firstNear ? d2 : d1); ++sp;
                            st.push(sp, firstNear ? k1 : k2, firstNear ? d1 : d2); ++sp;
                        } else if (h1 || h2) {
                            st.push(sp, h1 ? k1 : k2, h1 ? d1 : d2); ++sp;
                        }
                    }
                }
                if (sp == 0) { // mesh done: fold its result into the ray's
                    in_bvh = false;
                    if (any_hit) {
                        if (m_hit) { P.vis[slot] = 0u; active = false; }
                    } else if (m_hit && t_limit < best_t) {
                        best_t = t_limit; best_node = node_i; best_front = m_det > 0.0f; best_face = m_face; best_u = m_u; best_v = m_v;
                    }
                    ++node_i;
                }
            }
        }
    }
    flush_counters<COUNT>(ctr, pt, tl, 0);
}

} // namespace rtd
