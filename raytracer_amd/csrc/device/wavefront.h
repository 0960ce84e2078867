// Wavefront form of the path tracer: the same per-path arithmetic as path_step()
// (shade.h), split so that rays, not paths, are the unit of SIMD work.
//
//   k_wf_logic  one lane = one path slot of a fixed pool.  Adds the pending
//               next-event contribution, shades the slot's new hit
//               (src/main.cpp:173-264), leaves a shadow ray and/or the next
//               extension ray in the slot, and refills finished slots with fresh
//               camera paths (src/main.cpp:294-323).
//   k_rays<PoolIO>  (rays.h) traces every ray the slots hold, with ray-level
//               regeneration: the list it walks is simply "extension ray of slot
//               0..n-1, then shadow ray of slot 0..n-1", entries without a ray are
//               skipped, so k_wf_logic needs no queue and no atomics to hand rays over.
//
// The host alternates the two until the item range is used up and no slot is
// alive.  Per-path random streams, draw order and every floating-point
// expression are those of path_step(); results are bit-identical to it.
#pragma once
#include "rays.h"

namespace rtd {

enum : uint32_t { WF_EMPTY = 0, WF_ALIVE = 1, WF_DYING = 2, WF_PROBE = 3 };
enum : uint32_t { WF_HAS_EXT = 1, WF_HAS_SHADOW = 2 };

// structure of arrays over the pool's slots; every access is a 16-byte (or 8/4-byte) lane-contiguous one
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
    uint32_t *flags; // rays the slot wants traced: WF_HAS_EXT | WF_HAS_SHADOW
    uint32_t n_slots;
};

struct WfQueue {
    unsigned long long *item;        // next work item nobody has reserved yet
    unsigned long long *block_items; // [grid][2]: what each workgroup still holds of its last reservation
    uint32_t *alive;                 // slots still holding a path after k_wf_logic
};

// ray list of k_rays over the pool: entry e < n_slots is the extension ray of slot e, the others the shadow rays
struct PoolIO {
    WfPool P;
    RT_DEV bool fetch(uint32_t e, float4 &o, float4 &d, bool &any_hit, float &t_max) const
    {
        any_hit = e >= P.n_slots;
        const uint32_t slot = any_hit ? e - P.n_slots : e;
        if (!(P.flags[slot] & (any_hit ? WF_HAS_SHADOW : WF_HAS_EXT))) return false;
        if (any_hit) { o = P.sh_o[slot]; d = P.sh_d[slot]; t_max = o.w; }
        else { o = P.ray_o[slot]; d = P.ray_d[slot]; t_max = 0.0f; }
        return true;
    }
    RT_DEV void put_closest(uint32_t e, float t, int node, uint32_t face, int front, float u, float v) const
    {
        P.hit[e] = make_float4(t, __uint_as_float((uint32_t)node), __uint_as_float(face | ((uint32_t)front << 31)), 0.0f);
        P.hit_uv[e] = make_float2(u, v);
    }
    RT_DEV void put_shadow(uint32_t e, bool occluded) const { P.vis[e - P.n_slots] = occluded ? 0u : 1u; }
};

RT_DEV uint32_t pack_meta(int bounce, int lobe, uint32_t status, bool has_contrib)
{
    return ((uint32_t)bounce & 0xFFFFu) | ((uint32_t)lobe << 16) | (status << 20) | ((has_contrib ? 1u : 0u) << 24);
}

// Steps 1 and 2 of a slot's turn: add the pending next-event estimate, shade the new closest hit
// (src/main.cpp:173-264) and leave the next rays in the slot.  Returns the slot's status afterwards
// (WF_EMPTY: the path, if there was one, has ended and its radiance is in the sample buffer).
RT_DEV uint32_t wf_shade_slot(const DScene &S, const DFrame &F, const WfPool &P, uint32_t slot, uint4 &meta, float4 *sbuf,
                              const FramePlanes &out, bool &want_ext, bool &want_shadow)
{
    const rt_light &light = S.lights[0];
    uint32_t status = (meta.y >> 20) & 0xFu;
    PathState ps;
    path_begin(ps, Ray{mk(0, 0, 0), mk(0, 0, 1)});
    ps.rng.state = 0; ps.rng.inc = 1;
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
                ps.result = ps.result + textured_color(S, S.background, S.background_texture, ray.d) * ps.throughput; // main.cpp:175-176
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

    return status;
}

// Step 3 for one slot: start the camera path of work item `item` (src/main.cpp:294-323) in it.
RT_DEV uint32_t wf_start_slot(const DFrame &F, const WfPool &P, uint32_t slot, uint64_t item, const float *halton,
                          const int32_t *active, int k0, int nk, float4 *sbuf, uint4 &meta, bool &want_ext)
{
    uint32_t status;
    PathState ps;
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
    return status;
}

// Work items are reserved by the workgroup, kLogicChunk at a time, and what a launch does not
// use up stays with the workgroup for the next one (Q.block_items): the item counter sees one
// atomic per chunk instead of one per wave and launch.  The grid must be the same in every launch.
constexpr uint32_t kLogicChunk = 2048;

__global__ __launch_bounds__(kBlock) void k_wf_logic(DScene S, DFrame F, const float *__restrict__ halton,
                                                     const int32_t *__restrict__ active, uint64_t n_items, int k0, int nk,
                                                     float4 *__restrict__ sbuf, FramePlanes out, WfPool P, WfQueue Q,
                                                     DCounters *ctr, int first_call)
{
    __shared__ uint32_t s_wave[kBlock / 64];
    __shared__ unsigned long long s_r0, s_r1;
    __shared__ uint32_t s_l0, s_l1;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    // thread 0 keeps the workgroup's reserved item range [c_next, c_end)
    unsigned long long c_next = 0, c_end = 0;
    if (threadIdx.x == 0 && !first_call) { c_next = Q.block_items[blockIdx.x * 2]; c_end = Q.block_items[blockIdx.x * 2 + 1]; }
    uint32_t new_paths = 0, n_alive = 0;

    for (uint32_t base = blockIdx.x * kBlock; base < P.n_slots; base += gridDim.x * kBlock) {
    const uint32_t slot = base + threadIdx.x;
    const bool in_pool = slot < P.n_slots;
    uint32_t status = WF_EMPTY;
    uint4 meta = make_uint4(0, 0, 0, 0);
    bool want_ext = false, want_shadow = false;

    if (in_pool && !first_call) {
        meta = P.meta[slot];
        status = (meta.y >> 20) & 0xFu;
    }
    if (status != WF_EMPTY) status = wf_shade_slot(S, F, P, slot, meta, sbuf, out, want_ext, want_shadow);

    // 3. refill: finished and never-used slots take the next work items; the workgroup's requests are
    //    numbered through, served from what is left of its reserved range and then from a fresh chunk
    {
        const bool need = in_pool && status == WF_EMPTY;
        const uint64_t m = __ballot(need);
        const uint32_t rank_w = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) s_wave[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t total = 0;
            for (uint32_t w = 0; w < kBlock / 64; ++w) total += s_wave[w];
            const uint32_t l0 = (uint32_t)(c_end - c_next < total ? c_end - c_next : total);
            s_r0 = c_next; s_l0 = l0; c_next += l0;
            uint32_t l1 = 0;
            if (total > l0) {
                const unsigned long long b = atomicAdd(Q.item, (unsigned long long)kLogicChunk);
                c_next = b < n_items ? b : n_items;
                c_end = b + kLogicChunk < n_items ? b + kLogicChunk : n_items;
                l1 = (uint32_t)(c_end - c_next < total - l0 ? c_end - c_next : total - l0);
                s_r1 = c_next; c_next += l1;
            }
            s_l1 = l1;
        }
        __syncthreads();
        {
            uint32_t idx = rank_w;
            for (uint32_t w = 0; w < wave; ++w) idx += s_wave[w];
            uint64_t item = n_items; // none
            if (idx < s_l0) item = s_r0 + idx;
            else if (idx - s_l0 < s_l1) item = s_r1 + (idx - s_l0);
            if (need && item < n_items) {
                status = wf_start_slot(F, P, slot, item, halton, active, k0, nk, sbuf, meta, want_ext);
                ++new_paths;
            } else if (need) {
                meta.y = pack_meta(0, 0, WF_EMPTY, false);
                P.meta[slot] = meta;
            }
        }
    }

    // 4. what the slot wants traced
    if (in_pool) P.flags[slot] = (want_ext ? WF_HAS_EXT : 0u) | (want_shadow ? WF_HAS_SHADOW : 0u);
    if (status != WF_EMPTY) ++n_alive;
    __syncthreads(); // s_wave and the ranges are reused by the next round of slots
    }

    // liveness and camera paths started here (closest / shadow rays are counted where they are traced)
    const uint64_t al = wave_sum(n_alive), p = wave_sum(new_paths);
    if (lane == 0) {
        if (al) atomicAdd(Q.alive, (uint32_t)al);
        if (p) atomicAdd(&ctr->paths, (unsigned long long)p);
    }
    if (threadIdx.x == 0) { Q.block_items[blockIdx.x * 2] = c_next; Q.block_items[blockIdx.x * 2 + 1] = c_end; }
}

} // namespace rtd
