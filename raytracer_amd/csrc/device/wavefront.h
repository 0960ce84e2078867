// Wavefront form of the path tracer: the per-path arithmetic of path_step() (shade.h, src/main.cpp:160-268),
// split so that rays, not paths, are the unit of SIMD work, with compacted lists between the stages.
//
// All camera paths of a sampling round are in flight at once.  A path has a slot (= its item index in the sample
// buffer, where its radiance accumulates), but its state does not live there: it travels with the path from the list
// of one bounce to the list of the next, so every list is dense - entry e of a list is entry e of the state arrays -
// and a kernel that goes through a list reads whole memory lines of live paths, however few of the round's paths are
// left.  Per bounce the host runs
//
//   k_rays<WfIO>   (rays.h) one launch over "extension rays of the live paths, then the shadow rays of the previous
//                  bounce", with ray-level regeneration: a lane that finishes a ray takes the next one at once, so BVH
//                  steps run with nearly full waves however unequal the walks are.  A closest-hit answer goes to the
//                  ray's list entry; an unoccluded shadow ray adds the next-event contribution it carries to its
//                  path's accumulator (one shadow ray per path and bounce, so no two lanes add to the same one).
//   k_wf_shade     one lane = one live path: shades the new hit (src/main.cpp:173-264) and appends the shadow ray with
//                  the contribution it gates to the shadow list and, if the path goes on, the path with its next
//                  extension ray to the next bounce's list.  A workgroup reserves the places of 256 paths' entries
//                  with one atomic: a device-wide atomic per wave (4 M of them on the first bounce of a 1080p x 64 spp
//                  round, all on one address) took longer than the shading itself.
//
// A path's random stream, draw order and every floating-point expression are those of path_step().  The rays go
// through the four-wide walk of the ray-list kernel, which can report a hit one ulp away from the binary walk's when
// two triangles are hit within an ulp of each other (about one ray in ten million); the depth and id planes therefore
// do not come from these rays but from k_planes, which sends the first sample's camera ray of every pixel through the
// reference's binary walk (trace_closest).  Additions to an accumulator happen in the path's own order: the
// next-event term of vertex b is added by the ray launch that follows k_wf_shade(b), before k_wf_shade(b+1) adds what
// the path finds at vertex b+1.
#pragma once
#include "rays.h"

namespace rtd {

// the paths of one extension list, entry by entry; every access is a 16-byte (or 8-byte) one
struct WfPaths {
    float4 *ray_o;  // extension ray: origin xyz, w = probability of the bounce that produced it
    float4 *ray_d;  //                direction xyz, w = bits: bounce | lobe of that bounce << 16
    float4 *thr;    // path throughput xyz, w = bits: the path's slot
    uint2 *rng;     // PCG state (the increment follows from the sample index)
};
struct WfState {
    WfPaths in;     // the list the kernel goes through (the camera paths: entry e is slot e, thr and rng not yet written)
    WfPaths out;    // the next bounce's list, which k_wf_shade fills
    float4 *hit;    // per entry of `in`, the answer to its extension ray: t, node, face | front << 31, u
    float *hit_v;   //                                                     v
    float4 *sh_o;   // per entry of the shadow list: origin xyz, w = t_max
    float4 *sh_d;   //                               direction xyz
    float4 *sh_c;   //                               next-event contribution the ray gates, w = bits: the path's slot
    float4 *acc;    // per slot: radiance accumulators = the sample buffer of the round
};

// The lists of one bounce and their lengths, all in device memory: the host need not know the lengths to launch the
// kernels that read them.  A round's lengths are one array of these, entry b for bounce b.
struct WfCounts {
    uint32_t n_ext;    // extension rays: paths that go on from the previous bounce (bounce 0: the round's camera paths)
    uint32_t n_shadow; // shadow rays the previous bounce's shading left behind
    uint32_t pad[2];
};

// ray list of k_rays: the extension rays of the list's paths, then the shadow rays of the shadow list
struct WfIO {
    static constexpr bool kWide = true; // four-wide BVH walk: the depth / id planes come from k_planes, not from these rays
    static constexpr bool kDeferShadow = true;
    struct Pending { float4 c, a; };
    WfState P;
    const WfCounts *cnt;     // this bounce's list lengths
    uint32_t shadows_only;   // leave the extension rays alone: k_wf_finish takes them
    RT_DEV uint32_t setup(uint32_t, uint32_t &n_ext) const
    {
        n_ext = shadows_only ? 0u : cnt->n_ext;
        return n_ext + cnt->n_shadow;
    }
    // the lane keeps the ray's place in its own list
    RT_DEV bool fetch(uint32_t e, uint32_t n_ext, float4 &o, float4 &d, bool &any_hit, float &t_max, uint32_t &at) const
    {
        any_hit = e >= n_ext;
        at = any_hit ? e - n_ext : e;
        if (any_hit) { o = P.sh_o[at]; d = P.sh_d[at]; }
        else { o = P.in.ray_o[at]; d = P.in.ray_d[at]; }
        t_max = any_hit ? o.w : 0.0f;
        return true;
    }
    // an unoccluded shadow ray adds the contribution it gates to its path's accumulator
    RT_DEV void begin_put_shadow(uint32_t at, Pending &p) const { p.c = P.sh_c[at]; p.a = P.acc[__float_as_uint(p.c.w)]; }
    RT_DEV void end_put_shadow(uint32_t, const Pending &p) const
    {
        P.acc[__float_as_uint(p.c.w)] = make_float4(p.a.x + p.c.x, p.a.y + p.c.y, p.a.z + p.c.z, p.a.w);
    }
    RT_DEV void put_closest(uint32_t at, float t, int node, uint32_t face, int front, float u, float v) const
    {
        P.hit[at] = make_float4(t, __uint_as_float((uint32_t)node), __uint_as_float(face | ((uint32_t)front << 31)), u);
        P.hit_v[at] = v;
    }
    RT_DEV void put_shadow(uint32_t at, bool occluded) const
    {
        if (occluded) return;
        Pending p;
        begin_put_shadow(at, p);
        end_put_shadow(at, p);
    }
};

// camera rays of the round's items (src/main.cpp:294-323)
__global__ __launch_bounds__(kBlock) void k_wf_generate(DFrame F, const float *__restrict__ halton, const int32_t *__restrict__ active,
                                                        uint32_t n_items, int k0, int nk, WfState P, WfCounts *cnt0)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) *cnt0 = WfCounts{n_items, 0u, {0u, 0u}}; // the first bounce's lists (the rest of the array is zero)
    for (uint32_t item = blockIdx.x * kBlock + threadIdx.x; item < n_items; item += gridDim.x * kBlock) {
        const uint32_t a = item / (uint32_t)nk;
        const int k = k0 + (int)(item - a * (uint32_t)nk);
        const int pix = active[a];
        Rng rng;
        rng_set_sequence(rng, 0, pixel_seed(F.seed, (uint64_t)pix));
        float off[4];
        off[0] = rng_float(rng); off[1] = rng_float(rng); off[2] = rng_float(rng); off[3] = rng_float(rng);
        const Ray r = camera_ray(F, halton, pix % F.width, pix / F.width, k, off);
        P.in.ray_o[item] = make_float4(r.p.x, r.p.y, r.p.z, 1.0f);
        P.in.ray_d[item] = make_float4(r.d.x, r.d.y, r.d.z, __uint_as_float(0u));
    }
}

// depth / id planes of a round that starts at sample 0: the first sample's camera ray of every listed pixel, through the
// reference's own walk order (trace_closest) - bit for bit what the reference's TraceRay returns
template <int LEVELS>
__global__ __launch_bounds__(kBlock) void k_planes(DScene S, DFrame F, const float *__restrict__ halton,
                                                   const int32_t *__restrict__ active, uint32_t n_active, FramePlanes out)
{
    extern __shared__ uint32_t lds[];
    const LaneStack st = lane_stack(lds, S);
    for (uint32_t a = blockIdx.x * kBlock + threadIdx.x; a < n_active; a += gridDim.x * kBlock) {
        const int pix = active[a];
        Rng rng;
        rng_set_sequence(rng, 0, pixel_seed(F.seed, (uint64_t)pix));
        float off[4];
        off[0] = rng_float(rng); off[1] = rng_float(rng); off[2] = rng_float(rng); off[3] = rng_float(rng);
        const Ray ray = camera_ray(F, halton, pix % F.width, pix / F.width, 0, off);
        Hit h;
        Tally<false> none;
        const bool hit = trace_closest<false, LEVELS>(S, ray, st, h, none);
        if (out.z) out.z[pix] = h.z;
        if (out.object_id) out.object_id[pix] = hit ? h.node : -1;
    }
}

// What a path does at a surface vertex (src/main.cpp:212-264), with the shadow ray left for later: the next-event
// sample and the contribution its shadow ray gates (worked out now, with this vertex' throughput), then the
// material's sample.  Returns whether the path goes on (new_o / new_d / thr then describe the next extension ray).
struct VertexRays {
    float4 s_o, s_d, s_c;  // shadow ray: origin + t_max, direction, the contribution it gates
    float4 new_o, new_d;   // next extension ray in the slot layout of WfState
};
RT_DEV bool surface_vertex(const DScene &S, const DFrame &F, const rt_light &light, const Ray &ray, Hit &h, V3 &thr, int bounce,
                           Rng &rng, VertexRays &o)
{
    finish_hit(S, ray, h, false);
    Surface s; // SamplerInfo::SetHit (renderer.h:145-153)
    const V3 normal = normalize(h.n);
    s.p = h.p; s.n = normal; s.v = -normalize(ray.d); s.front = h.front;
    const rt_material &mtl = S.materials[hit_material(S, h.node, h.face)];
    // next-event estimation (main.cpp:214-251): the shadow ray is always sent, as in the reference
    Sample ne;
    light_generate(light, s.p, rng, ne);
    const float sgn = h.front ? 1.0f : -1.0f;
    const V3 so = h.p + (normal * F.bias) * sgn;
    V3 c = mk(0.0f, 0.0f, 0.0f);
    const float cosS = std_max(0.0f, dot(normal, ne.dir));
    if (cosS > 0.0f && ne.prob > 0.0f) {
        const float mp = mtl_pdf(mtl, s, ne.dir);
        float weight = 1.0f;
        if (mp > 0.0f) weight = (ne.prob * ne.prob) / (ne.prob * ne.prob + mp * mp);
        // the specular term of main.cpp:241-246 is gated on a field GetSampleInfo never writes: diffuse only
        const V3 brdf = ld3(mtl.diffuse) / kPi;
        c = ((((brdf * cosS) * ne.mult) * weight) / ne.prob) * thr;
    }
    o.s_o = make_float4(so.x, so.y, so.z, ne.dist - F.bias);
    o.s_d = make_float4(ne.dir.x, ne.dir.y, ne.dir.z, 0.0f);
    o.s_c = make_float4(c.x, c.y, c.z, 0.0f);
    // indirect bounce (main.cpp:253-264)
    Sample in;
    if (!(mtl_generate(mtl, s, rng, in) && bounce + 1 < F.max_bounces)) return false;
    const float sg = (dot(normal, in.dir) > 0.0f) ? 1.0f : -1.0f;
    const V3 no = h.p + (normal * F.bias) * sg;
    thr = thr * (in.mult / in.prob);
    o.new_o = make_float4(no.x, no.y, no.z, in.prob);
    o.new_d = make_float4(in.dir.x, in.dir.y, in.dir.z, __uint_as_float((uint32_t)(bounce + 1) | ((uint32_t)in.lobe << 16)));
    return true;
}

// A ray whose answer stays with the lane that traced it (k_wf_finish): nothing is fetched, nothing stored.
struct LaneIO {
    static constexpr bool kWide = true;
    static constexpr bool kDeferShadow = true; // the kernel reads RayWalker::unoccluded itself
    typedef NoPending Pending;
    Hit *h;
    RT_DEV void begin_put_shadow(uint32_t, Pending &) const {}
    RT_DEV void end_put_shadow(uint32_t, const Pending &) const {}
    RT_DEV void put_shadow(uint32_t, bool) const {}
    RT_DEV void put_closest(uint32_t, float t, int node, uint32_t face, int front, float u, float v) const
    {
        h->z = t; h->node = node; h->face = face; h->front = front; h->u = u; h->v = v;
    }
};

// The last of a round's paths, to their ends in one launch.  After a few bounces a handful of paths is left (glass,
// mirrors, valleys of a mesh), and a bounce of the pipeline - a ray launch for a thousand rays, a shading launch, the
// host reading the counts - takes 0.2 ms however few they are: forty more of those are 8 to 10 ms, a third of a
// frame's time on one eighth of the image.  What is left then is a chain of dependent steps - the longest path's
// bounces, each a closest-hit walk, a shadow walk and the shading between them - on a nearly idle chip, so this
// kernel is built for the length of that chain, not for throughput: a path belongs to a PAIR of lanes.  The even
// lane keeps the path and traces its extension rays; the odd lane traces the shadow ray of the vertex before, at the
// same time (the next direction does not depend on the shadow ray's answer, only the sum does, and that is added in
// the path's own order: next-event term of vertex b, then whatever vertex b + 1 adds).  Both walk through the ray-list
// kernel's walker (four-wide BVH, one step of one ray as the unit), so a bounce costs one walk's worth of dependent
// fetches instead of two binary ones.  A pair whose path has ended takes the next one from the list.
template <bool COUNT, int LEVELS>
__global__ __launch_bounds__(kBlock) void k_wf_finish(DScene S, DFrame F, int k0, int nk, WfState P,
                                                      const WfCounts *__restrict__ cnt, DCounters *ctr, unsigned long long *queue)
{
    const uint32_t n_in = cnt->n_ext;
    extern __shared__ uint32_t lds[];
    const LaneStack st = lane_stack(lds, S);
    const rt_light light = load_uniform(&S.lights[0]);
    const uint32_t lane = threadIdx.x & 63u;
    const bool keeper = (lane & 1u) == 0; // even lane: the path and its extension rays; odd lane: its shadow rays
    const int even = (int)(lane & ~1u), odd = (int)(lane | 1u);
    PathTally pt;
    Tally<COUNT> tl;
    Hit h;
    h.z = FLT_MAX; h.node = -1; h.face = 0; h.front = 1; h.u = 0; h.v = 0;
    const LaneIO io{&h};
    RayWalker<COUNT, LEVELS, LaneIO> w;
    // the keeper's path
    bool have_path = false, goes_on = false, owes_shadow = false;
    uint32_t slot = 0;
    Ray ray{mk(0, 0, 0), mk(0, 0, 1)};
    V3 thr = mk(0, 0, 0), result = mk(0, 0, 0);
    float acc_w = 0.0f, lastProb = 1.0f;
    int lastLobe = LOBE_NONE, bounce = 0;
    Rng rng;
    rng.state = 0; rng.inc = 1;
    VertexRays vr;
    vr.s_o = vr.s_d = vr.s_c = vr.new_o = vr.new_d = make_float4(0, 0, 0, 0);
    bool exhausted = false;
    pt.tick();
    for (;;) {
        // ---- pairs without a path take the next ones of the list (one atomic per wave) ----
        const uint64_t free_pairs = __ballot(keeper && !have_path);
        if (free_pairs != 0 && !exhausted) {
            const uint32_t want = (uint32_t)__popcll(free_pairs);
            unsigned long long b = 0;
            if (lane == 0) b = atomicAdd(queue, (unsigned long long)want);
            b = __shfl(b, 0, 64);
            const uint32_t rank = (uint32_t)__popcll(free_pairs & ((1ull << lane) - 1ull));
            if (b + want >= n_in) exhausted = true;
            if (keeper && !have_path && b + rank < n_in) {
                const uint32_t at = (uint32_t)b + rank;
                const float4 ro = P.in.ray_o[at], rd = P.in.ray_d[at], t = P.in.thr[at];
                const uint2 rs = P.in.rng[at];
                slot = __float_as_uint(t.w);
                const float4 a = P.acc[slot];
                const int k = k0 + (int)(slot % (uint32_t)nk);
                const uint32_t bits = __float_as_uint(rd.w);
                ray = Ray{mk(ro.x, ro.y, ro.z), mk(rd.x, rd.y, rd.z)};
                thr = mk(t.x, t.y, t.z);
                result = mk(a.x, a.y, a.z); acc_w = a.w;
                rng.state = ((uint64_t)rs.y << 32) | rs.x;
                rng.inc = (((uint64_t)k + 1) << 1u) | 1u;
                lastProb = ro.w; lastLobe = (int)(bits >> 16); bounce = (int)(bits & 0xFFFFu);
                have_path = true; goes_on = true; owes_shadow = false;
            }
        }
        // ---- this trip's rays: the path's extension ray on the even lane, the last vertex' shadow ray on the odd one ----
        const bool sh = __shfl((int)(keeper && owes_shadow), even, 64) != 0;
        const float4 so = make_float4(__shfl(vr.s_o.x, even, 64), __shfl(vr.s_o.y, even, 64), __shfl(vr.s_o.z, even, 64), __shfl(vr.s_o.w, even, 64));
        const float4 sd = make_float4(__shfl(vr.s_d.x, even, 64), __shfl(vr.s_d.y, even, 64), __shfl(vr.s_d.z, even, 64), 0.0f);
        if (keeper && have_path && goes_on) w.start(slot, make_float4(ray.p.x, ray.p.y, ray.p.z, 0.0f), make_float4(ray.d.x, ray.d.y, ray.d.z, 0.0f), false, 0.0f, pt);
        if (!keeper && sh) w.start(0, so, sd, true, so.w, pt);
        if (!__any(w.active)) break; // no pair has a path left, and the list is empty
        while (__any(w.active)) w.phases(S, io, st, pt, tl);
        // ---- the shadow ray's answer first (vertex b), then the new vertex (b + 1) ----
        const bool lit = __shfl((int)(!keeper && w.unoccluded), odd, 64) != 0;
        w.unoccluded = false;
        if (keeper && owes_shadow) {
            if (lit) result = result + mk(vr.s_c.x, vr.s_c.y, vr.s_c.z);
            owes_shadow = false;
        }
        if (keeper && have_path && goes_on) {
            w.finish_closest(S, io); // the lights, then the answer into h
            goes_on = false;
            if (h.node == -1) { // main.cpp:175-176
                result = result + textured_color(S, S.background, S.background_texture, ray.d) * thr;
            } else if (h.node < 0) { // light (main.cpp:184-210)
                const V3 rad = light_radiance(light);
                if (bounce == 0) result = result + rad * thr;
                else {
                    float weight = 1.0f;
                    if (lastLobe == LOBE_DIFFUSE) {
                        const float lp = light_pdf(light, ray.p, ray.d);
                        if (lp > 0.0f) weight = (lastProb * lastProb) / (lastProb * lastProb + lp * lp);
                    }
                    result = result + (rad * thr) * weight;
                }
            } else {
                goes_on = surface_vertex(S, F, light, ray, h, thr, bounce, rng, vr);
                owes_shadow = true;
                if (goes_on) {
                    ray = Ray{mk(vr.new_o.x, vr.new_o.y, vr.new_o.z), mk(vr.new_d.x, vr.new_d.y, vr.new_d.z)};
                    lastProb = vr.new_o.w;
                    const uint32_t bits = __float_as_uint(vr.new_d.w);
                    bounce = (int)(bits & 0xFFFFu); lastLobe = (int)(bits >> 16);
                }
            }
        }
        if (keeper && have_path && !goes_on && !owes_shadow) { // the path's radiance is complete
            P.acc[slot] = make_float4(result.x, result.y, result.z, acc_w);
            have_path = false;
        }
    }
    flush_counters<COUNT>(ctr, pt, tl, 0);
}

#ifndef RT_SHADE_ROUNDS
#define RT_SHADE_ROUNDS 8
#endif
constexpr uint32_t kShadeRounds = RT_SHADE_ROUNDS;        // paths per lane and chunk
constexpr uint32_t kShadeChunk = kShadeRounds * kBlock;   // paths a workgroup orders by what they hit
constexpr uint32_t kShadeBins = 64;                       // nothing, lights, then the scene's nodes (the last bin takes the rest)

// position of this lane's entry among the workgroup's entries of this trip: one LDS atomic per wave
RT_DEV uint32_t lds_append(uint32_t *count, bool want)
{
    const uint64_t m = __ballot(want);
    if (m == 0) return 0;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t leader = (uint32_t)__ffsll((unsigned long long)m) - 1u;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(count, (uint32_t)__popcll(m));
    base = __shfl(base, (int)leader, 64);
    return base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
}

// One bounce of every live path.  FIRST: the camera rays' hits - entry e is slot e, the path state is its initial
// value and need not have been written.  Appends to the lists of *next the paths that go on and the shadow rays.
template <bool FIRST>
__global__ __launch_bounds__(kBlock) void k_wf_shade(DScene S, DFrame F, const int32_t *__restrict__ active, int k0, int nk,
                                                     WfState P, const WfCounts *__restrict__ cnt, WfCounts *next, FramePlanes out)
{
    const uint32_t n_in = cnt->n_ext;
    // n_ext and n_shadow of the next bounce as one 64-bit word (low: paths that go on, high: shadow rays): a trip's
    // places in both lists are reserved with one atomic
    unsigned long long *const counts = (unsigned long long *)&next->n_ext;
    __shared__ uint32_t s_order[kShadeChunk]; // the chunk's paths, those that hit the same thing next to each other
    __shared__ uint32_t s_bin[kShadeBins];
    __shared__ uint32_t s_n[2][2], s_base[2]; // [trip parity][0: paths that go on, 1: shadow rays]
    const rt_light light = load_uniform(&S.lights[0]);
    const uint32_t n_chunks = (n_in + kShadeChunk - 1) / kShadeChunk;
    for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    if (threadIdx.x < 4) (&s_n[0][0])[threadIdx.x] = 0;
    if (threadIdx.x < kShadeBins) s_bin[threadIdx.x] = 0;
    __syncthreads();
    // What a path does here depends on what it hit (nothing, a light, a sphere, a plane, a mesh; which material), and
    // after the first bounce neighbouring list entries hit unrelated things.  The chunk's paths are therefore dealt
    // to the lanes ordered by the object they hit (a counting sort in LDS): a wave then runs one kind of shading,
    // not all of them.  Every path's arithmetic is its own, so the order changes no result.
    const uint32_t chunk_first = chunk * kShadeChunk;
    const uint32_t chunk_n = n_in - chunk_first < kShadeChunk ? n_in - chunk_first : kShadeChunk;
    // (Not the camera rays' hits: a chunk of them is a few neighbouring pixels' samples, which hit the same things.)
    if (FIRST) {
#pragma unroll
        for (uint32_t r = 0; r < kShadeRounds; ++r) s_order[r * kBlock + threadIdx.x] = chunk_first + r * kBlock + threadIdx.x;
    } else {
    uint32_t my_key[kShadeRounds], my_rank[kShadeRounds];
#pragma unroll
    for (uint32_t r = 0; r < kShadeRounds; ++r) {
        const uint32_t i = r * kBlock + threadIdx.x;
        if (i < chunk_n) {
            const int node = (int)__float_as_uint(P.hit[chunk_first + i].y); // -1: nothing, below: a light
            my_key[r] = node < 0 ? (node == -1 ? 1u : 0u) : ((uint32_t)node + 2u < kShadeBins ? (uint32_t)node + 2u : kShadeBins - 1u);
            my_rank[r] = atomicAdd(&s_bin[my_key[r]], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) { // bin counts -> bin starts
        const uint32_t c = threadIdx.x < kShadeBins ? s_bin[threadIdx.x] : 0u;
        uint32_t inc = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o, 64); if ((int)threadIdx.x >= o) inc += t; }
        if (threadIdx.x < kShadeBins) s_bin[threadIdx.x] = inc - c;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t r = 0; r < kShadeRounds; ++r) {
        const uint32_t i = r * kBlock + threadIdx.x;
        if (i < chunk_n) s_order[s_bin[my_key[r]] + my_rank[r]] = chunk_first + i;
    }
    }
    __syncthreads();
    for (uint32_t r = 0; r < kShadeRounds; ++r) { // every lane takes every trip: the appends are workgroup-wide
        const uint32_t i = r * kBlock + threadIdx.x;
        const bool live = i < chunk_n;
        bool want_ext = false, want_shadow = false;
        uint32_t slot = 0;
        float4 new_o = make_float4(0, 0, 0, 0), new_d = new_o, new_t = new_o, s_o = new_o, s_d = new_o, s_c = new_o;
        uint2 new_rng = make_uint2(0u, 0u);
        if (live) {
            const uint32_t at = s_order[i]; // the path's place in the list
            const float4 ro = P.in.ray_o[at], rd = P.in.ray_d[at];
            const Ray ray{mk(ro.x, ro.y, ro.z), mk(rd.x, rd.y, rd.z)};
            V3 thr = mk(1.0f, 1.0f, 1.0f);
            float lastProb = 1.0f;
            int lastLobe = LOBE_NONE, bounce = 0;
            slot = at;
            if (!FIRST) {
                const float4 t = P.in.thr[at];
                thr = mk(t.x, t.y, t.z);
                slot = __float_as_uint(t.w);
                lastProb = ro.w;
                const uint32_t bits = __float_as_uint(rd.w);
                bounce = (int)(bits & 0xFFFFu); lastLobe = (int)(bits >> 16);
            }
            const uint32_t a = slot / (uint32_t)nk;
            const int k = k0 + (int)(slot - a * (uint32_t)nk);
            const float4 hr = P.hit[at];
            Hit h;
            h.z = hr.x; h.node = (int)__float_as_uint(hr.y);
            const uint32_t ff = __float_as_uint(hr.z);
            h.face = ff & 0x7FFFFFFFu; h.front = (int)(ff >> 31);
            h.u = hr.w; h.v = P.hit_v[at];
            const bool hit = h.node != -1;
            V3 add = mk(0.0f, 0.0f, 0.0f);
            bool adds = false;
            if (!hit) { // main.cpp:175-176
                add = textured_color(S, S.background, S.background_texture, ray.d) * thr;
                adds = true;
            } else if (h.node < 0) { // light (main.cpp:184-210)
                const V3 rad = light_radiance(light);
                if (bounce == 0) add = rad * thr;
                else {
                    float weight = 1.0f;
                    if (lastLobe == LOBE_DIFFUSE) {
                        const float lp = light_pdf(light, ray.p, ray.d);
                        if (lp > 0.0f) weight = (lastProb * lastProb) / (lastProb * lastProb + lp * lp);
                    }
                    add = (rad * thr) * weight;
                }
                adds = true;
            } else {
                Rng rng;
                if (FIRST) rng_set_sequence(rng, (uint64_t)k + 1, pixel_seed(F.seed, (uint64_t)active[a]));
                else {
                    const uint2 rs = P.in.rng[at];
                    rng.state = ((uint64_t)rs.y << 32) | rs.x;
                    rng.inc = (((uint64_t)k + 1) << 1u) | 1u;
                }
                // (the arithmetic of surface_vertex, kept in line: through the call this kernel came out 10 % slower)
                finish_hit(S, ray, h, false);
                Surface s; // SamplerInfo::SetHit (renderer.h:145-153)
                const V3 normal = normalize(h.n);
                s.p = h.p; s.n = normal; s.v = -normalize(ray.d); s.front = h.front;
                const rt_material &mtl = S.materials[hit_material(S, h.node, h.face)];
                // next-event estimation (main.cpp:214-251): the shadow ray is always sent, as in the reference; what it
                // gates is worked out now, with this vertex' throughput
                Sample ne;
                light_generate(light, s.p, rng, ne);
                const float sgn = h.front ? 1.0f : -1.0f;
                const V3 so = h.p + (normal * F.bias) * sgn;
                V3 c = mk(0.0f, 0.0f, 0.0f);
                const float cosS = std_max(0.0f, dot(normal, ne.dir));
                if (cosS > 0.0f && ne.prob > 0.0f) {
                    const float mp = mtl_pdf(mtl, s, ne.dir);
                    float weight = 1.0f;
                    if (mp > 0.0f) weight = (ne.prob * ne.prob) / (ne.prob * ne.prob + mp * mp);
                    // the specular term of main.cpp:241-246 is gated on a field GetSampleInfo never writes: diffuse only
                    const V3 brdf = ld3(mtl.diffuse) / kPi;
                    c = ((((brdf * cosS) * ne.mult) * weight) / ne.prob) * thr;
                }
                s_o = make_float4(so.x, so.y, so.z, ne.dist - F.bias);
                s_d = make_float4(ne.dir.x, ne.dir.y, ne.dir.z, 0.0f);
                s_c = make_float4(c.x, c.y, c.z, __uint_as_float(slot));
                want_shadow = true;
                // indirect bounce (main.cpp:253-264)
                Sample in;
                if (mtl_generate(mtl, s, rng, in) && bounce + 1 < F.max_bounces) {
                    const float sg = (dot(normal, in.dir) > 0.0f) ? 1.0f : -1.0f;
                    const V3 no = h.p + (normal * F.bias) * sg;
                    thr = thr * (in.mult / in.prob);
                    new_o = make_float4(no.x, no.y, no.z, in.prob);
                    new_d = make_float4(in.dir.x, in.dir.y, in.dir.z, __uint_as_float((uint32_t)(bounce + 1) | ((uint32_t)in.lobe << 16)));
                    new_t = make_float4(thr.x, thr.y, thr.z, __uint_as_float(slot));
                    new_rng = make_uint2((uint32_t)rng.state, (uint32_t)(rng.state >> 32));
                    want_ext = true;
                }
            }
            if (FIRST) { add = mk(0.0f, 0.0f, 0.0f) + add; P.acc[slot] = make_float4(add.x, add.y, add.z, 0.0f); } // the accumulator starts here
            else if (adds) {
                float4 acc = P.acc[slot];
                acc.x = acc.x + add.x; acc.y = acc.y + add.y; acc.z = acc.z + add.z;
                P.acc[slot] = acc;
            }
        }
        // the trip's places in the two lists
        uint32_t *const n_trip = s_n[r & 1u];
        const uint32_t qi = lds_append(&n_trip[0], want_ext);
        const uint32_t si = lds_append(&n_trip[1], want_shadow);
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t n_e = n_trip[0], n_s = n_trip[1];
            unsigned long long b = 0;
            if (n_e | n_s) b = atomicAdd(counts, (unsigned long long)n_e | ((unsigned long long)n_s << 32));
            s_base[0] = (uint32_t)b; s_base[1] = (uint32_t)(b >> 32);
            s_n[(r & 1u) ^ 1u][0] = 0; s_n[(r & 1u) ^ 1u][1] = 0; // the next trip's counters
        }
        __syncthreads();
        if (want_ext) {
            const uint32_t e = s_base[0] + qi;
            P.out.ray_o[e] = new_o; P.out.ray_d[e] = new_d; P.out.thr[e] = new_t; P.out.rng[e] = new_rng;
        }
        if (want_shadow) {
            const uint32_t j = s_base[1] + si;
            P.sh_o[j] = s_o; P.sh_d[j] = s_d; P.sh_c[j] = s_c;
        }
    }
    __syncthreads(); // s_order and the counters are reused by the next chunk
    }
}

} // namespace rtd
