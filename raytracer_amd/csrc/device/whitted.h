// Whitted-style shading of the reference's project-12 snapshot (p12 = project12.zip) with the
// photon maps switched off:
//   MtlBlinn::Shade              p12 src/materials.cpp:14-202
//   PointLight::Illuminate       p12 src/lights.h:69-108   (adaptive 4..64-ray soft shadows)
//   ShadeInfo::TraceSecondaryRay p12 src/main.cpp:132-155
//   the pixel loop's sample      p12 src/main.cpp:212-231
//
// The reference recurses: Shade -> TraceSecondaryRay -> Shade, up to three children per level
// (Monte-Carlo indirect, glossy reflection, refraction with its Fresnel reflection), and draws
// from one random stream in depth-first order.  A lane runs the same depth-first walk with an
// explicit stack of shading frames and a small state machine whose unit of work is ONE ray:
// each trip of the kernel loop a lane either traces one closest-hit ray (and sets up the frame
// it lands on), or traces one shadow ray of the soft-shadow fan, or takes a new camera sample
// from the work queue.  Lanes of a wave therefore stay busy although one sample may need 5 and
// its neighbour 400 rays.
#pragma once
#include "photon.h"

namespace rtd {

#ifndef RT_GATHER_QUORUM
#define RT_GATHER_QUORUM 64 // measured on the photon-box frame: 16 / 32 / 48 / 64 lanes -> 4.04 / 3.56 / 3.11 / 2.99 s
#endif
#ifndef RT_WHITTED_MIN_WAVES
#define RT_WHITTED_MIN_WAVES 3 // waves per SIMD the register allocator must leave room for: 170 registers; at 4 (128 registers) the kernel spills 220 values and is 5 % slower on both Whitted workloads, at 2 it is 18 % slower
#endif
#ifndef RT_WHITTED_CULL
#define RT_WHITTED_CULL 1 // skip objects whose world-space box no ray of the wave reaches (trace_closest / trace_shadow); 0 builds the votes out
#endif
constexpr bool kWhittedCull = RT_WHITTED_CULL != 0;
constexpr int kGatherQuorum = RT_GATHER_QUORUM; // parked lanes that start a round of photon lookups
constexpr int kMaxWhittedBounces = 8; // the snapshot compiles in 5; the tree grows as 3^bounces

enum { W_IDLE = 0, W_TRACE = 1, W_SHADOW = 2, W_CTRL = 3, W_RET = 4, W_GATHER = 5 };
enum { STG_LIGHT = 0, STG_AFTER = 1, STG_BOUNCE = 6, STG_MC = 2, STG_REFL = 3, STG_REFR = 4, STG_TRANS = 5 };
enum { COMB_MC = 0, COMB_REFL = 1, COMB_TIR = 2, COMB_FRESNEL = 3, COMB_TRANS = 4 };

// One level of the Shade recursion: what Shade still needs after a child ray has returned.
struct WFrame {
    V3 final;      // colour accumulated so far
    V3 p, n, v;    // hit point, unit normal, unit view vector (SamplerInfo::SetHit, p12 renderer.h:146-154)
    V3 h;          // sampled half vector of the refraction lobe
    V3 kd;         // diffuse colour at the hit: diffuse.Eval(UVW), the one textured channel (p12 materials.cpp:22)
    float dist;    // ray parameter of the hit that created this frame
    uint32_t meta; // material index | front << 20 | pending combination << 24
};


// Blinn-Phong term of one light (p12 materials.cpp:33-47): Lambert and the normalised Blinn lobe, both weighted by
// the cosine between the normal and the light
RT_DEV void whitted_light_term(WFrame &f, const rt_material &m, V3 radiance, V3 to_light)
{
    const float cos_l = std_max(0.0f, dot(f.n, to_light));
    f.final = f.final + ((f.kd * kInvPiF) * radiance) * cos_l;
    const float cos_h = std_max(0.0f, dot(f.n, normalize(f.v + to_light)));
    const float lobe_norm = (float)((double)(m.glossiness + 2.0f) / kEightPi);
    f.final = f.final + (((ld3(m.specular) * lobe_norm) * powf(cos_h, m.glossiness)) * radiance) * cos_l;
}

// Shadow ray number i of the soft-shadow fan towards a spherical light (p12 lights.h:76-97):
// a point on the disk of radius `size` facing P, Halton(5,7) shifted by the two per-call offsets.
RT_DEV Ray whitted_fan_ray(const rt_light &L, V3 P, const float *__restrict__ halton, int period, int i, float offX, float offY)
{
    const V3 pos = ld3(L.position);
    const V3 dir = normalize(pos - P);
    V3 u, v;
    orthonormals(dir, u, v);
    const int hi = i % period;
    const float jt = frac_sum(offX, halton[128 + hi]);
    const float jr = frac_sum(offY, halton[192 + hi]);
    const float theta = (float)((double)(jt * 2.0f) * M_PI);
    const float radius = fsqrt(jr);
    float x, y; // diskRadius * cos(diskTheta), formed in double (see photon.h)
    scaled_sincos(radius, theta, y, x);
    const V3 dest = (pos + u * (x * L.size)) + v * (y * L.size);
    Ray r;
    r.d = dest - P;
    r.p = P + r.d * 0.0002f; // ShadeInfo::TraceShadowRay (p12 main.cpp:126-130)
    return r;
}

// (x u + y v) + z n (p12 materials.cpp:96,125,153)
RT_DEV V3 whitted_frame(V3 u, V3 v, V3 n, float x, float y, float z) { return (u * x + v * y) + n * z; }

// the double-precision pieces of the refraction branch (pow(float, int) is the double overload there,
// p12 materials.cpp:157-178); a*a is exact in double for a float a
RT_DEV float whitted_cos2(float ratio, float viewDotH)
{
    const double r = (double)ratio, c = (double)viewDotH;
    return (float)(1.0f - (r * r) * (1.0f - c * c));
}
RT_DEV float whitted_fresnel(float ior, float viewDotH)
{
    const double b = (double)((1 - ior) / (1 + ior));
    const float schlick = (float)(b * b);
    const double a = (double)(1 - viewDotH);
    const double a2 = a * a;
    return (float)(schlick + (1 - schlick) * (a2 * a2 * a));
}

template <bool COUNT, int LEVELS>
__global__ __launch_bounds__(kBlock, RT_WHITTED_MIN_WAVES) void k_whitted(DScene S, DFrame F, const float *__restrict__ halton,
                                                       const int32_t *__restrict__ active, uint64_t n_items, int k0, int nk,
                                                       float4 *__restrict__ sbuf, FramePlanes out, DCounters *ctr,
                                                       unsigned long long *queue)
{
    extern __shared__ uint32_t lds[];
    const LaneStack st = lane_stack(lds, S);
    const uint32_t lane = threadIdx.x & 63u;
    PathTally pt;
    Tally<COUNT> tl;
    uint32_t paths = 0, gathers = 0, visits = 0;
    const int period = F.halton_period;
    const int pm = F.photon_flags;

    WorkQueue wq(queue, n_items, F.chunk, S.cancel, S.progress);
    int state = W_IDLE;
    uint64_t item = 0;
    int pix = 0, k = 0, ps_h = 0;
    float fz = FLT_MAX; int fid = -1;
    Rng rng; rng.state = 0; rng.inc = 1;
    Ray ray{mk(0, 0, 0), mk(0, 0, 1)};
    float tmax = 0.0f;
    WFrame stk[kMaxWhittedBounces];
    WFrame cur{};
    int depth = 0, stage = STG_LIGHT, mtl = 0, front = 1;
    int l = 0, fan_i = 0, fan_hits = 0;
    float offX = 0.0f, offY = 0.0f;
    V3 ret = mk(0, 0, 0);
    float retDist = 0.0f;

    for (;;) {
        const uint64_t need = __ballot(state == W_IDLE);
        if (need != 0) {
            wq.refill(lane);
            const uint64_t avail = wq.end - wq.next;
            const uint32_t want = (uint32_t)__popcll(need);
            const uint32_t take = avail < want ? (uint32_t)avail : want;
            const uint32_t rank = (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
            if (state == W_IDLE && rank < take) {
                item = wq.next + rank;
                const uint32_t a = (uint32_t)(item / (uint32_t)nk);
                k = k0 + (int)(item - (uint64_t)a * (uint32_t)nk);
                pix = active[a];
                const int i = pix % F.width, j = pix / F.width;
                const uint64_t seed = pixel_seed(F.seed, (uint64_t)pix);
                rng_set_sequence(rng, 0, seed);
                float off[4];
                off[0] = rng_float(rng); off[1] = rng_float(rng); off[2] = rng_float(rng); off[3] = rng_float(rng);
                rng_set_sequence(rng, (uint64_t)k + 1, seed);
                ray = camera_ray(F, halton, i, j, k, off);
                ps_h = i % period; // SetPixelSample(i): the shader's Halton index is the pixel's x (p12 main.cpp:214)
                depth = 0;
                state = W_TRACE;
                ++paths;
            }
            wq.next += take;
        }
        if (!__any(state != W_IDLE)) {
            if (wq.exhausted) break;
            continue;
        }

        // Photon lookups are long (hundreds of kd-tree nodes) next to a ray of this kind of scene, and lanes reach
        // them at different times: parked lanes wait until the whole wave is parked or nothing else is left to do,
        // then gather together.
        const uint64_t parked = __ballot(state == W_GATHER);
        if (parked != 0 && (__popcll(parked) >= kGatherQuorum || !__any(state == W_TRACE || state == W_SHADOW))) {
            // kGatherSlots lanes at a time: that many arrays of found photons fit the wave's share of LDS
            const uint32_t rank = (uint32_t)__popcll(parked & ((1ull << lane) - 1ull));
            const uint32_t n_parked = (uint32_t)__popcll(parked);
            for (uint32_t first = 0; first < n_parked; first += kGatherSlots) {
                if (state == W_GATHER && rank >= first && rank < first + kGatherSlots) {
                    const V3 kd = cur.kd * kInvPiF;
                    const KdStack kst(st, lds, rank - first);
                    const LaneHeap heap = lane_heap(S, lds, rank - first);
                    if (pm & (RT_PM_DIRECT | RT_PM_INDIRECT)) {
                        cur.final = cur.final + kd * photon_irradiance<COUNT>(S.pmap[0], S.pmap_n[0], cur.p, cur.n, F.gather_radius, kst, heap, visits);
                        ++gathers;
                    }
                    if (pm & RT_PM_CAUSTICS) {
                        cur.final = cur.final + kd * photon_irradiance<COUNT>(S.pmap[1], S.pmap_n[1], cur.p, cur.n, F.gather_radius, kst, heap, visits);
                        ++gathers;
                    }
                    state = W_CTRL;
                }
            }
        }

        if (state == W_TRACE) { // camera ray (depth 0) or TraceSecondaryRay
            Hit h;
            ++pt.closest;
            const bool hit = trace_closest<COUNT, LEVELS, kWhittedCull>(S, ray, st, h, tl);
            if (depth == 0) { fz = h.z; fid = hit ? h.node : -1; }
            if (!hit) {
                if (depth == 0) { // p12 main.cpp:224-226: the background at the sample's un-jittered Halton point
                    const int hk = k % period;
                    const float u = ((float)(pix % F.width) + halton[hk]) / (float)(F.width - 1);
                    const float v = ((float)(pix / F.width) + halton[64 + hk]) / (float)(F.height - 1);
                    ret = textured_color(S, S.background, S.background_texture, mk(u, v, 1.0f));
                } else ret = textured_environment(S, S.environment, S.environment_texture, ray.d);
                retDist = FLT_MAX;
                state = W_RET;
            } else if (h.node < 0) { // a light seen directly is white (p12 main.cpp:147,228)
                ret = mk(1.0f, 1.0f, 1.0f);
                retDist = h.z;
                state = W_RET;
            } else {
                mtl = hit_material(S, h.node, h.face);
                const rt_material &hm = S.materials[mtl];
                finish_hit(S, ray, h, hm.diffuse_texture > 0);
                cur.kd = textured_color(S, hm.diffuse, hm.diffuse_texture, h.uvw);
                cur.p = h.p;
                cur.n = normalize(normalize(h.n)); // SetHit normalises, Shade normalises again
                cur.v = -normalize(ray.d);
                cur.dist = h.z;
                front = h.front;
                cur.final = ld3(hm.emission);
                stage = STG_LIGHT;
                l = 0;
                state = W_CTRL;
            }
        } else if (state == W_SHADOW) {
            ++pt.shadow;
            const bool occluded = trace_shadow<COUNT, LEVELS, kWhittedCull>(S, ray, tmax, st, tl);
            const rt_light L = load_uniform(&S.lights[l]);
            const rt_material &m = S.materials[mtl];
            if (L.type == RT_LIGHT_DIRECT) {
                whitted_light_term(cur, m, ld3(L.intensity) * (occluded ? 0.0f : 1.0f), -ld3(L.direction));
                ++l;
                state = W_CTRL;
            } else {
                fan_hits += occluded ? 0 : 1;
                ++fan_i;
                float n = 0.0f;
                if (fan_i == 4 && (fan_hits == 0 || fan_hits == 4)) n = 4.0f;
                else if (fan_i == 64) n = 64.0f;
                if (n != 0.0f) {
                    const V3 d = ld3(L.position) - cur.p;
                    const V3 li = ((ld3(L.intensity) * (float)fan_hits) / n) / len2(d);
                    whitted_light_term(cur, m, li, normalize(d));
                    ++l;
                    state = W_CTRL;
                } else {
                    ray = whitted_fan_ray(L, cur.p, halton, period, fan_i, offX, offY);
                }
            }
        }

        // bookkeeping between rays: runs until the lane has its next ray (or its sample is finished)
        while (state == W_CTRL || state == W_RET) {
            if (state == W_RET) {
                if (depth == 0) {
                    if (k == 0) {
                        if (out.z) out.z[pix] = fz;
                        if (out.object_id) out.object_id[pix] = fid;
                    }
                    sbuf[item] = make_float4(ret.x, ret.y, ret.z, 0.0f);
                    state = W_IDLE;
                    break;
                }
                --depth;
                cur = stk[depth];
                mtl = (int)(cur.meta & 0xFFFFFu);
                front = (int)((cur.meta >> 20) & 1u);
                const int comb = (int)(cur.meta >> 24);
                const rt_material &m = S.materials[mtl];
                state = W_CTRL;
                if (comb == COMB_MC) {
                    const V3 colorSum = mk(0, 0, 0) + ret * cur.kd;
                    cur.final = cur.final + colorSum / 1.0f;
                    stage = STG_REFL;
                } else if (comb == COMB_REFL) {
                    cur.final = cur.final + ret * ld3(m.reflection);
                    stage = STG_REFR;
                } else if (comb == COMB_TIR) {
                    ret = cur.final + ret * ld3(m.refraction);
                    retDist = cur.dist;
                    state = W_RET;
                } else if (comb == COMB_FRESNEL) {
                    const float F_ = whitted_fresnel(m.ior, dot(cur.v, cur.h));
                    cur.final = cur.final + (ret * ld3(m.refraction)) * F_;
                    stage = STG_TRANS;
                } else { // COMB_TRANS
                    const float F_ = whitted_fresnel(m.ior, dot(cur.v, cur.h));
                    V3 c = ret * ld3(m.refraction);
                    c.x *= expf(-m.absorption[0] * retDist);
                    c.y *= expf(-m.absorption[1] * retDist);
                    c.z *= expf(-m.absorption[2] * retDist);
                    ret = cur.final + c * (1.0f - F_);
                    retDist = cur.dist;
                    state = W_RET;
                }
                continue;
            }
            const rt_material &m = S.materials[mtl];
            V3 child_dir = mk(0, 0, 0);
            int child = -1; // pending combination of the child ray to spawn, -1 = none
            if (stage == STG_LIGHT) {
                if (l >= S.n_lights || (pm & RT_PM_DIRECT)) { stage = STG_AFTER; continue; } // no light loop when the map holds the direct light
                const rt_light L = load_uniform(&S.lights[l]);
                if (L.type == RT_LIGHT_AMBIENT) { // p12 lights.h:33, materials.cpp:28-31
                    cur.final = cur.final + cur.kd * ld3(L.intensity);
                    ++l;
                    continue;
                }
                if (L.type == RT_LIGHT_DIRECT) { // p12 lights.h:49
                    const V3 dir = -ld3(L.direction);
                    ray.p = cur.p + dir * 0.0002f;
                    ray.d = dir;
                    tmax = FLT_MAX;
                } else {
                    offX = rng_float(rng);
                    offY = rng_float(rng);
                    fan_i = 0;
                    fan_hits = 0;
                    ray = whitted_fan_ray(L, cur.p, halton, period, 0, offX, offY);
                    tmax = 1.0f;
                }
                state = W_SHADOW;
                break;
            }
            if (stage == STG_AFTER) {
                // irradiance from the photon maps (p12 materials.cpp:49-70): the lane parks until the wave gathers.
                // The reference also runs the estimate for a specular surface that is no photon surface and drops
                // the result; that lookup is skipped.
                stage = STG_BOUNCE;
                if (pm != 0 && is_photon_surface(m)) { state = W_GATHER; break; }
            }
            if (stage == STG_BOUNCE) {
                if (!(depth < F.max_bounces)) { ret = cur.final; retDist = cur.dist; state = W_RET; continue; } // CanBounce
                stage = STG_MC;
            }
            if (stage == STG_MC) { // Monte-Carlo indirect light on the first two bounces (p12 materials.cpp:74-103)
                stage = STG_REFL;
                if (!(pm & RT_PM_INDIRECT) && (cur.kd.x + cur.kd.y + cur.kd.z) > 0.0f && depth < 2) {
                    const float phiOffset = rng_float(rng);
                    const float thetaOffset = rng_float(rng);
                    const float phi = (float)(kTwoPi * frac_sum(halton[ps_h], phiOffset));
                    const float cosTheta = fsqrt(frac_sum(halton[64 + ps_h], thetaOffset));
                    const float sinTheta = fsqrt(1.0f - cosTheta * cosTheta);
                    V3 u, v;
                    orthonormals(cur.n, u, v);
                    float sp_, cp_;
                    sincos_2pi(phi, sp_, cp_);
                    child_dir = whitted_frame(u, v, cur.n, sinTheta * cp_, sinTheta * sp_, cosTheta);
                    child = COMB_MC;
                }
            } else if (stage == STG_REFL) { // glossy reflection (p12 materials.cpp:106-133)
                stage = STG_REFR;
                if (sum3(m.reflection) > 0.0f) {
                    const float phiOffset = rng_float(rng);
                    const float thetaOffset = rng_float(rng);
                    const float phi = (float)(kTwoPi * frac_sum(halton[ps_h], phiOffset));
                    const float cosTheta = powf(frac_sum(halton[64 + ps_h], thetaOffset), 1.0f / (m.glossiness + 1.0f));
                    const float sinTheta = fsqrt(1.0f - cosTheta * cosTheta);
                    V3 u, v;
                    orthonormals(cur.n, u, v);
                    float sp_, cp_;
                    sincos_2pi(phi, sp_, cp_);
                    const V3 h = whitted_frame(u, v, cur.n, sinTheta * cp_, sinTheta * sp_, cosTheta);
                    const V3 rd = (h * 2.0f) * dot(cur.v, h) - cur.v;
                    if (dot(rd, cur.n) > 0.0f) { child_dir = rd; child = COMB_REFL; }
                }
            } else if (stage == STG_REFR) { // refraction (p12 materials.cpp:136-199)
                if (!(sum3(m.refraction) > 0.0f)) { ret = cur.final; retDist = cur.dist; state = W_RET; continue; }
                const float phiOffset = rng_float(rng);
                const float thetaOffset = rng_float(rng);
                const V3 N = front ? cur.n : -cur.n;
                V3 u, v;
                orthonormals(N, u, v);
                const float phi = (float)(kTwoPi * frac_sum(halton[ps_h], phiOffset));
                const float cosTheta = powf(frac_sum(halton[64 + ps_h], thetaOffset), 1.0f / (m.glossiness + 1.0f));
                const float sinTheta = fsqrt(1.0f - cosTheta * cosTheta);
                float sp_, cp_;
                sincos_2pi(phi, sp_, cp_);
                const V3 h = whitted_frame(u, v, N, sinTheta * cp_, sinTheta * sp_, cosTheta);
                cur.h = h;
                const float viewDotH = dot(cur.v, h);
                const float ratio = front ? 1.0f / m.ior : m.ior;
                const float cos2 = whitted_cos2(ratio, viewDotH);
                const V3 rd = (h * 2.0f) * viewDotH - cur.v;
                if (cos2 < 0.0f) { // total internal reflection
                    if (dot(rd, N) <= 0.0f) { ret = cur.final; retDist = cur.dist; state = W_RET; continue; }
                    child_dir = rd; child = COMB_TIR;
                } else if (dot(rd, N) > 0.0f) {
                    child_dir = rd; child = COMB_FRESNEL;
                } else {
                    cur.final = cur.final + mk(0, 0, 0) * whitted_fresnel(m.ior, viewDotH);
                    stage = STG_TRANS;
                }
            }
            if (child < 0 && stage == STG_TRANS) { // the refracted ray (p12 materials.cpp:185-196)
                const float viewDotH = dot(cur.v, cur.h);
                const float ratio = front ? 1.0f / m.ior : m.ior;
                const float cos2 = whitted_cos2(ratio, viewDotH);
                child_dir = cur.v * (-ratio) - cur.h * (fsqrt(cos2) - ratio * viewDotH);
                child = COMB_TRANS;
            }
            if (child >= 0) {
                cur.meta = (uint32_t)mtl | ((uint32_t)front << 20) | ((uint32_t)child << 24);
                stk[depth] = cur;
                ++depth;
                ray.p = cur.p + child_dir * 0.0002f; // p12 main.cpp:136
                ray.d = child_dir;
                state = W_TRACE;
            }
        }
    }
    flush_counters<COUNT>(ctr, pt, tl, paths);
    if (pm != 0) {
        const uint64_t g = wave_sum(gathers), v = wave_sum(visits);
        if (lane == 0) {
            atomicAdd(&ctr->gathers, (unsigned long long)g);
            if (COUNT) atomicAdd(&ctr->photon_visits, (unsigned long long)v);
        }
    }
}

} // namespace rtd
