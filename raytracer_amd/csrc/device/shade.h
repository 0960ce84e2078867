// Light and material samplers and the path integrator: the device form of
// PointLight::GenerateSample/GetSampleInfo (src/lights.h:208-288),
// MtlBlinn::GenerateSample/GetSampleInfo (src/materials.h:87-303) and
// tracePath (src/main.cpp:160-268).  The order of random draws is part of the
// contract: light r1, light phi, lobe, then two (or three) draws of the lobe.
#pragma once
#include "texture.h"
#include "trace.h"

#ifdef RT_ABLATE_NATIVE_MATH // timing experiment only: hardware approximations, results change
#define sinf __sinf
#define cosf __cosf
#define powf __powf
#endif

namespace rtd {

constexpr float kPi = (float)3.141592653589793238462643383279502884197169; // cyCore.h:262

enum { LOBE_NONE = 0, LOBE_DIFFUSE = 1, LOBE_SPECULAR = 2, LOBE_TRANSMISSION = 4 };

struct Surface {
    V3 p, n, v; // position, unit shading normal, unit view vector
    int front;
};
struct Sample {
    V3 dir, mult;
    float prob, dist;
    int lobe;
};

RT_DEV V3 light_radiance(const rt_light &l) { return ld3(l.intensity) / ((kPi)*l.size * l.size); } // lights.h:86

// src/lights.h:208-242
RT_DEV void light_generate(const rt_light &l, V3 P, Rng &rng, Sample &o)
{
    V3 toC = ld3(l.position) - P;
    const float dist = len(toC);
    toC = normalize(toC);
    const float sinMax = l.size / dist;
    const float cosMax = fsqrt(1.0f - (sinMax * sinMax));
    const float r1 = rng_float(rng);
    const float cosT = 1.0f - r1 + r1 * cosMax;
    const float sinT = fsqrt(1.0f - (cosT * cosT));
    const float phi = 2.0f * kPi * rng_float(rng);
    float sp, cp;
    sincos_2pi(phi, sp, cp);
    const float x = sinT * cp;
    const float y = sinT * sp;
    const float z = cosT;
    V3 u, v;
    orthonormals(toC, u, v);
    const float hyp = dist;
    const float adj = hyp * cosT;
    const float oppSq = (hyp * hyp) - (adj * adj);
    const float inside = (l.size * l.size) - oppSq;
    const float tOff = fsqrt(std_max(0.0f, inside));
    o.dist = adj - tOff;
    o.dir = normalize((u * x + v * y) + toC * z);
    o.mult = light_radiance(l);
    const float omc = (sinMax * sinMax) / (1.0f + cosMax);
    o.prob = 1.0f / (2.0f * kPi * omc);
}

// src/lights.h:243-288
RT_DEV float light_pdf(const rt_light &l, V3 P, V3 dir)
{
    V3 toC = ld3(l.position) - P;
    const float dist = len(toC);
    const float dist2 = len2(toC);
    if ((double)dist < 1e-5) return 0.0f;
    toC = normalize(toC);
    const float sinMaxSq = (l.size * l.size) / dist2;
    const float cosRay = dot(dir, toC);
    if (cosRay <= 0.0f) return 0.0f;
    const float cosRaySq = cosRay * cosRay;
    const float cosMaxSq = 1.0f - sinMaxSq;
    if (cosRaySq >= cosMaxSq) {
        const float cosMax = fsqrt(std_max(0.0f, cosMaxSq));
        const float omc = sinMaxSq / (1.0f + cosMax);
        return 1.0f / (2.0f * kPi * omc);
    }
    return 0.0f;
}

RT_DEV float gray(const float *c) { return (c[0] + c[1] + c[2]) / 3.0f; } // cyColor.h:86-87

RT_DEV void lobe_probs(const rt_material &m, float &pd, float &ps, float &pt) // materials.h:93-114
{
    pd = gray(m.diffuse); ps = gray(m.specular); pt = gray(m.refraction);
    const float total = pd + ps + pt;
    if (total > 1.0f) { pd /= total; ps /= total; pt /= total; }
}

// src/materials.h:87-264
RT_DEV bool mtl_generate(const rt_material &m, const Surface &s, Rng &rng, Sample &o)
{
    const V3 kd = ld3(m.diffuse), ks = ld3(m.specular), kt = ld3(m.refraction);
    float pd, ps, pt;
    lobe_probs(m, pd, ps, pt);
    const V3 Ns = s.n, V = s.v;
    o.dist = 0.0f;
    const float rnd = rng_float(rng);
    if (rnd < pd) {
        o.lobe = LOBE_DIFFUSE;
        const float r = fsqrt(rng_float(rng));
        const float theta = 2.0f * kPi * rng_float(rng);
        float st_, ct_;
        sincos_2pi(theta, st_, ct_);
        const float x = r * ct_;
        const float y = r * st_;
        const float z = fsqrt(std_max(0.0f, 1.0f - x * x - y * y));
        V3 u, v;
        orthonormals(Ns, u, v);
        const V3 dir = (u * x + v * y) + Ns * z;
        const float g = std_max(0.0f, dot(Ns, dir));
        o.dir = dir;
        o.mult = (kd * g) / kPi;
        o.prob = (g / kPi) * pd;
        return true;
    }
    if (rnd < pd + ps) {
        o.lobe = LOBE_SPECULAR;
        const float alpha = m.glossiness;
        const float phi = 2.0f * kPi * rng_float(rng);
        const float cosT = powf(1.0f - rng_float(rng), 1.0f / (alpha + 1.0f));
        const float sinT = fsqrt(1.0f - cosT * cosT);
        float sp, cp;
        sincos_2pi(phi, sp, cp);
        const float x = sinT * cp;
        const float y = sinT * sp;
        V3 u, v;
        orthonormals(Ns, u, v);
        const V3 h = (u * x + v * y) + Ns * cosT;
        const float vh = dot(V, h);
        o.dir = normalize((h * 2.0f) * std_max(0.0f, vh) - V);
        const float cos_nh = dot(Ns, h);
        if (cos_nh < 0.0f) return false;
        const float half_pdf = ((alpha + 1.0f) / (8.0f * kPi)) * powf(cosT, alpha);
        o.prob = (half_pdf / (4.0f * vh)) * ps;
        const float lobe_scale = (alpha + 2.0f) / (8.0f * kPi);
        o.mult = ((ks * lobe_scale) * powf(cos_nh, alpha)) / (4.0f * vh);
        return true;
    }
    if (rnd < pd + ps + pt) {
        o.lobe = LOBE_TRANSMISSION;
        const float schlick = powf((1.0f - m.ior) / (1.0f + m.ior), 2.0f);
        V3 N = Ns;
        float n_from = 1.0f, n_into = m.ior;
        if (!s.front) { N = -N; const float t = n_from; n_from = n_into; n_into = t; }
        const float eta = n_from / n_into;
        const float alpha = m.glossiness;
        const float phi = 2.0f * kPi * rng_float(rng);
        const float cosT = powf(1.0f - rng_float(rng), 1.0f / (alpha + 1.0f));
        const float sinT = fsqrt(1.0f - cosT * cosT);
        float sp, cp;
        sincos_2pi(phi, sp, cp);
        const float x = sinT * cp;
        const float y = sinT * sp;
        V3 u, v;
        orthonormals(N, u, v);
        const V3 h = (u * x + v * y) + N * cosT;
        const float cos_vh = dot(V, h);
        const float k = 1.0f - eta * eta * (1.0f - cos_vh * cos_vh);
        o.mult = kt;
        o.prob = pt;
        const V3 refl = (h * 2.0f) * std_max(0.0f, cos_vh) - V;
        if (k < 0.0f) { o.dir = refl; return true; }
        const V3 dir = h * (eta * cos_vh - fsqrt(k)) - V * eta;
        o.dir = dir;
        if (fabsf(dot(N, dir)) < 1e-5f) return false;
        const float F = schlick + (1.0f - schlick) * powf(1.0f - cos_vh, 5.0f);
        const float tf = 1.0f - F;
        if (rng_float(rng) > tf) o.dir = refl;
        return true;
    }
    return false;
}

// src/materials.h:266-303
RT_DEV float mtl_pdf(const rt_material &m, const Surface &s, V3 dir)
{
    float pd, ps, pt;
    lobe_probs(m, pd, ps, pt);
    const float nDotDir = dot(s.n, dir);
    const bool isRefl = nDotDir > 0.0f;
    float prob = 0.0f;
    if (pd > 0.0f && isRefl) prob += pd * (nDotDir / kPi);
    const V3 h = normalize(s.v + dir);
    const float cos_nh = dot(s.n, h);
    const float cos_vh = dot(s.v, h);
    if (isRefl && cos_nh > 0.0f && cos_vh > 0.0f && ps > 0.0f) {
        const float alpha = m.glossiness;
        const float lobe_scale = (alpha + 2.0f) / (8.0f * kPi);
        const float half_pdf = lobe_scale * powf(cos_nh, alpha);
        prob += ps * (half_pdf / (4.0f * cos_vh));
    }
    return prob;
}

struct PathTally {
    uint32_t closest = 0, shadow = 0;
#ifdef RT_PROFILE_PHASES
    unsigned long long ph[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t0 = 0;
    RT_DEV void tick() { t0 = wall_clock64(); }
    RT_DEV void tock(int i) { const unsigned long long t = wall_clock64(); ph[i] += t - t0; t0 = t; }
    // one execution of phase i with `lanes` active lanes, ended now
    RT_DEV void ran(int i, uint64_t lanes) { tock(i); ph[4 + i] += 1; ph[8 + i] += (unsigned long long)__popcll(lanes); }
#else
    RT_DEV void tick() {}
    RT_DEV void tock(int) {}
    RT_DEV void ran(int, uint64_t) {}
#endif
};

// One camera path in flight.  A lane keeps this across bounces and swaps in a
// fresh one when the path ends, so a wave stays full however uneven path lengths are.
struct PathState {
    Ray ray;
    V3 throughput, result;
    Rng rng;
    float lastProb;
    int lastLobe;
    int bounce;
};

RT_DEV void path_begin(PathState &ps, Ray ray)
{
    ps.ray = ray;
    ps.throughput = mk(1.0f, 1.0f, 1.0f);
    ps.result = mk(0.0f, 0.0f, 0.0f);
    ps.lastProb = 1.0f;
    ps.lastLobe = LOBE_NONE;
    ps.bounce = 0;
}

// One iteration of the bounce loop of tracePath (src/main.cpp:170-265).  Returns
// false when the path has ended; ps.result then holds its radiance.  first_z /
// first_id receive depth and node id of the camera ray's hit when non-null.
// CULL: the scene walks skip objects whose world-space box no ray of the wave reaches (the hierarchy over the boxes
// where the scene has one): which objects are looked at changes, no answer does.
template <bool COUNT, int LEVELS, bool TEXBG = true, bool CULL = false>
RT_DEV bool path_step(const DScene &S, const DFrame &F, PathState &ps, LaneStack st, PathTally &pt,
                      Tally<COUNT> &tl, float *first_z, int *first_id)
{
    if (ps.bounce >= F.max_bounces) return false;
    tl.step(6);
    const rt_light light = load_uniform(&S.lights[0]);
    const float bias = F.bias;
    const Ray ray = ps.ray;
    Hit h;
    bool hit;
    pt.tick();
    ++pt.closest;
    hit = trace_closest<COUNT, LEVELS, CULL>(S, ray, st, h, tl);
    pt.tock(0);
    if (ps.bounce == 0 && first_z) { *first_z = h.z; *first_id = hit ? h.node : -1; }
    if (!hit) {
        // main.cpp:175-176: background.Eval(ray.dir).  The texture lookup is compiled in only for scenes that have a
        // background texture: present but never taken, it still cost the 1M-triangle frame 11 % (register allocation)
        const V3 bg = TEXBG ? textured_color(S, S.background, S.background_texture, ray.d) : ld3(S.background);
        ps.result = ps.result + bg * ps.throughput;
        return false;
    }
    if (h.node < 0) { // light (main.cpp:184-210)
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
        return false;
    }
    tl.step(7);
    finish_hit(S, ray, h, false);
    Surface s; // SamplerInfo::SetHit (renderer.h:145-153)
    const V3 normal = normalize(h.n);
    s.p = h.p; s.n = normal; s.v = -normalize(ray.d); s.front = h.front;
    const rt_material &mtl = S.materials[hit_material(S, h.node, h.face)];

    // next-event estimation (main.cpp:214-251)
    Sample ne;
    light_generate(light, s.p, ps.rng, ne);
    {
        const float sign = h.front ? 1.0f : -1.0f;
        Ray sr;
        sr.p = h.p + (normal * bias) * sign;
        sr.d = ne.dir;
        ++pt.shadow;
        pt.tock(1);
        const bool occluded = trace_shadow<COUNT, LEVELS, CULL>(S, sr, ne.dist - bias, st, tl);
        pt.tock(2);
        if (!occluded) {
            const float cosS = std_max(0.0f, dot(normal, sr.d));
            if (cosS > 0.0f && ne.prob > 0.0f) {
                const float mp = mtl_pdf(mtl, s, ne.dir);
                float weight = 1.0f;
                if (mp > 0.0f) weight = (ne.prob * ne.prob) / (ne.prob * ne.prob + mp * mp);
                // the specular term of main.cpp:241-246 is gated on a field that
                // GetSampleInfo never writes (it reads as NONE): diffuse only
                const V3 brdf = ld3(mtl.diffuse) / kPi;
                ps.result = ps.result + ((((brdf * cosS) * ne.mult) * weight) / ne.prob) * ps.throughput;
            }
        }
    }

    // indirect bounce (main.cpp:253-264)
    Sample in;
    if (!mtl_generate(mtl, s, ps.rng, in)) return false;
    ps.lastProb = in.prob;
    ps.lastLobe = in.lobe;
    ps.ray.d = in.dir;
    const float sign = (dot(normal, in.dir) > 0.0f) ? 1.0f : -1.0f;
    ps.ray.p = h.p + (normal * bias) * sign;
    ps.throughput = ps.throughput * (in.mult / in.prob);
    ++ps.bounce;
    pt.tock(3);
    return true;
}

} // namespace rtd
