// Known-answer queries for the samplers and the random streams the kernels use (tests): the device functions the
// path tracer, the Whitted-style shader and the photon emission call, evaluated one call per lane on the caller's inputs.
#pragma once
#include "photon.h"
#include "rt_mi355x.h"

namespace rtd {

__global__ __launch_bounds__(kBlock) void k_debug_samplers(DScene S, int which, const rt_sampler_query *__restrict__ q, int n,
                                                           rt_sampler_answer *__restrict__ out)
{
    const int i = (int)(blockIdx.x * kBlock + threadIdx.x);
    if (i >= n) return;
    const rt_sampler_query in = q[i];
    rt_sampler_answer a;
    memset(&a, 0, sizeof a);
    Rng rng;
    rng_set_sequence(rng, in.seq, in.seed);
    Surface s;
    s.p = mk(in.p[0], in.p[1], in.p[2]); s.n = mk(in.n[0], in.n[1], in.n[2]); s.v = mk(in.v[0], in.v[1], in.v[2]);
    s.front = in.front;
    const V3 dir = mk(in.dir[0], in.dir[1], in.dir[2]);
    const rt_light &L = S.lights[0];
    const rt_material &M = S.materials[in.material];
    Sample o;
    o.dir = o.mult = mk(0, 0, 0); o.prob = 0; o.dist = 0; o.lobe = 0;
    if (which == RT_SAMPLER_LIGHT) { // PointLight::GenerateSample / GetSampleInfo (src/lights.h:208-288)
        light_generate(L, s.p, rng, o);
        a.ok = 1;
        a.pdf = light_pdf(L, s.p, dir);
    } else if (which == RT_SAMPLER_MATERIAL) { // MtlBlinn::GenerateSample / GetSampleInfo (src/materials.h:87-303)
        a.ok = mtl_generate(M, s, rng, o) ? 1 : 0;
        a.pdf = mtl_pdf(M, s, dir);
    } else if (which == RT_SAMPLER_PHOTON_MATERIAL) { // the snapshot's MtlBlinn::GenerateSample (photon scattering)
        a.ok = p12_mtl_generate(M, s.n, s.v, s.front, rng, o.dir, o.mult, o.prob, o.lobe) ? 1 : 0;
    } else { // PointLight::RandomPhoton (src/lights.h:92-124)
        Ray r;
        light_random_photon(L, rng, r, o.mult);
        o.dir = r.d;
        a.origin[0] = r.p.x; a.origin[1] = r.p.y; a.origin[2] = r.p.z;
        a.ok = 1;
    }
    a.dir[0] = o.dir.x; a.dir[1] = o.dir.y; a.dir[2] = o.dir.z;
    a.mult[0] = o.mult.x; a.mult[1] = o.mult.y; a.mult[2] = o.mult.z;
    a.prob = o.prob; a.dist = o.dist; a.lobe = o.lobe;
    a.next_u32 = rng_u32(rng); // tells how many numbers the call drew
    out[i] = a;
}

// n_draws numbers of each of n_streams PCG streams (seq, seed), as raw words and, from the stream seeded again, as floats
__global__ __launch_bounds__(kBlock) void k_debug_rng(const uint64_t *__restrict__ seq, const uint64_t *__restrict__ seed, int n_streams,
                                                      int n_draws, uint32_t *__restrict__ words, float *__restrict__ floats)
{
    const int i = (int)(blockIdx.x * kBlock + threadIdx.x);
    if (i >= n_streams) return;
    Rng rng;
    rng_set_sequence(rng, seq[i], seed[i]);
    for (int k = 0; k < n_draws; ++k) words[(size_t)i * n_draws + k] = rng_u32(rng);
    rng_set_sequence(rng, seq[i], seed[i]);
    for (int k = 0; k < n_draws; ++k) floats[(size_t)i * n_draws + k] = rng_float(rng);
}

} // namespace rtd
