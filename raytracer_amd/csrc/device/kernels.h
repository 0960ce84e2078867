// gfx950 kernels of the rendering hot path.
//
//   k_paths    persistent kernel, one lane = one (pixel, sample) camera path at a
//              time: seeds its PCG stream, builds the camera ray
//              (src/main.cpp:294-323), runs the path integrator one bounce per
//              loop trip and writes the sample's radiance to the sample buffer.
//              A lane whose path has ended takes the next path from a work queue,
//              so a wave stays full although path lengths vary from 1 to 50 bounces.
//   k_resolve  one lane = one pixel: in-order running sums over the new samples,
//              the variance stop test (src/main.cpp:357-372), and for finished
//              pixels the mean, sRGB and 8-bit quantisation (src/main.cpp:375-380)
//   k_trace*   batch forms of TraceRay / TraceShadowRay / tracePath for arrays of rays
#pragma once
#include "shade.h"
#include <limits.h>

namespace rtd {

constexpr int kBlock = 256;

struct FramePlanes {
    uint8_t *rgb;
    float *z;
    int32_t *sample_count;
    int32_t *object_id;
    float *linear;
};

// per-pixel running state between sampling rounds
struct PixelState {
    float sum[3];
    float sum_sq[3];
    int32_t n;
    int32_t pad;
};

static_assert(kBlock == (int)kStackStride, "stack[level][thread] layout assumes one block per stride");
RT_DEV LaneStack lane_stack(uint32_t *lds, const DScene &S)
{
    LaneStack st;
    st.lds = (LdsStackPtr)(u32x2 *)lds + threadIdx.x;
    st.ovf = (GlobalStackPtr)(u32x2 *)S.stack_overflow + ((size_t)blockIdx.x * kBlock + threadIdx.x);
    st.lds_levels = (uint32_t)S.lds_levels;
    st.ovf_stride = gridDim.x * kBlock;
    return st;
}

RT_DEV uint64_t wave_sum(uint64_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down((unsigned long long)v, o, 64);
    return v;
}

template <bool COUNT>
RT_DEV void flush_counters(DCounters *ctr, const PathTally &pt, const Tally<COUNT> &tl, uint32_t paths)
{
    const uint64_t c = wave_sum(pt.closest), s = wave_sum(pt.shadow), p = wave_sum(paths);
    uint64_t b = 0, t = 0;
    if (COUNT) { b = wave_sum(tl.box); t = wave_sum(tl.tri); }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&ctr->closest, (unsigned long long)c);
        atomicAdd(&ctr->shadow, (unsigned long long)s);
        atomicAdd(&ctr->paths, (unsigned long long)p);
        if (COUNT) { atomicAdd(&ctr->box_tests, (unsigned long long)b); atomicAdd(&ctr->tri_tests, (unsigned long long)t); }
#ifdef RT_PROFILE_PHASES
        for (int i = 0; i < 16; ++i) atomicAdd(&ctr->phase[i], pt.ph[i]); // lane 0's clocks stand for the wave
#endif
    }
#ifdef RT_CULL_CHECK
    { const uint64_t cv = wave_sum(tl.cull_violations); if ((threadIdx.x & 63) == 0) atomicAdd(&ctr->phase[11], (unsigned long long)cv); }
#endif
#ifdef RT_COUNT_OBJECTS
    { const uint64_t ob = wave_sum(tl.objects); if ((threadIdx.x & 63) == 0) atomicAdd(&ctr->phase[10], (unsigned long long)ob); }
#endif
#ifdef RT_PROFILE_LANES
    for (int i = 0; i < 8; ++i) {
        const uint64_t wv = wave_sum(tl.wave[i]), ln = wave_sum(tl.lanes[i]);
        if ((threadIdx.x & 63) == 0) { atomicAdd(&ctr->phase[i], (unsigned long long)wv); atomicAdd(&ctr->phase[8 + i], (unsigned long long)ln); }
    }
#endif
}

RT_DEV uint64_t pixel_seed(uint64_t frame, uint64_t pix) { return frame + pix * 0x9E3779B97F4A7C15ULL; }

// fmod(a + b, 1) for a, b in [0,1): exact in binary32
RT_DEV float frac_sum(float a, float b)
{
    const float s = a + b;
    return s >= 1.0f ? s - 1.0f : s;
}

// camera ray of sample k of pixel (i, j) (src/main.cpp:303-323)
RT_DEV Ray camera_ray(const DFrame &F, const float *halton, int i, int j, int k, const float off[4])
{
    const int hk = k % F.halton_period;
    const float jx = frac_sum(halton[hk], off[0]);
    const float jy = frac_sum(halton[64 + hk], off[1]);
    const float px = (float)i + jx;
    const float py = (float)j + jy;
    const float sx = -F.half_w + F.pixel_size * px;
    const float sy = F.half_h - F.pixel_size * py;
    const V3 cpos = ld3(F.cam_pos);
    const V3 dest = cpos + m3_mul(F.cam2world, mk(sx, sy, -F.focaldist));
    const float jt = frac_sum(halton[128 + hk], off[2]);
    const float jr = frac_sum(halton[192 + hk], off[3]);
    const float theta = (float)((double)(jt * 2.0f) * M_PI); // float*float, then *double (main.cpp:313)
    const float radius = fsqrt(jr);
    float st_, ct_;
    sincos_2pi(theta, st_, ct_);
    const V3 lens = mk(radius * F.dof * ct_, radius * F.dof * st_, 0.0f);
    Ray r;
    r.p = cpos + m3_mul(F.cam2world, lens);
    r.d = dest - r.p;
    return r;
}

// Work queue of a launch: items [0, n_items) are handed out in chunks; a wave draws a
// chunk with one atomic and feeds its lanes from it, so a queue word sees one atomic per
// kChunk paths.  The range can be cut into RT_QUEUES contiguous parts with a queue word
// each (on its own cache line): workgroups whose blockIdx agree modulo 8 share an XCD and
// would start on the same part, keeping each L2 on one region of the image, and a wave
// whose part is used up goes on to the next.  Measured on the 1M-triangle frame this
// was no faster than one queue (all XCDs working through the same few tiles, 2.53 vs
// 2.46 Grays/s), so one queue is the default.
#ifndef RT_QUEUES
#define RT_QUEUES 1
#endif
#ifndef RT_CHUNK
#define RT_CHUNK 512
#endif
constexpr uint32_t kChunk = RT_CHUNK;
constexpr uint32_t kQueues = RT_QUEUES;
constexpr uint32_t kQueueStride = 16; // unsigned long longs between queue words (128 bytes)
constexpr size_t kQueueBytes = kQueues * kQueueStride * sizeof(unsigned long long);

struct WorkQueue {
    unsigned long long *words;
    uint64_t n_items, part; // part: items per queue, a multiple of kChunk
    uint32_t q, tried;      // current part, parts found empty so far
    uint32_t draws = 0;     // draws of this wave so far
    uint64_t next = 0, end = 0;
    bool exhausted = false;
    uint32_t chunk; // items per draw: large chunks keep a wave on neighbouring pixels (coherent walks), small ones even
                    // out the end of the launch; the host picks it from the items each wave will get (DFrame::chunk)
    const uint32_t *cancel;              // flag of rt_cancel (pinned host memory)
    unsigned long long *progress = nullptr; // where to report the queue position for rt_progress (pinned host memory), or null
    RT_DEV WorkQueue(unsigned long long *w, uint64_t n, uint32_t chunk_items, const uint32_t *cancel_flag, unsigned long long *progress_word = nullptr)
        : words(w), n_items(n), chunk(chunk_items), cancel(cancel_flag), progress(progress_word)
    {
        const uint64_t chunks = (n + chunk - 1) / chunk;
        part = ((chunks + kQueues - 1) / kQueues) * chunk;
        q = blockIdx.x % kQueues;
        tried = 0;
    }
    // wave-uniform: make [next, end) non-empty if any work is left anywhere
    RT_DEV void refill(uint32_t lane)
    {
        while (next >= end && !exhausted) {
            // the flag lives in host memory (a read costs microseconds): a wave looks at it at every eighth draw
            if ((draws++ & 7u) == 0u && __hip_atomic_load(cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) { exhausted = true; break; }
            unsigned long long b = 0;
            if (lane == 0) {
                b = atomicAdd(&words[q * kQueueStride], (unsigned long long)chunk);
                if (kQueues == 1 && progress) __hip_atomic_store(progress, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); // what had been handed out before this draw
            }
            b = __shfl(b, 0, 64);
            const uint64_t lo = (uint64_t)q * part, hi = lo + part < n_items ? lo + part : n_items;
            const uint64_t s = lo + b;
            if (s < hi) { next = s; end = s + chunk < hi ? s + chunk : hi; }
            else if (++tried >= kQueues) exhausted = true;
            else q = (q + 1) % kQueues;
        }
    }
};

#ifndef RT_PATHS_MIN_WAVES
#define RT_PATHS_MIN_WAVES 3 // waves per SIMD the register allocator must leave room for: 170 registers.  The kernel runs the scenes without meshes (a scene with one goes through the wavefront pipeline), where it is bound by its own arithmetic: at 6 waves (80 registers) it spills 220 values and the reference's scene takes 65 ms, at 4: 56 ms, at 3: 38 ms, at 2: 46 ms
#endif

template <bool COUNT, int LEVELS, bool TEXBG, bool CULL = false>
__global__ __launch_bounds__(kBlock, RT_PATHS_MIN_WAVES) void k_paths(DScene S, DFrame F, const float *__restrict__ halton,
                                                  const int32_t *__restrict__ active, uint64_t n_items, int k0, int nk,
                                                  float4 *__restrict__ sbuf, FramePlanes out, DCounters *ctr,
                                                  unsigned long long *queue)
{
    extern __shared__ uint32_t lds[];
    const LaneStack st = lane_stack(lds, S);
    const uint32_t lane = threadIdx.x & 63u;
    PathTally pt;
    Tally<COUNT> tl;
    uint32_t paths = 0;

#ifdef RT_PROFILE_TAIL
    const unsigned long long tail_t0 = wall_clock64();
#endif
    WorkQueue wq(queue, n_items, F.chunk, S.cancel, S.progress); // wave-uniform
    bool alive = false;
    uint64_t item = 0;
    int pix = 0, k = 0;
    float fz = FLT_MAX; int fid = -1;
    PathState ps;
    path_begin(ps, Ray{mk(0, 0, 0), mk(0, 0, 1)});
    ps.rng.state = 0; ps.rng.inc = 1;

    for (;;) {
        pt.tick();
        const uint64_t need = __ballot(!alive);
        if (need != 0) {
            wq.refill(lane);
            const uint64_t avail = wq.end - wq.next;
            const uint32_t want = (uint32_t)__popcll(need);
            const uint32_t take = avail < want ? (uint32_t)avail : want;
            const uint32_t rank = (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
            if (!alive && rank < take) {
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
                fz = FLT_MAX; fid = -1;
                alive = true;
                ++paths;
            }
            wq.next += take;
        }
        if (!__any(alive)) {
            if (wq.exhausted) break; // every wave gets here: the queue words only grow
            continue;
        }
        pt.tock(4);
        if (alive) {
            if (!path_step<COUNT, LEVELS, TEXBG, CULL>(S, F, ps, st, pt, tl, &fz, &fid)) {
                if (k == 0) {
                    if (out.z) out.z[pix] = fz;
                    if (out.object_id) out.object_id[pix] = fid;
                }
                sbuf[item] = make_float4(ps.result.x, ps.result.y, ps.result.z, 0.0f);
                alive = false;
            }
        }
    }
    flush_counters<COUNT>(ctr, pt, tl, paths);
#ifdef RT_PROFILE_TAIL
    // diagnostic build: when does each wave run out of work?  phase[12] = earliest wave start, [13] = latest wave end,
    // [14] = sum of wave ends, [15] = waves (100 MHz clock); the mean of (latest end - wave end) is the launch's tail
    if (lane == 0) {
        const unsigned long long t1 = wall_clock64();
        atomicMin(&ctr->phase[12], tail_t0);
        atomicMax(&ctr->phase[13], t1);
        atomicAdd(&ctr->phase[14], t1 - tail_t0);
        atomicAdd(&ctr->phase[15], 1ull);
    }
#endif
}

// Zero bounces: tracePath's loop body never runs, so every sample is black; only the depth / id planes need the
// first sample's camera ray.  A kernel of its own, because the same code as a never-taken branch of k_paths cost
// the 1M-triangle frame 7 % (a second inlined copy of the scene walk in the register allocator's way).
// the per-face shading records of a mesh (DMesh::face_rec) from its index and normal arrays
__global__ __launch_bounds__(kBlock) void k_face_records(const uint32_t *__restrict__ fn, const float *__restrict__ vn,
                                                         const uint32_t *__restrict__ fmtl, uint32_t n_faces, float4 *__restrict__ out)
{
    for (uint32_t f = blockIdx.x * kBlock + threadIdx.x; f < n_faces; f += gridDim.x * kBlock) {
        const float *a = vn + (size_t)fn[f * 3] * 3, *b = vn + (size_t)fn[f * 3 + 1] * 3, *c = vn + (size_t)fn[f * 3 + 2] * 3;
        float4 *rec = out + (size_t)f * 4;
        rec[0] = make_float4(a[0], a[1], a[2], b[0]);
        rec[1] = make_float4(b[1], b[2], c[0], c[1]);
        rec[2] = make_float4(c[2], __uint_as_float(fmtl ? fmtl[f] : 0u), 0.0f, 0.0f);
        rec[3] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
}

template <int LEVELS>
__global__ __launch_bounds__(kBlock) void k_probe(DScene S, DFrame F, const float *__restrict__ halton,
                                                  const int32_t *__restrict__ active, uint64_t n_items, int k0, int nk,
                                                  float4 *__restrict__ sbuf, FramePlanes out, DCounters *ctr)
{
    extern __shared__ uint32_t lds[];
    const LaneStack st = lane_stack(lds, S);
    uint32_t paths = 0;
    for (uint64_t item = (uint64_t)blockIdx.x * kBlock + threadIdx.x; item < n_items; item += (uint64_t)gridDim.x * kBlock) {
        const uint32_t a = (uint32_t)(item / (uint32_t)nk);
        const int k = k0 + (int)(item - (uint64_t)a * (uint32_t)nk);
        ++paths;
        sbuf[item] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (k != 0) continue;
        const int pix = active[a];
        const uint64_t seed = pixel_seed(F.seed, (uint64_t)pix);
        Rng rng;
        rng_set_sequence(rng, 0, seed);
        float off[4];
        off[0] = rng_float(rng); off[1] = rng_float(rng); off[2] = rng_float(rng); off[3] = rng_float(rng);
        const Ray ray = camera_ray(F, halton, pix % F.width, pix / F.width, 0, off);
        Hit h;
        Tally<false> none;
        const bool hit = trace_closest<false, LEVELS>(S, ray, st, h, none);
        if (out.z) out.z[pix] = h.z;
        if (out.object_id) out.object_id[pix] = hit ? h.node : -1;
    }
    const uint64_t p = wave_sum(paths);
    if ((threadIdx.x & 63) == 0) atomicAdd(&ctr->paths, (unsigned long long)p);
}

// Scheduling keys: what the first sample's camera ray costs (box and triangle tests; 0 for a ray that misses
// everything).  Pixels whose camera ray is expensive tend to have expensive paths (long walks over grazing
// geometry, glossy or glass surfaces), and a launch ends sooner if they start first: the host sorts the pixel
// list by this key, descending, before the first sampling round.  The order decides only when a pixel is
// rendered, never what it looks like.
template <int LEVELS>
__global__ __launch_bounds__(kBlock) void k_order_keys(DScene S, DFrame F, const float *__restrict__ halton,
                                                       const int32_t *__restrict__ active, uint32_t n, uint32_t *__restrict__ keys)
{
    extern __shared__ uint32_t lds[];
    const LaneStack st = lane_stack(lds, S);
    for (uint32_t a = blockIdx.x * kBlock + threadIdx.x; a < n; a += gridDim.x * kBlock) {
        const int pix = active[a];
        const uint64_t seed = pixel_seed(F.seed, (uint64_t)pix);
        Rng rng;
        rng_set_sequence(rng, 0, seed);
        float off[4];
        off[0] = rng_float(rng); off[1] = rng_float(rng); off[2] = rng_float(rng); off[3] = rng_float(rng);
        const Ray ray = camera_ray(F, halton, pix % F.width, pix / F.width, 0, off);
        Hit h;
        Tally<true> tl;
        const bool hit = trace_closest<true, LEVELS>(S, ray, st, h, tl);
        const rt_material &m = S.materials[hit && h.node >= 0 ? hit_material(S, h.node, h.face) : 0];
        // a surface that keeps a path going (glossy, glass) weighs more than one that mostly ends it
        const float keep = hit && h.node >= 0 ? gray(m.specular) + gray(m.refraction) : 0.0f;
        const uint32_t walk = tl.box + 4u * tl.tri;
        // 24 bits: how much of a path the surface keeps going first, the cost of the walk second
        const uint32_t k_keep = (uint32_t)((keep < 1.0f ? keep : 1.0f) * 255.0f);
        keys[a] = hit ? (k_keep << 16) | (walk < 65535u ? walk + 1u : 65535u) : 0u;
    }
}

// int(float) as x86 cvttss2si does it: NaN and out-of-range give INT_MIN
RT_DEV int x86_f2i(float f)
{
    if (!(f > -2147483904.0f && f < 2147483648.0f)) return INT_MIN;
    return (int)f;
}
RT_DEV uint8_t to_byte(float r) // cyColor.h:324-325
{
    const int v = x86_f2i(r * 255 + 0.5f);
    return v < 0 ? 0 : (v > 255 ? 255 : (uint8_t)v);
}
RT_DEV float lin2srgb(float cl) // cyColor.h:98
{
    return cl < 0.0031308f ? cl * 12.92f : powf(cl, 0.41666f) * 1.055f - 0.055f;
}

__global__ __launch_bounds__(kBlock) void k_resolve(DFrame F, const int32_t *__restrict__ active, uint32_t n_active,
                                                    int k0, int nk, const float4 *__restrict__ sbuf,
                                                    PixelState *__restrict__ state, FramePlanes out,
                                                    int32_t *__restrict__ next_active, uint32_t *next_count)
{
    const uint32_t a = blockIdx.x * kBlock + threadIdx.x;
    if (a >= n_active) return;
    const int pix = active[a];
    PixelState s;
    if (k0 == 0) { s.sum[0] = s.sum[1] = s.sum[2] = 0.0f; s.sum_sq[0] = s.sum_sq[1] = s.sum_sq[2] = 0.0f; s.n = 0; s.pad = 0; }
    else s = state[pix];
    bool done = false;
    // A pixel's samples are consecutive (16 bytes each): a lane takes them eight at a time, one 128-byte line per
    // trip, and adds them up in order.  One sample per trip kept every line alive for eight trips of a whole wave
    // (64 lines per lane-row), more than the caches hold: the lines were fetched again and again (2.0 ms at 1080p).
    const float4 *row = sbuf + (uint64_t)a * (uint32_t)nk;
    for (int base = 0; base < nk && !done; base += 8) {
        float4 c8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) c8[j] = row[base + j < nk ? base + j : nk - 1];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (base + j >= nk || done) continue;
            const float4 c = c8[j];
            ++s.n;
            s.sum[0] += c.x; s.sum[1] += c.y; s.sum[2] += c.z;
            s.sum_sq[0] += c.x * c.x; s.sum_sq[1] += c.y * c.y; s.sum_sq[2] += c.z * c.z;
            if (s.n < F.min_spp) continue;
            // src/main.cpp:360-372
            bool below = true;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                float sigma = (s.sum_sq[ch] - ((s.sum[ch] * s.sum[ch]) / (float)s.n)) / (float)(s.n - 1);
                sigma = fsqrt(fmaxf(0.0f, sigma));
                const float delta = 3.0f * (sigma / fsqrt((float)s.n));
                below = below && (delta < F.delta_max);
            }
            if (below) done = true;
        }
    }
    if (s.n >= F.max_spp) done = true;
    if (!done) {
        state[pix] = s;
        next_active[atomicAdd(next_count, 1u)] = pix;
        return;
    }
    float lin[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) lin[ch] = s.sum[ch] / (float)s.n;
    if (out.linear) { out.linear[pix * 3] = lin[0]; out.linear[pix * 3 + 1] = lin[1]; out.linear[pix * 3 + 2] = lin[2]; }
    if (out.rgb) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) out.rgb[pix * 3 + ch] = to_byte(F.srgb ? lin2srgb(lin[ch]) : lin[ch]);
    }
    if (out.sample_count) out.sample_count[pix] = s.n;
}

// ---- frame exchange between GPUs ----
// A rank's pixels travel as one 16-byte record each: (r | g << 8 | b << 16, z, sample count, object id).  Record s of
// rank r belongs to tile r + (s / ts^2) * world, pixel (s % ts, s % ts^2 / ts) of it: every tile takes ts^2 records,
// the ones of a ragged tile's missing pixels are zero, so the place of a pixel follows from arithmetic alone.
RT_DEV bool exchange_pixel(uint32_t s, int rank, int world, int ts, int width, int height, int &pix)
{
    const int tx = (width + ts - 1) / ts, ty = (height + ts - 1) / ts;
    const int per = ts * ts;
    const int t = rank + (int)(s / (uint32_t)per) * world;
    if (t >= tx * ty) return false;
    const int o = (int)(s % (uint32_t)per);
    const int x = (t % tx) * ts + o % ts, y = (t / tx) * ts + o / ts;
    if (x >= width || y >= height) return false;
    pix = y * width + x;
    return true;
}
__global__ __launch_bounds__(kBlock) void k_frame_pack(FramePlanes in, int width, int height, int ts, int rank, int world,
                                                       uint32_t n_records, uint4 *__restrict__ packed)
{
    const uint32_t s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= n_records) return;
    int pix;
    uint4 rec = make_uint4(0, 0, 0, 0);
    if (exchange_pixel(s, rank, world, ts, width, height, pix)) {
        rec.x = (uint32_t)in.rgb[pix * 3] | ((uint32_t)in.rgb[pix * 3 + 1] << 8) | ((uint32_t)in.rgb[pix * 3 + 2] << 16);
        rec.y = __float_as_uint(in.z[pix]);
        rec.z = (uint32_t)in.sample_count[pix];
        rec.w = (uint32_t)in.object_id[pix];
    }
    packed[s] = rec;
}
// packed_all = the ranks' record arrays one after the other; the pixels of `skip_rank` (this rank's own) stay as they are
__global__ __launch_bounds__(kBlock) void k_frame_unpack(const uint4 *__restrict__ packed_all, int width, int height, int ts,
                                                         int skip_rank, int world, uint32_t n_records, FramePlanes out)
{
    const uint32_t s = blockIdx.x * kBlock + threadIdx.x;
    const int rank = (int)blockIdx.y;
    if (s >= n_records || rank == skip_rank) return;
    int pix;
    if (!exchange_pixel(s, rank, world, ts, width, height, pix)) return;
    const uint4 rec = packed_all[(size_t)rank * n_records + s];
    out.rgb[pix * 3] = (uint8_t)rec.x; out.rgb[pix * 3 + 1] = (uint8_t)(rec.x >> 8); out.rgb[pix * 3 + 2] = (uint8_t)(rec.x >> 16);
    out.z[pix] = __uint_as_float(rec.y);
    out.sample_count[pix] = (int32_t)rec.z;
    out.object_id[pix] = (int32_t)rec.w;
}

// ---- batch queries: grid-stride loops, the grid is bounded by the host so that the
// per-thread stack overflow area covers every thread ----
__global__ __launch_bounds__(kBlock) void k_trace(DScene S, const float *__restrict__ rays6, int n, rt_hit *__restrict__ out)
{
    extern __shared__ uint32_t lds[];
    const LaneStack st = lane_stack(lds, S);
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        Ray r;
        r.p = ld3(rays6 + (size_t)i * 6); r.d = ld3(rays6 + (size_t)i * 6 + 3);
        Hit h;
        Tally<false> tl;
        const bool hit = trace_closest<false, kMaxSceneDepth - 1>(S, r, st, h, tl);
        if (hit && h.node >= 0) finish_hit(S, r, h, true);
        else { h.p = mk(0, 0, 0); h.n = mk(0, 0, 0); h.uvw = mk(0.5f, 0.5f, 0.5f); }
        rt_hit o;
        o.z = h.z;
        o.p[0] = h.p.x; o.p[1] = h.p.y; o.p[2] = h.p.z;
        o.n[0] = h.n.x; o.n[1] = h.n.y; o.n[2] = h.n.z;
        o.uvw[0] = h.uvw.x; o.uvw[1] = h.uvw.y; o.uvw[2] = h.uvw.z;
        o.node = hit ? h.node : -1;
        o.front = h.front;
        out[i] = o;
    }
}

__global__ __launch_bounds__(kBlock) void k_trace_shadow(DScene S, const float *__restrict__ rays6,
                                                         const float *__restrict__ t_max, int n, int32_t *__restrict__ out)
{
    extern __shared__ uint32_t lds[];
    const LaneStack st = lane_stack(lds, S);
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        Ray r;
        r.p = ld3(rays6 + (size_t)i * 6); r.d = ld3(rays6 + (size_t)i * 6 + 3);
        Tally<false> tl;
        out[i] = trace_shadow<false, kMaxSceneDepth - 1>(S, r, t_max[i], st, tl) ? 1 : 0;
    }
}

__global__ __launch_bounds__(kBlock) void k_trace_paths(DScene S, DFrame F, const float *__restrict__ rays6, int n,
                                                        float *__restrict__ radiance3, DCounters *ctr)
{
    extern __shared__ uint32_t lds[];
    const LaneStack st = lane_stack(lds, S);
    PathTally pt;
    Tally<false> tl;
    uint32_t paths = 0;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        Ray r;
        r.p = ld3(rays6 + (size_t)i * 6); r.d = ld3(rays6 + (size_t)i * 6 + 3);
        PathState ps;
        path_begin(ps, r);
        rng_set_sequence(ps.rng, (uint64_t)i + 1, F.seed);
        while (path_step<false, kMaxSceneDepth - 1>(S, F, ps, st, pt, tl, nullptr, nullptr)) {}
        radiance3[(size_t)i * 3] = ps.result.x; radiance3[(size_t)i * 3 + 1] = ps.result.y; radiance3[(size_t)i * 3 + 2] = ps.result.z;
        ++paths;
    }
    flush_counters<false>(ctr, pt, tl, paths);
}

} // namespace rtd
