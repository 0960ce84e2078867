// Photon maps on the GPU: emission and scattering of photon paths, and the k-nearest-photon
// irradiance estimate over the left-balanced kd-tree.
//   PointLight::RandomPhoton            src/lights.h:92-124
//   MtlBlinn::GenerateSample (snapshot) project12.zip src/materials.h:82-193   (p12 below)
//   emission loops                      p12 src/main.cpp:292-470
//   PhotonMap::AddPhoton / SetPower     src/photonmap.h:148-172
//   PhotonMap::IrradianceEstimate       src/photonmap.h:268-309
//   PhotonMap::LocatePhotons            src/photonmap.h:339-414
//
// The snapshot's lights.h and materials.h call cos / sin / pow / sqrt from inside class bodies
// that are compiled before <math.h> adds the float overloads, so those calls run in double and
// the product with them is narrowed once; the arithmetic below does the same.
#pragma once
#include "kernels.h"

namespace rtd {

constexpr float kInvPiF = (float)(1.0f / M_PI);
constexpr double kEightPi = 8.0f * M_PI;
constexpr double kTwoPi = 2.0f * M_PI;
RT_DEV float sum3(const float *c) { return c[0] + c[1] + c[2]; }

// (float)(a * cos(t)), (float)(a * sin(t)) with a, t float: the product is formed in double
RT_DEV void scaled_sincos(float a, float t, float &as, float &ac)
{
    double s, c;
    dsincos((double)t, s, c);
    as = (float)((double)a * s);
    ac = (float)((double)a * c);
}

// PointLight::RandomPhoton (src/lights.h:92-124): 4 draws
RT_DEV void light_random_photon(const rt_light &l, Rng &rng, Ray &r, V3 &c)
{
    float randU = rng_float(rng);
    float randV = rng_float(rng);
    const float theta = (float)(kTwoPi * randU);
    const float posZ = l.size * (1.0f - 2.0f * randV);
    const float rProj = fsqrt(l.size * l.size - posZ * posZ);
    float posX, posY;
    scaled_sincos(rProj, theta, posY, posX);
    const V3 position = ld3(l.position);
    const V3 rayPos = mk(posX, posY, posZ) + position;
    r.p = rayPos;
    randU = rng_float(rng);
    randV = rng_float(rng);
    const float phi = (float)(kTwoPi * randU);
    const float cosTheta = randV;
    const float sinTheta = fsqrt(1.0f - cosTheta * cosTheta);
    float dirX, dirY;
    scaled_sincos(sinTheta, phi, dirY, dirX);
    const V3 norm = normalize(rayPos - position);
    V3 u, v;
    orthonormals(norm, u, v);
    r.d = (u * dirX + v * dirY) + norm * cosTheta;
    c = ((((ld3(l.intensity) * 8.0f) * (float)M_PI) * l.size) * l.size) * cosTheta;
}

// The snapshot's MtlBlinn::GenerateSample (p12 materials.h:82-193): 3 draws unless absorbed.
// n = unit shading normal, v = unit view vector.  Returns the `bounce` flag of the emission loops.
RT_DEV bool p12_mtl_generate(const rt_material &m, V3 Ns, V3 V, int front, Rng &rng, V3 &dir, V3 &mult, float &prob, int &lobe)
{
    float pd, ps, pt;
    lobe_probs(m, pd, ps, pt);
    const float randomNum = rng_float(rng);
    if (randomNum < pd) {
        lobe = LOBE_DIFFUSE;
        mult = ld3(m.diffuse);
        prob = pd;
        const float r1 = rng_float(rng);
        const float r2 = rng_float(rng);
        const float z = fsqrt(r1);
        const float sinTheta = fsqrt(1.0f - r1);
        const float phi = (float)(kTwoPi * r2);
        float x, y;
        scaled_sincos(sinTheta, phi, y, x);
        V3 u, v;
        orthonormals(Ns, u, v);
        dir = (u * x + v * y) + Ns * z;
        return true;
    }
    if (!(randomNum < pd + ps + pt)) return false;
    const bool spec = randomNum < pd + ps;
    const V3 N = (spec || front) ? Ns : -Ns;
    lobe = spec ? LOBE_SPECULAR : LOBE_TRANSMISSION;
    prob = spec ? ps : pt;
    const float r1 = rng_float(rng);
    const float r2 = rng_float(rng);
    const float phi = (float)(kTwoPi * r1);
    const float cosTheta = (float)pow((double)r2, (double)(1.0f / (m.glossiness + 1.0f)));
    const float sinTheta = fsqrt(1.0f - cosTheta * cosTheta);
    float x, y;
    scaled_sincos(sinTheta, phi, y, x);
    V3 u, v;
    orthonormals(N, u, v);
    const V3 h = (u * x + v * y) + N * cosTheta;
    const V3 refl = (h * 2.0f) * dot(V, h) - V;
    if (spec) {
        dir = refl;
        mult = ld3(m.specular) * dot(dir, Ns);
        return dot(dir, Ns) > 0.0f;
    }
    const float ratio = front ? 1.0f / m.ior : m.ior;
    const float cos_i = dot(V, h);
    const float sin_t_sq = (ratio * ratio) * (1.0f - cos_i * cos_i);
    if (sin_t_sq > 1.0f) { // total internal reflection
        lobe = LOBE_SPECULAR;
        dir = refl;
        mult = ld3(m.refraction) * fabsf(dot(dir, Ns));
        return dot(dir, N) > 0.0f;
    }
    const float cos_t = fsqrt(1.0f - sin_t_sq);
    dir = (-V) * ratio + h * (ratio * cos_i - cos_t);
    mult = ld3(m.refraction) * fabsf(dot(dir, Ns));
    return dot(dir, N) < 0.0f;
}

RT_DEV bool is_photon_surface(const rt_material &m) { return sum3(m.diffuse) > 0.0f; } // p12 materials.h:50

// PhotonData::SetPower + AddPhoton (src/photonmap.h:148-172): one 32-byte record, two 16-byte stores
RT_DEV void photon_store(rt_photon *dst, V3 pos, V3 dir, V3 c)
{
    float power = c.x;
    if (power < c.y) power = c.y;
    if (power < c.z) power = c.z;
    const V3 q = c / power;
    const uint32_t rgb = (uint32_t)to_byte(q.x) | ((uint32_t)to_byte(q.y) << 8) | ((uint32_t)to_byte(q.z) << 16);
    float4 *d = (float4 *)dst;
    d[0] = make_float4(pos.x, pos.y, pos.z, power);
    d[1] = make_float4(dir.x, dir.y, dir.z, __uint_as_float(rgb));
}

constexpr uint64_t kSeedGlobalMap = 0x50484F54ULL, kSeedCausticsMap = 0x43415553ULL;

// One lane = one photon path at a time (paths come from the launch's work queue, so a wave
// stays full although paths are 1..max_segments rays long); one trip of the loop = one ray.
// The photons a path stores must land where the serial loop of the reference would have put
// them, so the launch runs twice: WRITE = false counts the photons of every path, the host
// scans the counts into offsets, WRITE = true traces the same paths again (same streams, same
// arithmetic) and writes photon `seq` of path `i` to out[offsets[i] + seq] if that is below
// `capacity`.
#ifndef RT_PHOTON_MIN_WAVES
#define RT_PHOTON_MIN_WAVES 3 // waves per SIMD the register allocator must leave room for (1.1 M photons emitted in 9.3 / 6.9 / 5.6 / 5.5 ms at 6 / 4 / 3 / 2)
#endif
template <bool WRITE, int LEVELS>
__global__ __launch_bounds__(kBlock, RT_PHOTON_MIN_WAVES) void k_photon_paths(DScene S, rt_photon_params P, int which, uint64_t path_begin,
                                                            uint32_t n_paths, uint32_t *__restrict__ counts,
                                                            const uint32_t *__restrict__ offsets, uint64_t base,
                                                            rt_photon *__restrict__ out, uint64_t capacity, DCounters *ctr,
                                                            unsigned long long *queue)
{
    extern __shared__ uint32_t lds[];
    const LaneStack st = lane_stack(lds, S);
    const uint32_t lane = threadIdx.x & 63u;
    PathTally pt;
    Tally<false> tl;
    uint32_t paths = 0;
    const rt_light light = load_uniform(&S.lights[0]);
    const bool caustics_map = which == RT_MAP_CAUSTICS;
    const bool direct_only = !caustics_map && (P.flags & (RT_PM_DIRECT | RT_PM_MONTE_CARLO));
    const uint64_t seed = P.seed ^ (caustics_map ? kSeedCausticsMap : kSeedGlobalMap);

    WorkQueue wq(queue, n_paths, kChunk, S.cancel);
    bool alive = false, firstHit = true;
    uint32_t item = 0, seq = 0;
    int seg = 0;
    Rng rng; rng.state = 0; rng.inc = 1;
    Ray ray{mk(0, 0, 0), mk(0, 0, 1)};
    V3 c = mk(0, 0, 0);

    for (;;) {
        const uint64_t need = __ballot(!alive);
        if (need != 0) {
            wq.refill(lane);
            const uint64_t avail = wq.end - wq.next;
            const uint32_t want = (uint32_t)__popcll(need);
            const uint32_t take = avail < want ? (uint32_t)avail : want;
            const uint32_t rank = (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
            if (!alive && rank < take) {
                item = (uint32_t)(wq.next + rank);
                rng_set_sequence(rng, path_begin + item, seed);
                light_random_photon(light, rng, ray, c);
                firstHit = true; seq = 0; seg = 0;
                alive = true;
                ++paths;
            }
            wq.next += take;
        }
        if (!__any(alive)) {
            if (wq.exhausted) break;
            continue;
        }
        if (alive) {
            bool go_on = false;
            ray.p = ray.p + ray.d * 0.0002f;
            Hit h;
            ++pt.closest;
            const bool hit = trace_closest<false, LEVELS>(S, ray, st, h, tl);
            if (hit && h.node >= 0) { // a miss or a light ends the path
                finish_hit(S, ray, h, false);
                const rt_material &m = S.materials[hit_material(S, h.node, h.face)];
                bool store = false;
                V3 newDir = mk(0, 0, 0), mult = mk(0, 0, 0);
                float prob = 1.0f;
                int lobe = LOBE_NONE;
                if (direct_only) store = is_photon_surface(m); // first hits only (p12 main.cpp:298-316)
                else {
                    const bool bounce = p12_mtl_generate(m, normalize(h.n), -normalize(ray.d), h.front, rng, newDir, mult, prob, lobe);
                    if (firstHit) { // the first hit is never stored; it decides which map the path belongs to
                        go_on = caustics_map ? (bounce && lobe != LOBE_DIFFUSE)
                                             : (bounce && !((P.flags & RT_PM_CAUSTICS) && lobe != LOBE_DIFFUSE));
                        firstHit = false;
                    } else {
                        store = is_photon_surface(m);
                        go_on = bounce && !(caustics_map && store); // a caustic photon ends where it is stored
                    }
                }
                if (store) {
                    if (WRITE) {
                        const uint64_t idx = base + offsets[item] + seq;
                        if (idx < capacity) photon_store(out + idx, h.p, ray.d, c);
                    }
                    ++seq;
                }
                if (go_on) {
                    ray.d = newDir;
                    ray.p = h.p;
                    c = c * (mult / prob);
                }
            }
            ++seg;
            if (!go_on || seg >= P.max_segments) {
                if (!WRITE) counts[item] = seq;
                alive = false;
            }
        }
    }
    flush_counters<false>(ctr, pt, tl, paths);
}

// ---- exclusive scan of the per-path photon counts (three small launches) ----
constexpr uint32_t kScanTile = 1024; // elements per workgroup: 4 per thread
__global__ __launch_bounds__(kBlock) void k_scan_tiles(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, uint32_t n,
                                                       uint32_t *__restrict__ tile_sums)
{
    __shared__ uint32_t part[kBlock];
    const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * 4;
    uint32_t v[4], sum = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = base + i < n ? in[base + i] : 0u; sum += v[i]; }
    part[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t o = 1; o < kBlock; o <<= 1) { // Hillis-Steele over the 256 thread sums
        const uint32_t add = threadIdx.x >= o ? part[threadIdx.x - o] : 0u;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum; // exclusive prefix of this thread within the tile
#pragma unroll
    for (int i = 0; i < 4; ++i) { if (base + i < n) out[base + i] = run; run += v[i]; }
    if (threadIdx.x == kBlock - 1) tile_sums[blockIdx.x] = part[kBlock - 1];
}
// one workgroup: exclusive scan of the tile sums in place, grand total to *total
__global__ __launch_bounds__(kBlock) void k_scan_sums(uint32_t *__restrict__ tile_sums, uint32_t n_tiles, unsigned long long *total)
{
    __shared__ uint32_t part[kBlock];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t first = 0; first < n_tiles; first += kBlock) {
        const uint32_t i = first + threadIdx.x;
        const uint32_t v = i < n_tiles ? tile_sums[i] : 0u;
        part[threadIdx.x] = v;
        __syncthreads();
        for (uint32_t o = 1; o < kBlock; o <<= 1) {
            const uint32_t add = threadIdx.x >= o ? part[threadIdx.x - o] : 0u;
            __syncthreads();
            part[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < n_tiles) tile_sums[i] = carry + part[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 0) carry += part[kBlock - 1];
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}
__global__ __launch_bounds__(kBlock) void k_scan_add(uint32_t *__restrict__ out, uint32_t n, const uint32_t *__restrict__ tile_sums)
{
    const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * 4;
    const uint32_t add = tile_sums[blockIdx.x];
#pragma unroll
    for (int i = 0; i < 4; ++i) if (base + i < n) out[base + i] += add;
}

// ---- irradiance estimate ----
// EstimateIrradiance<128>(irrad, dir, radius, pos, normal, 1) (src/photonmap.h:98,268-309) with
// LocatePhotons<true> (src/photonmap.h:339-414).  The reference recurses: near child, then the far
// child if the split plane is closer than the current search radius, then the node's own photon.
// The loop below visits the nodes in exactly that order with an explicit stack (one entry per
// level: node, which child is the far one, squared plane distance), keeps the reference's array
// of found photons and its max-heap as indices into the map, and sums their powers in array order,
// so the estimate has the reference's bits.  `tree` is the balanced map, node 1 first.
constexpr int kGatherK = RT_GATHER_MAX_PHOTONS;
constexpr uint32_t kFarIsLow = 0x80000000u, kFarDone = 0x40000000u;

// A lookup keeps the reference's array of found photons (128 entries of (dist2, photon), a max-heap once it is full):
// every photon closer than the current search radius sifts through up to seven of its levels.  With the array in
// global memory (1 KB per lane, no cache holds those of all lanes in flight) the lookup waited for its own heap more
// than for the photons; all of it in LDS leaves room for too few lookups at a time to hide the latency of the tree.
// So: the top six levels (entries 1..63), the ones every replacement passes, live in LDS as 4 bytes of dist2 and one
// byte that names one of 128 places in a table of photon ids; the last level (entries 64..128) and the table live in
// global memory.  A photon's id is written to the table once, when it is found (a replacement reuses the place of the
// entry it drives out), and read once, for the sum; the last level costs one 16-byte read at the end of a deep sift.
// A wave looks photons up for kGatherSlots of its lanes at a time: a lane takes part in round
// (rank among the wave's lookups) / kGatherSlots with slot rank % kGatherSlots.  Entry e of slot s of wave w lies at
// dist2[e - 1][w * kGatherSlots + s]: the LDS bank depends on the slot alone, whatever entries the lanes are at.
#ifndef RT_GATHER_SLOTS
#define RT_GATHER_SLOTS 32
#endif
constexpr uint32_t kGatherSlots = RT_GATHER_SLOTS;
static_assert(kGatherSlots == 8 || kGatherSlots == 16 || kGatherSlots == 32 || kGatherSlots == 64, "slots per wave: a power of two that keeps the LDS bank a function of the slot");
constexpr uint32_t kHeapSlotStride = kGatherSlots * (kBlock / 64); // slots of a workgroup
constexpr int kHeapLdsEntries = 63;                                 // whole levels: 1 + 2 + ... + 32 (the replacement's rounds are written for this split)
constexpr size_t kHeapLdsBytes = (size_t)kHeapLdsEntries * kHeapSlotStride * 5;
static_assert(kGatherK == 128, "the replacement's rounds are written for a heap of seven levels and entry 128");
constexpr uint32_t kHeapGlobalEntries = 66; // entries 64..128 and one of padding: pairs are read with one 16-byte load
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct LaneHeap {
    typedef RT_GLOBAL_AS u32x2 *GH2;
    typedef RT_GLOBAL_AS u32x4 *GH4;
    typedef RT_GLOBAL_AS uint32_t *GI;
    float *d2;    // entry e <= 63: dist2 at d2[(e - 1) * kHeapSlotStride]
    uint8_t *tag; //                its place in the id table at tag[(e - 1) * kHeapSlotStride]
    GH2 g;        // entry e >= 64 at g[e - 64] as (dist2, place)
    GI ids;       // [128] photon of each place
    RT_DEV u32x2 get(int e) const
    {
        if (e <= kHeapLdsEntries) return u32x2{__float_as_uint(d2[(uint32_t)(e - 1) * kHeapSlotStride]), tag[(uint32_t)(e - 1) * kHeapSlotStride]};
        return g[e - 64];
    }
    RT_DEV void put(int e, u32x2 v) const
    {
        if (e <= kHeapLdsEntries) { d2[(uint32_t)(e - 1) * kHeapSlotStride] = __uint_as_float(v.x); tag[(uint32_t)(e - 1) * kHeapSlotStride] = (uint8_t)v.y; }
        else g[e - 64] = v;
    }
    RT_DEV u32x4 lds_pair(int j) const // j even, j + 1 <= 63
    {
        const uint32_t o = (uint32_t)(j - 1) * kHeapSlotStride;
        return u32x4{__float_as_uint(d2[o]), tag[o], __float_as_uint(d2[o + kHeapSlotStride]), tag[o + kHeapSlotStride]};
    }
    RT_DEV u32x4 global_pair(int j) const { return *(GH4)&g[j - 64]; } // j even, j >= 64
    RT_DEV void lds_put(int e, float dist2, uint32_t place) const
    {
        d2[(uint32_t)(e - 1) * kHeapSlotStride] = dist2;
        tag[(uint32_t)(e - 1) * kHeapSlotStride] = (uint8_t)place;
    }
    // entries j and j + 1, j even: both on the same side of the boundary
    RT_DEV u32x4 pair(int j) const
    {
        if (j < kHeapLdsEntries) {
            const uint32_t o = (uint32_t)(j - 1) * kHeapSlotStride;
            return u32x4{__float_as_uint(d2[o]), tag[o], __float_as_uint(d2[o + kHeapSlotStride]), tag[o + kHeapSlotStride]};
        }
        return *(GH4)&g[j - 64];
    }
};
// lds: the workgroup's dynamic LDS, the lanes' stacks first (S.lds_levels entries per thread), then the heaps' tops
RT_DEV LaneHeap lane_heap(const DScene &S, uint32_t *lds, uint32_t slot)
{
    LaneHeap h;
    const uint32_t in_block = (threadIdx.x >> 6) * kGatherSlots + slot;
    float *base = (float *)(lds + (size_t)S.lds_levels * kStackStride * 2);
    h.d2 = base + in_block;
    h.tag = (uint8_t *)(base + (size_t)kHeapLdsEntries * kHeapSlotStride) + in_block;
    const size_t in_grid = (size_t)blockIdx.x * kHeapSlotStride + in_block;
    h.g = (LaneHeap::GH2)((u32x2 *)S.gather_heaps + in_grid * kHeapGlobalEntries);
    h.ids = (LaneHeap::GI)(S.gather_ids + in_grid * RT_GATHER_MAX_PHOTONS);
    return h;
}

// The walk's stack (one entry per tree level).  While a wave looks photons up none of its lanes is inside a BVH walk,
// so the slots share the LDS stack storage of all 64 lanes: 64 / kGatherSlots kd levels per BVH level.  A BVH walk is
// busiest near the bottom of its stack, a kd-tree walk near the top (the levels next to the leaves), so the *deepest*
// levels are the ones in LDS and the levels next to the root go to the lane's overflow array if the tree is taller.
struct KdStack {
    LaneStack st;
    typedef __attribute__((address_space(3))) u32x2 *LdsEntry; // named address space: the pop must not wait for the heap's global stores
    LdsEntry wave_lds; // level 0 of lane 0 of this wave
    uint32_t slot;
    uint32_t capacity; // kd levels the LDS storage holds
    uint32_t base;     // first level kept in LDS = levels that live in the overflow array
    RT_DEV KdStack(const LaneStack &s, uint32_t *lds, uint32_t slot_) : st(s), slot(slot_), base(0)
    {
        wave_lds = (LdsEntry)((u32x2 *)lds + (threadIdx.x & ~63u));
        capacity = s.lds_levels * (64u / kGatherSlots);
    }
    RT_DEV void set_height(uint32_t height) { base = height > capacity ? height - capacity : 0u; }
    RT_DEV LdsEntry at(uint32_t level) const
    {
        constexpr uint32_t per = 64u / kGatherSlots;
        return wave_lds + (level / per) * kStackStride + (level % per) * kGatherSlots + slot;
    }
    RT_DEV void push(int sp, uint32_t data, float dist) const
    {
        const u32x2 e = {data, __float_as_uint(dist)};
        if ((uint32_t)sp >= base) *at((uint32_t)sp - base) = e;
        else st.ovf[(size_t)(uint32_t)sp * st.ovf_stride] = e;
    }
    RT_DEV void pop(int sp, uint32_t &data, float &dist) const
    {
        u32x2 e;
        if ((uint32_t)sp >= base) e = *at((uint32_t)sp - base);
        else e = st.ovf[(size_t)(uint32_t)sp * st.ovf_stride];
        data = e.x; dist = __uint_as_float(e.y);
    }
};

template <bool COUNT>
RT_DEV V3 photon_irradiance(const rt_photon *tree0, uint32_t n, V3 pos, V3 normal, float radius, KdStack st,
                            const LaneHeap H, uint32_t &visits)
{
    if (n == 0) return mk(0, 0, 0);
    st.set_height(32u - (uint32_t)__clz((int)n)); // levels of the balanced tree = most stack entries + 1
    const GF4 tree = (GF4)(const void *)tree0 - 2; // node i = tree[2 i], tree[2 i + 1]
    const int half = (int)n / 2 - 1;               // nodes below this index are treated as internal (src/photonmap.h:197,346)
    int found = 0;
    float lim = radius * radius; // dist2[0]
    int sp = 0;
    uint32_t cur = 1;
    bool going_down = true;
    float top2[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // entries 1..7 of the full heap: dist2 and place
    uint32_t ttag[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // One node per trip, whichever way the lane is going, so that the lanes of a wave stay together: on the way down
    // an inner node is passed (its far side noted on the stack); a leaf, or a node come back to from below, has its
    // own photon looked at, and then the walk backs up by one stack entry.
    for (;;) {
        const f4 q0 = tree[2 * cur], q1 = tree[2 * cur + 1];
        if (going_down && (int)cur < half) {
            const uint32_t axis = (__float_as_uint(q1.w) >> 24) & 3u;
            const float split = axis == 0 ? q0.x : (axis == 1 ? q0.y : q0.z);
            const float qa = axis == 0 ? pos.x : (axis == 1 ? pos.y : pos.z);
            const float dist = qa - split;
            const bool high_first = dist > 0.0f;
            st.push(sp, cur | (high_first ? kFarIsLow : 0u), dist * dist);
            ++sp;
            if (COUNT) ++visits;
            cur = high_first ? 2 * cur + 1 : 2 * cur;
            continue;
        }
        if (COUNT && going_down) ++visits;
        // the entry the walk backs up to is read now: its latency passes while the photon is looked at and entered
        uint32_t data = 0; float plane2 = 0.0f;
        if (sp > 0) st.pop(sp - 1, data, plane2);
        { // the node's own photon (src/photonmap.h:357-413)
            const uint32_t node = cur;
            const V3 dif = mk(q0.x, q0.y, q0.z) - pos;
            const float dist2 = len2(dif);
            if (dist2 < lim && !(dot(mk(q1.x, q1.y, q1.z), normal) >= 0.0f)) {
                if (found < kGatherK) {
                    H.ids[found] = node;
                    ++found;
                    H.put(found, u32x2{__float_as_uint(dist2), (uint32_t)(found - 1)});
                    if (found == kGatherK) { // build the max-heap
                        const int half_found = found >> 1;
                        for (int k = half_found; k >= 1; --k) {
                            int parent = k;
                            const u32x2 t = H.get(k);
                            const float td2 = __uint_as_float(t.x);
                            while (parent <= half_found) {
                                int j = parent + parent;
                                const u32x4 c = H.pair(j);
                                float dj = __uint_as_float(c.x);
                                uint32_t ij = c.y;
                                if (j < found && dj < __uint_as_float(c.z)) { ++j; dj = __uint_as_float(c.z); ij = c.w; }
                                if (td2 >= dj) break;
                                H.put(parent, u32x2{__float_as_uint(dj), ij});
                                parent = j;
                            }
                            H.put(parent, t);
                        }
#pragma unroll
                        for (int k = 1; k < 8; ++k) { const u32x2 e = H.get(k); top2[k] = __uint_as_float(e.x); ttag[k] = e.y; }
                    }
                } else { // replace the farthest; the newcomer takes over its place in the id table
                    // the three levels every replacement passes are in registers (top[1..7]); selects, no memory
                    const uint32_t place = ttag[1];
                    H.ids[place] = node;
                    bool placed = false;
                    int parent;
                    { // entry 1 against entries 2, 3
                        const bool right = top2[2] < top2[3];
                        const float dj = right ? top2[3] : top2[2];
                        const uint32_t ij = right ? ttag[3] : ttag[2];
                        placed = dist2 > dj;
                        top2[1] = placed ? dist2 : dj;
                        ttag[1] = placed ? place : ij;
                        parent = right ? 3 : 2;
                    }
                    { // entry 2 or 3 against its children among 4..7
                        const bool low = parent == 2;
                        const float c0 = low ? top2[4] : top2[6], c1 = low ? top2[5] : top2[7];
                        const uint32_t i0 = low ? ttag[4] : ttag[6], i1 = low ? ttag[5] : ttag[7];
                        const bool right = c0 < c1;
                        const float dj = right ? c1 : c0;
                        const uint32_t ij = right ? i1 : i0;
                        const bool here = !placed && dist2 > dj; // the newcomer stays at `parent`
                        const float v = here ? dist2 : dj;
                        const uint32_t vi = here ? place : ij;
                        if (!placed) {
                            top2[2] = low ? v : top2[2]; ttag[2] = low ? vi : ttag[2];
                            top2[3] = low ? top2[3] : v; ttag[3] = low ? ttag[3] : vi;
                        }
                        placed = placed || here;
                        parent = 2 * parent + (right ? 1 : 0);
                    }
                    // Below the registers two levels are decided per round trip: the children of the entry and all four
                    // grandchildren are read at once (8..15 and 16..31 from LDS; then 32..63 from LDS together with
                    // 64..127 from global memory).
                    if (!placed) { // entry 4..7 (registers) against 8..15, the winner against its children among 16..31
                        const int c = 2 * parent;
                        const u32x4 ch = H.lds_pair(c), g01 = H.lds_pair(2 * c), g23 = H.lds_pair(2 * c + 2);
                        const bool r1 = __uint_as_float(ch.x) < __uint_as_float(ch.z);
                        const float d1 = r1 ? __uint_as_float(ch.z) : __uint_as_float(ch.x);
                        const uint32_t t1 = r1 ? ch.w : ch.y;
                        const bool here0 = dist2 > d1;
                        const float v = here0 ? dist2 : d1;
                        const uint32_t vi = here0 ? place : t1;
#pragma unroll
                        for (int k = 4; k < 8; ++k) {
                            top2[k] = parent == k ? v : top2[k];
                            ttag[k] = parent == k ? vi : ttag[k];
                        }
                        placed = here0;
                        if (!here0) {
                            const int j1 = c + (r1 ? 1 : 0);
                            const u32x4 gg = r1 ? g23 : g01;
                            const bool r2 = __uint_as_float(gg.x) < __uint_as_float(gg.z);
                            const float d2_ = r2 ? __uint_as_float(gg.z) : __uint_as_float(gg.x);
                            const uint32_t t2 = r2 ? gg.w : gg.y;
                            const bool here1 = dist2 > d2_;
                            H.lds_put(j1, here1 ? dist2 : d2_, here1 ? place : t2);
                            placed = here1;
                            parent = 2 * j1 + (r2 ? 1 : 0);
                        }
                    }
                    if (!placed) { // entry 16..31 against 32..63 (LDS), the winner against its children among 64..127 (global)
                        const int c = 2 * parent;
                        const u32x4 g01 = H.global_pair(2 * c), g23 = H.global_pair(2 * c + 2);
                        const u32x4 ch = H.lds_pair(c);
                        const bool r1 = __uint_as_float(ch.x) < __uint_as_float(ch.z);
                        const float d1 = r1 ? __uint_as_float(ch.z) : __uint_as_float(ch.x);
                        const uint32_t t1 = r1 ? ch.w : ch.y;
                        const bool here0 = dist2 > d1;
                        H.lds_put(parent, here0 ? dist2 : d1, here0 ? place : t1);
                        placed = here0;
                        if (!here0) {
                            const int j1 = c + (r1 ? 1 : 0);
                            const u32x4 gg = r1 ? g23 : g01;
                            const bool r2 = __uint_as_float(gg.x) < __uint_as_float(gg.z);
                            const float d2_ = r2 ? __uint_as_float(gg.z) : __uint_as_float(gg.x);
                            const uint32_t t2 = r2 ? gg.w : gg.y;
                            const bool here1 = dist2 > d2_;
                            H.lds_put(j1, here1 ? dist2 : d2_, here1 ? place : t2);
                            placed = here1;
                            parent = 2 * j1 + (r2 ? 1 : 0);
                        }
                    }
                    int j = placed ? found + 1 : 2 * parent;
                    while (j <= found) { // what is left: entry 64 against entry 128
                        const u32x4 c = H.pair(j);
                        float dj = __uint_as_float(c.x);
                        uint32_t ij = c.y;
                        if (j < found && dj < __uint_as_float(c.z)) { ++j; dj = __uint_as_float(c.z); ij = c.w; }
                        if (dist2 > dj) break;
                        H.put(parent, u32x2{__float_as_uint(dj), ij});
                        parent = j;
                        j <<= 1;
                    }
                    if (!placed) H.put(parent, u32x2{__float_as_uint(dist2), place});
                    lim = top2[1];
                }
            }
        }
        // back up: the parent's far child if it can still hold something closer, else the parent's own photon
        if (sp == 0) break;
        const uint32_t parent = data & 0x3FFFFFFFu;
        if (!(data & kFarDone) && plane2 < lim) {
            st.push(sp - 1, data | kFarDone, plane2);
            cur = (data & kFarIsLow) ? 2 * parent : 2 * parent + 1;
            going_down = true;
        } else {
            --sp;
            cur = parent;
            going_down = false;
        }
    }
    if (found == kGatherK) {
#pragma unroll
        for (int k = 1; k < 8; ++k) H.put(k, u32x2{__float_as_uint(top2[k]), ttag[k]});
    }
    // the sum runs in array order; the three dependent reads per entry (place, photon id, photon) are issued for eight
    // entries at a time
    V3 sum = mk(0, 0, 0);
    const RT_GLOBAL_AS float *treef = (const RT_GLOBAL_AS float *)(const void *)tree0;
    for (int i0 = 1; i0 <= found; i0 += 8) {
        uint32_t place[8], id[8], rgb[8];
        float power[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) place[u] = H.get(i0 + u <= found ? i0 + u : found).y;
#pragma unroll
        for (int u = 0; u < 8; ++u) id[u] = H.ids[place[u]];
#pragma unroll
        for (int u = 0; u < 8; ++u) { // node i = floats 8 (i - 1) ..: power at 3, packed colour at 7
            power[u] = treef[(size_t)(id[u] - 1u) * 8u + 3u];
            rgb[u] = __float_as_uint(treef[(size_t)(id[u] - 1u) * 8u + 7u]);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (i0 + u > found) break;
            const V3 col = mk((float)(rgb[u] & 255u) / 255.0f, (float)((rgb[u] >> 8) & 255u) / 255.0f, (float)((rgb[u] >> 16) & 255u) / 255.0f);
            sum = sum + col * power[u]; // Color24::ToColor() * power, constant filter
        }
    }
    if (found > 0) {
        const float area = kPi * lim;
        if (area > 0.0f) sum = sum * (1.0f / area);
    }
    return sum;
}

__global__ __launch_bounds__(kBlock) void k_photon_irradiance(DScene S, const rt_photon *tree, uint32_t n_photons,
                                                              const float *__restrict__ pos3, const float *__restrict__ nrm3,
                                                              int n, float radius, float *__restrict__ out3, DCounters *ctr)
{
    extern __shared__ uint32_t lds[];
    const LaneStack st = lane_stack(lds, S);
    uint32_t visits = 0, gathers = 0;
    const uint32_t lane = threadIdx.x & 63u, slot = lane % kGatherSlots;
    const KdStack kd(st, lds, slot);
    const LaneHeap heap = lane_heap(S, lds, slot);
    for (int i0 = blockIdx.x * kBlock; i0 < n; i0 += gridDim.x * kBlock) {
        const int i = i0 + (int)threadIdx.x;
        for (uint32_t round = 0; round < 64u / kGatherSlots; ++round) { // neighbouring queries together
            if (i < n && lane / kGatherSlots == round) {
                const V3 e = photon_irradiance<true>(tree, n_photons, ld3(pos3 + (size_t)i * 3), ld3(nrm3 + (size_t)i * 3), radius, kd, heap, visits);
                out3[(size_t)i * 3] = e.x; out3[(size_t)i * 3 + 1] = e.y; out3[(size_t)i * 3 + 2] = e.z;
                ++gathers;
            }
        }
    }
    const uint64_t g = wave_sum(gathers), v = wave_sum(visits);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&ctr->gathers, (unsigned long long)g);
        atomicAdd(&ctr->photon_visits, (unsigned long long)v);
    }
}

} // namespace rtd
