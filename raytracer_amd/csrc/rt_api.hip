// C ABI of the GPU backend: context, scene upload, frame rendering, batch queries.
// Host-side control only; the arithmetic lives in device/*.h.
#include "device/wavefront.h"
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include "device/whitted.h"
#include "device/kdbuild.h"
#include "device/lbvh.h"
#include "device/debug.h"
#include "rt_mi355x.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <memory>
#include <string>
#include <vector>

using namespace rtd;

namespace {

thread_local std::string g_init_error;
// dynamic LDS a kernel may ask for: the CU's 160 KiB less what kernels declare statically (work-queue windows, list staging)
constexpr int kMaxDynamicLds = 160 * 1024 - 1024;

struct DeviceBuf {
    void *p = nullptr;
    size_t bytes = 0;
    ~DeviceBuf() { if (p) (void)hipFree(p); }
    DeviceBuf() = default;
    DeviceBuf(const DeviceBuf &) = delete;
    DeviceBuf &operator=(const DeviceBuf &) = delete;
    hipError_t reserve(size_t n)
    {
        if (n <= bytes) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
        hipError_t e = hipMalloc(&p, n ? n : 1);
        if (e == hipSuccess) bytes = n;
        return e;
    }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

} // namespace

// ---- scratch of the entry point that allocates nothing (rt_trace_brief_noalloc) ----
// Everything its kernel needs beyond the caller's rays and answers: the work-queue word, the counters and the
// levels of the lanes' BVH stacks that do not fit LDS.  One copy per GPU, held until the process ends; the name
// below is only a host-side handle, the device address comes from hipGetSymbolAddress (rt_init).
constexpr size_t kStaticQueueOff = 0, kStaticCountersOff = 256, kStaticStackOff = 4096;
constexpr size_t kStaticStackEntries = (size_t)1 << 22; // eight-byte stack entries: 32 MiB
constexpr size_t kStaticScratchBytes = kStaticStackOff + kStaticStackEntries * sizeof(uint2);
static_assert(sizeof(DCounters) <= kStaticStackOff - kStaticCountersOff, "counters overlap the stack area");
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

struct rt_ctx {
    int device = 0;
    unsigned char *static_scratch = nullptr; // device address of `scratch` on this context's GPU
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, evk0 = nullptr, evk1 = nullptr;
    std::string error;
    bool counting = false;

    // scene
    bool have_scene = false, have_meshes = false;
    DScene scene{};
    DeviceBuf d_nodes, d_tnodes, d_tlas, d_meshes, d_materials, d_lights, d_textures;
    std::vector<std::unique_ptr<DeviceBuf>> mesh_bufs;
    std::vector<int> host_node_material; // for validation
    // camera
    bool have_camera = false;
    rt_camera camera{};
    // frame
    DeviceBuf d_halton, d_active[2], d_active_all, d_active_sorted, d_keys[2], d_sort_tmp, d_next_count, d_state, d_sbuf, d_counters, d_queue;
    int chunk_items = 0;      // RT_CHUNK_ITEMS: fixed work-queue chunk of the path kernel (experiments), 0 = chosen per launch
    int order_pixels = -1;    // RT_ORDER: 1 = always deal a frame's pixels expensive-first, 0 = never (tile order), default: for tile shards only
    DeviceBuf d_stack_ovf;
    // wavefront pipeline (device/wavefront.h): per-slot path state, the bounce lists, list counts, event pairs around the ray launches
    DeviceBuf d_wf_state, d_wf_counts;
    // The list lengths a round went through, by what the round was (frame constants, items, samples, scene): a round
    // that comes again (the next frame of the same view) is launched from them without the host waiting for the device
    // after every bounce; the lengths the device then reports are compared with them afterwards.
    struct WfProfile { uint64_t key = 0; std::vector<WfCounts> counts; };
    std::vector<WfProfile> wf_profiles; // a handful, oldest replaced first
    size_t wf_profile_next = 0;
    uint64_t scene_generation = 0;
    bool wf_replay = true;              // RT_WF_REPLAY=0: always wait for the device's counts
    static constexpr int kMaxRayLaunches = 96;
    hipEvent_t ray_ev[2 * kMaxRayLaunches] = {};
    uint32_t finish_below = 0; // RT_FINISH_BELOW: paths left in a round at which the wavefront pipeline hands over to k_wf_finish (default: 512 per CU; measured 8 k / 32 k / 128 k / 512 k: 101.3 / 99.8 / 98.3 / 99.2 ms per 1080p frame, 20.3 / 19.3 / 17.9 / 19.1 ms on an eighth of it)
    int pipeline = -1; // RT_PIPELINE: 0 = lock-step path kernel (k_paths), 1 = wavefront; -1: chosen per scene (wavefront when it has meshes)
    size_t ovf_threads = 0;
    bool query_wide = false;  // RT_QUERY_WIDE=1: rt_trace_brief goes through the four-wide walk of the frames' rays (a test gate; the answers may differ from the reference's in the last bit of t)
    bool wide_from_gpu = true; // RT_WIDE_GPU_BVH=0: the four-wide BVH of the frames' rays is the reference's binary tree collapsed, not a tree of its own built on the GPU
    bool gpu_bvh = false;     // RT_GPU_BVH=1: build every mesh's BVH on the GPU even if the scene brings one
    double bvh_build_ms = 0;  // GPU time of the last upload's BVH builds
    int tlas_min_objects = 16; // RT_TLAS_MIN: object nodes from which a scene gets the hierarchy over their boxes (0 = never).  The reference's own scene (nine objects, five of them the walls of a box every ray is inside of) is slower through it: 38 -> 56 ms per frame
    int lds_levels_max = 13;
    int n_cu = 256;
    int ray_levels = 1; // hierarchy levels that have children
    DeviceBuf d_rgb, d_z, d_cnt, d_id, d_lin; // internal planes when the caller wants host output
    DeviceBuf d_q_in, d_q_aux, d_q_out;       // batch queries
    // photon maps: balanced trees the shader reads, and scratch of the emission launches
    DeviceBuf d_gather_heaps, d_pmap[2], d_ph_raw, d_ph_counts, d_ph_offsets, d_ph_tiles, d_ph_total;
    uint32_t pmap_n[2] = {0, 0};
    // Node i of a map (1-based, children 2i and 2i+1) sits at byte 32 i of its buffer: one unused slot in front puts
    // every pair of siblings on one 64-byte line and the four grandchildren of a node on one 128-byte line.
    rt_photon *pmap(int which) const { return (rt_photon *)((char *)d_pmap[which].p + sizeof(rt_photon)); }
    int pmap_flags = -1; // RT_PM_* bits the maps were filled with, -1 = not known (built from caller's photons)
    std::vector<int32_t> active_host;
    int active_key[6] = {-1, -1, -1, -1, -1, -1};
    rt_counters counters{};

    // rt_cancel / rt_progress: the only state another thread may touch while a call runs on this context
    // two words of pinned host memory the GPU reads and writes directly (a copy on another stream would wait behind a
    // persistent kernel that fills the chip): [0] the flag the kernels read when they draw from their queues,
    // [1] how far the running lock-step kernel's queue has been handed out
    unsigned long long *host_words = nullptr;
    std::atomic<int> cancel_req{0};
    int cancellable_depth = 0; // > 0 inside rt_photons_fill: the rt_photons_emit calls it makes belong to its cancellable call
    std::atomic<uint64_t> prog_pixels_total{0}, prog_pixels_done{0}, prog_round_items{0}, prog_round_done{0};
    std::atomic<int> prog_round{0}, prog_rendering{0}, prog_live_queue{0}; // live_queue: the round's progress is the queue word of the running kernel

    int fail(int code, const char *fmt, ...)
    {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        error = buf;
        return code;
    }
};

#define HIP_TRY(ctx, expr)                                                                          \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) return (ctx)->fail(RT_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

static bool cancelled(const rt_ctx *ctx) { return ctx->cancel_req.load(std::memory_order_relaxed) != 0; }

namespace {

// main() of the reference, src/main.cpp:404-414 (host arithmetic, libm tanf)
void derive_camera(const rt_camera &cam, DFrame &F)
{
    auto norm = [](const float *v, float *o) {
        const float l = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        o[0] = v[0] / l; o[1] = v[1] / l; o[2] = v[2] / l;
    };
    auto crossf = [](const float *a, const float *p, float *o) {
        o[0] = a[1] * p[2] - a[2] * p[1]; o[1] = a[2] * p[0] - a[0] * p[2]; o[2] = a[0] * p[1] - a[1] * p[0];
    };
    float camZ[3], camX[3], camY[3], t[3];
    norm(cam.dir, camZ);
    for (float &c : camZ) c = -c;
    crossf(cam.up, camZ, t);
    norm(t, camX);
    crossf(camZ, camX, camY);
    memcpy(F.cam2world, camX, 12); memcpy(F.cam2world + 3, camY, 12); memcpy(F.cam2world + 6, camZ, 12);
    const float Pi = 3.14159265358979323846f;
    F.half_h = cam.focaldist * tanf((cam.fov * Pi / 180.0f) / 2.0f);
    const float aspect = (float)cam.width / (float)cam.height;
    F.half_w = aspect * F.half_h;
    F.pixel_size = (F.half_w * 2.0f) / (float)cam.width;
    memcpy(F.cam_pos, cam.pos, 12);
    F.focaldist = cam.focaldist; F.dof = cam.dof;
    F.width = cam.width; F.height = cam.height; F.srgb = cam.srgb;
}

float halton(int index, int base) // src/rng.h:103-112
{
    float r = 0;
    float f = 1.0f / (float)base;
    for (int i = index; i > 0; i /= base) {
        r += f * (float)(i % base);
        f /= (float)base;
    }
    return r;
}

int validate_params(rt_ctx *ctx, const rt_render_params *p)
{
    if (!p) return ctx->fail(RT_ERR_INVALID, "null render params");
    if (p->mode != RT_MODE_PATH && p->mode != RT_MODE_WHITTED) return ctx->fail(RT_ERR_UNSUPPORTED, "unknown render mode %d", p->mode);
    if (p->min_spp < 1 || p->max_spp < p->min_spp) return ctx->fail(RT_ERR_INVALID, "need 1 <= min_spp <= max_spp");
    if (p->max_bounces < 0 || p->max_bounces > 65535) return ctx->fail(RT_ERR_INVALID, "max_bounces must be 0..65535");
    if (p->mode == RT_MODE_WHITTED && p->max_bounces > kMaxWhittedBounces)
        return ctx->fail(RT_ERR_INVALID, "the Whitted-style shader takes max_bounces 0..%d (its ray tree grows as 3^bounces)", kMaxWhittedBounces);
    if (p->halton_period < 1 || p->halton_period > 64) return ctx->fail(RT_ERR_INVALID, "halton_period must be 1..64");
    if (p->tile_size < 1) return ctx->fail(RT_ERR_INVALID, "tile_size < 1");
    if (p->mode == RT_MODE_WHITTED && p->photon_flags != 0) {
        if (p->photon_flags & ~(RT_PM_DIRECT | RT_PM_INDIRECT | RT_PM_CAUSTICS | RT_PM_MONTE_CARLO))
            return ctx->fail(RT_ERR_INVALID, "unknown photon_flags bits 0x%x", p->photon_flags);
        if (!(p->gather_radius > 0.0f)) return ctx->fail(RT_ERR_INVALID, "gather_radius must be > 0");
        if ((p->photon_flags & (RT_PM_DIRECT | RT_PM_INDIRECT)) && ctx->pmap_n[RT_MAP_GLOBAL] == 0)
            return ctx->fail(RT_ERR_STATE, "photon_flags ask for the global photon map, which has not been built (rt_photons_fill / rt_photons_build)");
        if ((p->photon_flags & RT_PM_CAUSTICS) && ctx->pmap_n[RT_MAP_CAUSTICS] == 0)
            return ctx->fail(RT_ERR_STATE, "photon_flags ask for the caustics photon map, which has not been built (rt_photons_fill / rt_photons_build)");
        // maps filled by rt_photons_fill hold the light their flags asked for: reading them under other flags would count
        // direct light twice or drop it (maps built from the caller's own photons, pmap_flags -1, are taken on trust)
        if (ctx->pmap_flags != -1 && ctx->pmap_flags != p->photon_flags)
            return ctx->fail(RT_ERR_STATE, "photon maps were filled with flags 0x%x, the frame asks for 0x%x", ctx->pmap_flags, p->photon_flags);
    }
    if (p->tile_world < 1 || p->tile_rank < 0 || p->tile_rank >= p->tile_world)
        return ctx->fail(RT_ERR_INVALID, "need 0 <= tile_rank < tile_world");
    return RT_OK;
}

void fill_frame(const rt_ctx *ctx, const rt_render_params *p, DFrame &F)
{
    memset(&F, 0, sizeof F);
    derive_camera(ctx->camera, F);
    F.min_spp = p->min_spp; F.max_spp = p->max_spp; F.delta_max = p->delta_max;
    F.max_bounces = p->max_bounces; F.bias = p->bias; F.halton_period = p->halton_period; F.seed = p->seed;
    F.photon_flags = p->mode == RT_MODE_WHITTED ? p->photon_flags : 0;
    F.gather_radius = p->gather_radius;
    F.chunk = kChunk;
}

template <bool TEXBG> const void *paths_kernel_lockstep(bool counting, int ray_levels)
{
    if (ray_levels <= 1) return counting ? (const void *)k_paths<true, 1, TEXBG> : (const void *)k_paths<false, 1, TEXBG>;
    if (ray_levels == 2) return counting ? (const void *)k_paths<true, 2, TEXBG> : (const void *)k_paths<false, 2, TEXBG>;
    return counting ? (const void *)k_paths<true, 5, TEXBG> : (const void *)k_paths<false, 5, TEXBG>;
}

// the path kernel whose scene walks skip objects out of the wave's reach: scenes of many objects (not the counting
// variant: its box and triangle tests are the reference's, which looks at every object)
template <bool TEXBG> const void *paths_kernel_culling(int ray_levels)
{
    if (ray_levels <= 1) return (const void *)k_paths<false, 1, TEXBG, true>;
    if (ray_levels == 2) return (const void *)k_paths<false, 2, TEXBG, true>;
    return (const void *)k_paths<false, 5, TEXBG, true>;
}

const void *paths_kernel(bool counting, int ray_levels, bool textured_background, bool culling)
{
    if (culling && !counting) return textured_background ? paths_kernel_culling<true>(ray_levels) : paths_kernel_culling<false>(ray_levels);
    return textured_background ? paths_kernel_lockstep<true>(counting, ray_levels) : paths_kernel_lockstep<false>(counting, ray_levels);
}

const void *whitted_kernel(bool counting, int ray_levels)
{
    if (ray_levels <= 1) return counting ? (const void *)k_whitted<true, 1> : (const void *)k_whitted<false, 1>;
    if (ray_levels == 2) return counting ? (const void *)k_whitted<true, 2> : (const void *)k_whitted<false, 2>;
    return counting ? (const void *)k_whitted<true, 5> : (const void *)k_whitted<false, 5>;
}

// ray_levels 0: the scene is walked through the hierarchy over its objects' boxes (scene_levels() below says when)
const void *rays_kernel(bool counting, int ray_levels)
{
    if (ray_levels == 0) return (const void *)k_rays<false, 0, RayListIO>;
    if (ray_levels <= 1) return counting ? (const void *)k_rays<true, 1, RayListIO> : (const void *)k_rays<false, 1, RayListIO>;
    if (ray_levels == 2) return counting ? (const void *)k_rays<true, 2, RayListIO> : (const void *)k_rays<false, 2, RayListIO>;
    return counting ? (const void *)k_rays<true, 5, RayListIO> : (const void *)k_rays<false, 5, RayListIO>;
}

// rt_trace_brief through the four-wide walk (RT_QUERY_WIDE=1; the counting variant always walks the binary tree)
const void *rays_kernel_wide(int ray_levels)
{
    if (ray_levels == 0) return (const void *)k_rays<false, 0, RayListWideIO>;
    if (ray_levels <= 1) return (const void *)k_rays<false, 1, RayListWideIO>;
    if (ray_levels == 2) return (const void *)k_rays<false, 2, RayListWideIO>;
    return (const void *)k_rays<false, 5, RayListWideIO>;
}

const void *wf_trace_kernel(bool counting, int ray_levels)
{
    if (ray_levels == 0) return (const void *)k_rays<false, 0, WfIO>;
    if (ray_levels <= 1) return counting ? (const void *)k_rays<true, 1, WfIO> : (const void *)k_rays<false, 1, WfIO>;
    if (ray_levels == 2) return counting ? (const void *)k_rays<true, 2, WfIO> : (const void *)k_rays<false, 2, WfIO>;
    return counting ? (const void *)k_rays<true, 5, WfIO> : (const void *)k_rays<false, 5, WfIO>;
}

// which instantiation of the ray-list kernels a scene takes: the one that walks the object hierarchy if the scene has one
// (not the counting variant: its box and triangle tests are the reference's, which looks at every object)
int scene_levels(const rt_ctx *ctx) { return (!ctx->counting && ctx->scene.n_tlas > 0) ? 0 : ctx->ray_levels; }

size_t stack_lds_bytes(const DScene &S) { return (size_t)S.lds_levels * kBlock * 8; }
// kernels that look photons up keep the arrays of found photons of kGatherSlots lanes per wave behind the stacks
size_t gather_lds_bytes(const DScene &S) { return stack_lds_bytes(S) + kHeapLdsBytes; }

} // namespace

// One sampling round of `items` camera paths through the wavefront pipeline (device/wavefront.h): camera rays, then per
// bounce one ray launch (extension rays of the live paths, shadow rays of the previous bounce) and one shading launch, until no path and no shadow ray is left.  The lists' lengths stay
// in device memory, one WfCounts per bounce, where the kernels read them; the host reads them too, after every shading
// launch, to size the next launches and to see when to hand over to k_wf_finish.
static int render_wavefront(rt_ctx *ctx, const DFrame &F, const int32_t *act, uint64_t items64, int k0, int nk,
                            const FramePlanes &planes, size_t lds, double *ray_ms, uint32_t *ray_launches)
{
    hipStream_t s = ctx->stream;
    const uint32_t items = (uint32_t)items64;
    // per-entry arrays of the lists, 256-byte aligned sub-allocations of one buffer: the paths of the list a bounce goes
    // through and of the list it fills (each at most `items` long), the answers to the extension rays, the shadow list
    const size_t sizes[13] = {16, 16, 16, 8, 16, 16, 16, 8, 16, 4, 16, 16, 16};
    size_t offs[13], total = 0;
    for (int i = 0; i < 13; ++i) { offs[i] = total; total += ((size_t)items * sizes[i] + 255) / 256 * 256; }
    HIP_TRY(ctx, ctx->d_wf_state.reserve(total));
    const int n_entries = F.max_bounces + 4;
    HIP_TRY(ctx, ctx->d_wf_counts.reserve((size_t)n_entries * sizeof(WfCounts)));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_wf_counts.p, 0, (size_t)n_entries * sizeof(WfCounts), s));
    WfCounts *const cnt = ctx->d_wf_counts.as<WfCounts>();
    char *base = ctx->d_wf_state.as<char>();
    WfPaths lists[2];
    for (int i = 0; i < 2; ++i) {
        lists[i].ray_o = (float4 *)(base + offs[4 * i]); lists[i].ray_d = (float4 *)(base + offs[4 * i + 1]);
        lists[i].thr = (float4 *)(base + offs[4 * i + 2]); lists[i].rng = (uint2 *)(base + offs[4 * i + 3]);
    }
    WfState P;
    P.in = lists[0]; P.out = lists[1];
    P.hit = (float4 *)(base + offs[8]); P.hit_v = (float *)(base + offs[9]);
    P.sh_o = (float4 *)(base + offs[10]); P.sh_d = (float4 *)(base + offs[11]); P.sh_c = (float4 *)(base + offs[12]);
    P.acc = ctx->d_sbuf.as<float4>();
    // bounce b goes through lists[b & 1] and fills the other
    const auto state_of = [&](int b) { WfState q = P; q.in = lists[b & 1]; q.out = lists[(b & 1) ^ 1]; return q; };

    if (k0 == 0 && (planes.z || planes.object_id)) { // depth / id planes: the first samples' camera rays through the reference's walk
        const uint32_t n_px = items / (uint32_t)nk;
        const void *fp = ctx->ray_levels <= 1 ? (const void *)k_planes<1> : ctx->ray_levels == 2 ? (const void *)k_planes<2> : (const void *)k_planes<5>;
        const dim3 grid((unsigned)std::min<uint64_t>(((uint64_t)n_px + kBlock - 1) / kBlock, ctx->ovf_threads / kBlock));
        DScene p_scene = ctx->scene;
        DFrame p_frame = F;
        const float *p_hal = ctx->d_halton.as<float>();
        uint32_t p_n = n_px;
        FramePlanes p_planes = planes;
        void *pargs[] = {&p_scene, &p_frame, &p_hal, &act, &p_n, &p_planes};
        HIP_TRY(ctx, hipLaunchKernel(fp, grid, dim3(kBlock), pargs, lds, s));
    }
    const dim3 wide((unsigned)std::min<uint64_t>(((uint64_t)items + kBlock - 1) / kBlock, (uint64_t)ctx->n_cu * 32));
    hipLaunchKernelGGL(k_wf_generate, wide, dim3(kBlock), 0, s, F, ctx->d_halton.as<float>(), act, items, k0, nk, P, cnt);
    HIP_TRY(ctx, hipGetLastError());

    const void *fn = wf_trace_kernel(ctx->counting, scene_levels(ctx));
    int per_cu = 1;
    const size_t lds_rays = lds;
    HIP_TRY(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, kBlock, lds_rays));
    per_cu = std::max(1, std::min(per_cu, 8));
    DScene a_scene = ctx->scene;
    DCounters *a_ctr = ctx->d_counters.as<DCounters>();
    unsigned long long *a_queue = ctx->d_queue.as<unsigned long long>();
    int launch = 0;
    // one ray launch over the lists of bounce b (n_rays: what the host knows of their total length, for the grid)
    const auto launch_rays = [&](int b, uint64_t n_rays, bool shadows_only) -> int {
        const dim3 grid((unsigned)std::max<uint64_t>(1, std::min<uint64_t>((n_rays + kBlock - 1) / kBlock, (uint64_t)ctx->n_cu * per_cu)));
        WfIO a_io{state_of(b), cnt + b, shadows_only ? 1u : 0u};
        uint32_t a_n = (uint32_t)b, a_chunk = 0; // the kernel takes the list's length from cnt[b] and sizes the portions itself (the bounce number only serves diagnostic builds)
        void *targs[] = {&a_scene, &a_io, &a_n, &a_ctr, &a_queue, &a_chunk};
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_queue.p, 0, kQueueBytes, s));
        const bool timed = launch < rt_ctx::kMaxRayLaunches;
        if (timed) HIP_TRY(ctx, hipEventRecord(ctx->ray_ev[2 * launch], s));
        HIP_TRY(ctx, hipLaunchKernel(fn, grid, dim3(kBlock), targs, lds_rays, s));
        if (timed) { HIP_TRY(ctx, hipEventRecord(ctx->ray_ev[2 * launch + 1], s)); ++launch; }
        return RT_OK;
    };

    // what this round is, as far as its list lengths go
    uint64_t key = 1469598103934665603ull;
    const auto mix = [&](const void *p, size_t n) { for (size_t i = 0; i < n; ++i) key = (key ^ ((const unsigned char *)p)[i]) * 1099511628211ull; };
    mix(&F, sizeof F); mix(&items, sizeof items); mix(&k0, sizeof k0); mix(&nk, sizeof nk); mix(&act, sizeof act);
    mix(&ctx->scene_generation, sizeof ctx->scene_generation); mix(&ctx->finish_below, sizeof ctx->finish_below);
    mix(ctx->active_key, sizeof ctx->active_key); // which pixels: image size, tiles, shard
    mix(&ctx->counting, sizeof ctx->counting);     // (the counting variant walks the reference's binary tree: a hit an ulp away can end a path elsewhere)
    const rt_ctx::WfProfile *known = nullptr;
    if (ctx->wf_replay) for (const auto &pr : ctx->wf_profiles) if (pr.key == key) known = &pr;
    std::vector<WfCounts> seen; // cnt[0..] as the host learns them (or takes them from the profile)
    static const bool wf_debug = getenv("RT_WF_DEBUG") != nullptr;
    if (wf_debug) fprintf(stderr, "wavefront round: key %016llx %s\n", (unsigned long long)key, known ? "replayed from its profile" : "new");

    for (int attempt = 0; attempt < 2; ++attempt) {
    const bool replay = known != nullptr;
    seen.assign(1, WfCounts{items, 0u, {0u, 0u}});
    if (attempt > 0) { // the device's lists were not what the profile said: once more, waiting for every bounce's counts
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_wf_counts.p, 0, (size_t)n_entries * sizeof(WfCounts), s));
        hipLaunchKernelGGL(k_wf_generate, wide, dim3(kBlock), 0, s, F, ctx->d_halton.as<float>(), act, items, k0, nk, P, cnt);
        HIP_TRY(ctx, hipGetLastError());
    }
    WfCounts h = seen[0]; // the host's idea of cnt[b]
    bool finishing = false;
    for (int b = 0;; ++b) {
        const uint64_t n_live = h.n_ext;
        if (finishing) { // the shadow rays of the last shaded bounce, then the remaining paths to their ends in one launch
            if (h.n_shadow > 0) if (int rc = launch_rays(b, h.n_shadow, true)) return rc;
            const void *ff = ctx->counting ? (ctx->ray_levels <= 1 ? (const void *)k_wf_finish<true, 1> : ctx->ray_levels == 2 ? (const void *)k_wf_finish<true, 2> : (const void *)k_wf_finish<true, 5>)
                                           : (scene_levels(ctx) == 0 ? (const void *)k_wf_finish<false, 0> : ctx->ray_levels <= 1 ? (const void *)k_wf_finish<false, 1> : ctx->ray_levels == 2 ? (const void *)k_wf_finish<false, 2> : (const void *)k_wf_finish<false, 5>);
            // a pair of lanes per path
            const dim3 grid((unsigned)std::min<uint64_t>((n_live * 2 + kBlock - 1) / kBlock, ctx->ovf_threads / kBlock));
            DFrame f_frame = F;
            int f_k0 = k0, f_nk = nk;
            const WfCounts *f_cnt = cnt + b;
            HIP_TRY(ctx, hipMemsetAsync(ctx->d_queue.p, 0, kQueueBytes, s));
            WfState f_state = state_of(b);
            void *fargs[] = {&a_scene, &f_frame, &f_k0, &f_nk, &f_state, &f_cnt, &a_ctr, &a_queue};
            HIP_TRY(ctx, hipLaunchKernel(ff, grid, dim3(kBlock), fargs, lds, s));
            break;
        }
        if (n_live + h.n_shadow == 0) break;
        if (int rc = launch_rays(b, n_live + h.n_shadow, false)) return rc;
        if (n_live == 0) break; // those were the last shadow rays
        {
            const dim3 grid((unsigned)std::min<uint64_t>((n_live + kShadeChunk - 1) / kShadeChunk, (uint64_t)ctx->n_cu * 16));
            if (b == 0) hipLaunchKernelGGL(k_wf_shade<true>, grid, dim3(kBlock), 0, s, a_scene, F, act, k0, nk, state_of(b), cnt + b, cnt + b + 1, planes);
            else hipLaunchKernelGGL(k_wf_shade<false>, grid, dim3(kBlock), 0, s, a_scene, F, act, k0, nk, state_of(b), cnt + b, cnt + b + 1, planes);
            HIP_TRY(ctx, hipGetLastError());
        }
        if (replay && (size_t)b + 1 < known->counts.size()) h = known->counts[b + 1];
        else if (replay) h = WfCounts{0u, 0u, {0u, 0u}}; // (the profile ends here; the check below tells if the device's lists do too)
        else {
            // the next bounce's list lengths, into pinned memory (a pageable destination goes through a staging copy)
            volatile uint32_t *got = (volatile uint32_t *)&ctx->host_words[2];
            HIP_TRY(ctx, hipMemcpyAsync((void *)got, cnt + b + 1, sizeof(WfCounts), hipMemcpyDeviceToHost, s));
            HIP_TRY(ctx, hipStreamSynchronize(s));
            h = WfCounts{got[0], got[1], {0u, 0u}};
            ctx->prog_round_done.store(items - std::min<uint64_t>(items, h.n_ext)); // paths that have ended
            if (cancelled(ctx)) break; // the round's kernels hand out no more rays; rt_render reports the cancellation
        }
        seen.push_back(h);
        // few paths left: their shadow rays next (a ray launch without extension rays), then k_wf_finish
        if (h.n_ext > 0 && (h.n_ext <= ctx->finish_below || b + 3 >= n_entries)) finishing = true;
    }
    if (!replay) break;
    // launched from the profile: are the device's list lengths the profile's?  (They are, for the same frame again: a
    // path's fate does not depend on when it is traced.  A frame that only looks the same is done again.)
    std::vector<WfCounts> dev(seen.size() + 1);
    HIP_TRY(ctx, hipMemcpyAsync(dev.data(), cnt, std::min(dev.size(), (size_t)n_entries) * sizeof(WfCounts), hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    bool same = true;
    for (size_t i = 0; i < seen.size(); ++i) same = same && dev[i].n_ext == seen[i].n_ext && dev[i].n_shadow == seen[i].n_shadow;
    if (same) break;
    if (wf_debug) fprintf(stderr, "wavefront round: the profile did not hold, once more\n");
    known = nullptr;
    launch = 0;
    }
    HIP_TRY(ctx, hipStreamSynchronize(s));
    ctx->prog_round_done.store(items);
    if (!known && !cancelled(ctx) && ctx->wf_replay) { // remember what this round went through
        constexpr size_t kProfiles = 32;
        rt_ctx::WfProfile *pr = nullptr;
        for (auto &have : ctx->wf_profiles) if (have.key == key) pr = &have; // (one that did not hold: replaced)
        if (!pr && ctx->wf_profiles.size() < kProfiles) { ctx->wf_profiles.emplace_back(); pr = &ctx->wf_profiles.back(); }
        if (!pr) { pr = &ctx->wf_profiles[ctx->wf_profile_next]; ctx->wf_profile_next = (ctx->wf_profile_next + 1) % kProfiles; }
        pr->key = key; pr->counts = seen;
    }
    for (int i = 0; i < launch; ++i) {
        float ms = 0;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ray_ev[2 * i], ctx->ray_ev[2 * i + 1]));
        *ray_ms += ms;
    }
    *ray_launches += (uint32_t)launch;
    return RT_OK;
}

// everything rt_init created, whatever it got to before it failed
static void destroy_ctx(rt_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) { (void)hipStreamSynchronize(ctx->stream); (void)hipStreamDestroy(ctx->stream); }
    if (ctx->host_words) (void)hipHostFree(ctx->host_words);
    for (hipEvent_t ev : ctx->ray_ev) if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : {ctx->ev0, ctx->ev1, ctx->evk0, ctx->evk1}) if (ev) (void)hipEventDestroy(ev);
    delete ctx;
}

// start of a call that rt_cancel can stop: the flag of the previous call is cleared, on the host and on the device
static int begin_cancellable(rt_ctx *ctx)
{
    ctx->cancel_req.store(0);
    __atomic_store_n(&ctx->host_words[0], 0ull, __ATOMIC_RELEASE); // nothing of this context runs on the GPU between calls
    __atomic_store_n(&ctx->host_words[1], 0ull, __ATOMIC_RELEASE);
    return RT_OK;
}

extern "C" {

const char *rt_last_error(const rt_ctx *ctx) { return ctx ? ctx->error.c_str() : g_init_error.c_str(); }

int rt_init(int device, rt_ctx **out)
{
    if (!out) { g_init_error = "null output pointer"; return RT_ERR_INVALID; }
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) {
        g_init_error = std::string("no HIP device available: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
        return RT_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n) { g_init_error = "device ordinal out of range"; return RT_ERR_INVALID; }
    auto *ctx = new (std::nothrow) rt_ctx();
    if (!ctx) { g_init_error = "out of host memory"; return RT_ERR_INVALID; }
    ctx->device = device;
    auto bail = [&](const char *what, hipError_t err) {
        g_init_error = std::string(what) + ": " + hipGetErrorString(err);
        destroy_ctx(ctx);
        return RT_ERR_HIP;
    };
    if ((e = hipSetDevice(device)) != hipSuccess) return bail("hipSetDevice", e);
    if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    for (hipEvent_t *ev : {&ctx->ev0, &ctx->ev1, &ctx->evk0, &ctx->evk1})
        if ((e = hipEventCreate(ev)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipGetSymbolAddress((void **)&ctx->static_scratch, HIP_SYMBOL(scratch))) != hipSuccess) return bail("hipGetSymbolAddress", e);
    if ((e = ctx->d_counters.reserve(sizeof(DCounters))) != hipSuccess) return bail("hipMalloc", e);
    if ((e = ctx->d_next_count.reserve(sizeof(uint32_t))) != hipSuccess) return bail("hipMalloc", e);
    // per-lane BVH stacks may need more than the default 64 KiB of LDS per workgroup
    for (const void *k : {(const void *)k_rays<false, 1, RayListIO>, (const void *)k_rays<false, 2, RayListIO>, (const void *)k_rays<false, 5, RayListIO>,
                          (const void *)k_rays<true, 1, RayListIO>, (const void *)k_rays<true, 2, RayListIO>, (const void *)k_rays<true, 5, RayListIO>,
                          (const void *)k_paths<false, 1, false>, (const void *)k_paths<false, 2, false>, (const void *)k_paths<false, 5, false>,
                          (const void *)k_paths<true, 1, false>, (const void *)k_paths<true, 2, false>, (const void *)k_paths<true, 5, false>,
                          (const void *)k_paths<false, 1, true>, (const void *)k_paths<false, 2, true>, (const void *)k_paths<false, 5, true>,
                          (const void *)k_paths<true, 1, true>, (const void *)k_paths<true, 2, true>, (const void *)k_paths<true, 5, true>,
                          (const void *)k_paths<false, 1, false, true>, (const void *)k_paths<false, 2, false, true>, (const void *)k_paths<false, 5, false, true>,
                          (const void *)k_paths<false, 1, true, true>, (const void *)k_paths<false, 2, true, true>, (const void *)k_paths<false, 5, true, true>,
                          (const void *)k_rays<false, 1, WfIO>, (const void *)k_rays<false, 2, WfIO>, (const void *)k_rays<false, 5, WfIO>,
                          (const void *)k_rays<true, 1, WfIO>, (const void *)k_rays<true, 2, WfIO>, (const void *)k_rays<true, 5, WfIO>,
                          (const void *)k_whitted<false, 1>, (const void *)k_whitted<false, 2>, (const void *)k_whitted<false, 5>,
                          (const void *)k_whitted<true, 1>, (const void *)k_whitted<true, 2>, (const void *)k_whitted<true, 5>,
                          (const void *)k_probe<1>, (const void *)k_probe<2>, (const void *)k_probe<5>,
                          (const void *)k_planes<1>, (const void *)k_planes<2>, (const void *)k_planes<5>,
                          (const void *)k_wf_finish<false, 1>, (const void *)k_wf_finish<false, 2>, (const void *)k_wf_finish<false, 5>,
                          (const void *)k_wf_finish<true, 1>, (const void *)k_wf_finish<true, 2>, (const void *)k_wf_finish<true, 5>,
                          (const void *)k_order_keys<1>, (const void *)k_order_keys<2>, (const void *)k_order_keys<5>,
                          (const void *)k_trace, (const void *)k_trace_shadow, (const void *)k_trace_paths})
        (void)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxDynamicLds);
    for (const void *k : {(const void *)k_rays<false, 0, RayListIO>, (const void *)k_rays<false, 0, RayListWideIO>, (const void *)k_rays<false, 0, WfIO>,
                          (const void *)k_rays<false, 0, BriefIO>, (const void *)k_wf_finish<false, 0>})
        (void)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxDynamicLds);
    for (const void *k : {(const void *)k_rays<false, 1, BriefIO>, (const void *)k_rays<false, 2, BriefIO>, (const void *)k_rays<false, 5, BriefIO>,
                          (const void *)k_rays<false, 1, RayListWideIO>, (const void *)k_rays<false, 2, RayListWideIO>, (const void *)k_rays<false, 5, RayListWideIO>})
        (void)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxDynamicLds);
    for (const void *k : {(const void *)k_photon_paths<false, 1>, (const void *)k_photon_paths<false, 2>, (const void *)k_photon_paths<false, 5>,
                          (const void *)k_photon_paths<true, 1>, (const void *)k_photon_paths<true, 2>, (const void *)k_photon_paths<true, 5>,
                          (const void *)k_photon_irradiance})
        (void)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxDynamicLds);
    (void)hipGetLastError(); // (a refused attribute must not show up as the error of a later launch)
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) return bail("hipGetDeviceProperties", e);
    ctx->n_cu = prop.multiProcessorCount;
    ctx->finish_below = (uint32_t)ctx->n_cu * 512u;
    if (const char *e9 = getenv("RT_FINISH_BELOW")) ctx->finish_below = (uint32_t)std::max(0, atoi(e9));
    if (const char *e3 = getenv("RT_PIPELINE")) ctx->pipeline = strcmp(e3, "wave") == 0 ? 1 : strcmp(e3, "mega") == 0 ? 0 : -1;
    for (hipEvent_t &ev : ctx->ray_ev)
        if ((e = hipEventCreate(&ev)) != hipSuccess) return bail("hipEventCreate", e);
    if (const char *e8 = getenv("RT_CHUNK_ITEMS")) ctx->chunk_items = std::max(0, atoi(e8)) / 64 * 64;
    if (const char *e7 = getenv("RT_ORDER")) ctx->order_pixels = atoi(e7) != 0 ? 1 : 0;
    if (const char *e11 = getenv("RT_GPU_BVH")) ctx->gpu_bvh = atoi(e11) != 0;
    if (const char *e12 = getenv("RT_QUERY_WIDE")) ctx->query_wide = atoi(e12) != 0;
    if (const char *e14 = getenv("RT_WIDE_GPU_BVH")) ctx->wide_from_gpu = atoi(e14) != 0;
    if (const char *e13 = getenv("RT_WF_REPLAY")) ctx->wf_replay = atoi(e13) != 0;
    if (const char *e10 = getenv("RT_TLAS_MIN")) ctx->tlas_min_objects = std::max(0, atoi(e10));
    if (const char *e2 = getenv("RT_LDS_LEVELS")) ctx->lds_levels_max = std::max(1, std::min(64, atoi(e2))); // tuning knob
    if ((e = ctx->d_queue.reserve(kQueueBytes)) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipHostMalloc((void **)&ctx->host_words, 64, hipHostMallocMapped | hipHostMallocCoherent)) != hipSuccess) return bail("hipHostMalloc", e);
    memset(ctx->host_words, 0, 64);
    *out = ctx;
    return RT_OK;
}

void rt_shutdown(rt_ctx *ctx) { destroy_ctx(ctx); }

int rt_cancel(rt_ctx *ctx)
{
    if (!ctx) return RT_ERR_INVALID;
    // no HIP call and no error text here: both belong to the thread inside the call being cancelled
    ctx->cancel_req.store(1);
    __atomic_store_n(&ctx->host_words[0], 1ull, __ATOMIC_RELEASE);
    return RT_OK;
}

int rt_progress(rt_ctx *ctx, rt_progress_info *out)
{
    if (!ctx || !out) return RT_ERR_INVALID;
    out->pixels_total = ctx->prog_pixels_total.load(); out->pixels_done = ctx->prog_pixels_done.load();
    out->round_items = ctx->prog_round_items.load(); out->round_items_done = ctx->prog_round_done.load();
    out->round = ctx->prog_round.load(); out->rendering = ctx->prog_rendering.load();
    if (out->rendering && ctx->prog_live_queue.load()) // a lock-step kernel is running: it reports its queue position
        out->round_items_done = std::min<uint64_t>(__atomic_load_n(&ctx->host_words[1], __ATOMIC_ACQUIRE), out->round_items);
    return RT_OK;
}

int rt_set_counting(rt_ctx *ctx, int on)
{
    if (!ctx) return RT_ERR_INVALID;
    ctx->counting = on != 0;
    return RT_OK;
}

// One mesh's BVH built on the GPU (device/lbvh.h), in pool numbering: binary nodes from pool id node_base (the root is
// node_base + 1), four-wide records from rec_base, triangle records from tri_base.
struct GpuBvh {
    std::unique_ptr<DeviceBuf> nodes, recs, tris;
    uint32_t n_nodes = 0, n_recs = 0; // pool entries (binary nodes incl. the unused first two; records)
    uint32_t root_word = 0, root4 = 0, depth = 0, wide_levels = 0; // levels of the binary tree, of the four-wide one
    float bounds[6] = {0, 0, 0, 0, 0, 0};
    double ms = 0;
};

static int build_bvh_gpu(rt_ctx *ctx, const rt_mesh &m, uint32_t node_base, uint32_t rec_base, uint32_t tri_base, uint32_t max_leaf, GpuBvh &out)
{
    hipStream_t s = ctx->stream;
    const uint32_t n = m.nf;
    out.nodes = std::make_unique<DeviceBuf>(); out.recs = std::make_unique<DeviceBuf>(); out.tris = std::make_unique<DeviceBuf>();
    // bounds of the triangles = of the vertices the faces name
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (size_t k = 0; k < (size_t)n * 3; ++k) {
        const float *q = m.v + (size_t)m.f[k] * 3;
        for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], q[a]); hi[a] = std::max(hi[a], q[a]); }
    }
    for (int a = 0; a < 3; ++a) { out.bounds[a] = lo[a]; out.bounds[3 + a] = hi[a]; }
    DeviceBuf d_v, d_f, keys[2], order[2], child, parent, leaf_parent, range, ibox, lbox, arrived, inner, inner_rank, front, level_count, maxd, tmp;
    DeviceBuf unit, unit_count, unit_first, run_start, tsize, toff;
    HIP_TRY(ctx, d_v.reserve((size_t)m.nv * 12)); HIP_TRY(ctx, d_f.reserve((size_t)n * 12));
    HIP_TRY(ctx, hipMemcpyAsync(d_v.p, m.v, (size_t)m.nv * 12, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(d_f.p, m.f, (size_t)n * 12, hipMemcpyHostToDevice, s));
    const size_t ni = n > 1 ? n - 1 : 1;
    for (auto &b : keys) HIP_TRY(ctx, b.reserve((size_t)n * 8));
    for (auto &b : order) HIP_TRY(ctx, b.reserve((size_t)n * 4));
    HIP_TRY(ctx, child.reserve(ni * 8)); HIP_TRY(ctx, parent.reserve(ni * 4)); HIP_TRY(ctx, leaf_parent.reserve((size_t)n * 4));
    HIP_TRY(ctx, range.reserve(ni * 8)); HIP_TRY(ctx, ibox.reserve(ni * 24)); HIP_TRY(ctx, lbox.reserve((size_t)n * 24));
    HIP_TRY(ctx, arrived.reserve(ni * 4)); HIP_TRY(ctx, inner.reserve(ni * 4)); HIP_TRY(ctx, inner_rank.reserve(ni * 4));
    HIP_TRY(ctx, maxd.reserve(4));
    HIP_TRY(ctx, hipMemsetAsync(maxd.p, 0, 4, s));
    HIP_TRY(ctx, hipEventRecord(ctx->evk0, s));
    LbvhArrays A{};
    A.v = d_v.as<float>(); A.f = d_f.as<uint32_t>(); A.n_faces = n; A.n = n; A.max_leaf = max_leaf;
    HIP_TRY(ctx, unit.reserve((size_t)n * 4)); HIP_TRY(ctx, tsize.reserve((size_t)n * 4)); HIP_TRY(ctx, toff.reserve((size_t)n * 4));
    A.unit = unit.as<uint32_t>(); A.tsize = tsize.as<uint32_t>(); A.toff = toff.as<uint32_t>();
    A.keys = keys[0].as<unsigned long long>(); A.order = order[0].as<uint32_t>();
    A.child = child.as<uint2>(); A.parent = parent.as<uint32_t>(); A.leaf_parent = leaf_parent.as<uint32_t>(); A.range = range.as<uint2>();
    A.ibox = ibox.as<float>(); A.lbox = lbox.as<float>(); A.arrived = arrived.as<uint32_t>(); A.inner = inner.as<uint32_t>();
    A.max_depth = maxd.as<uint32_t>();
    const dim3 grid((unsigned)std::min<size_t>(((size_t)n + kBlock - 1) / kBlock, (size_t)ctx->n_cu * 16)), block(kBlock);
    const auto scale = [&](int a) { return hi[a] > lo[a] ? 2097151.0f / (hi[a] - lo[a]) : 0.0f; };
    const float extent[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
    LbvhAxes axes = lbvh_axis_order(extent);
    if (getenv("RT_LBVH_PLAIN_MORTON")) { // (for comparison: x, y, z in turn)
        axes = LbvhAxes{{0ull, 0ull}};
        for (int i = 0; i < 63; ++i) axes.w[i / 32] |= (unsigned long long)(i % 3) << (2 * (i % 32));
    }
    // an exclusive prefix sum of `count` values and its total
    const auto prefix_sum = [&](uint32_t *values, uint32_t *sums, uint32_t count, uint32_t &total) -> int {
        size_t bytes = 0;
        HIP_TRY(ctx, rocprim::exclusive_scan(nullptr, bytes, values, sums, 0u, (size_t)count, rocprim::plus<uint32_t>(), s));
        HIP_TRY(ctx, tmp.reserve(std::max<size_t>(1, bytes)));
        HIP_TRY(ctx, rocprim::exclusive_scan(tmp.p, bytes, values, sums, 0u, (size_t)count, rocprim::plus<uint32_t>(), s));
        uint32_t last[2] = {0, 0};
        HIP_TRY(ctx, hipMemcpyAsync(&last[0], values + (count - 1), 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(ctx, hipMemcpyAsync(&last[1], sums + (count - 1), 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(ctx, hipStreamSynchronize(s));
        total = last[0] + last[1];
        return RT_OK;
    };
    // the units the tree is built over: pairs of consecutive faces that share an edge, and the faces that are left (device/lbvh.h)
    uint32_t nu = 0;
    HIP_TRY(ctx, unit_count.reserve((size_t)n * 4)); HIP_TRY(ctx, unit_first.reserve((size_t)n * 4)); HIP_TRY(ctx, run_start.reserve((size_t)n * 4));
    hipLaunchKernelGGL(k_lbvh_run_marks, grid, block, 0, s, A.f, n, unit_count.as<uint32_t>());
    {
        size_t bytes = 0;
        HIP_TRY(ctx, rocprim::inclusive_scan(nullptr, bytes, unit_count.as<uint32_t>(), run_start.as<uint32_t>(), (size_t)n, rocprim::maximum<uint32_t>(), s));
        HIP_TRY(ctx, tmp.reserve(std::max<size_t>(1, bytes)));
        HIP_TRY(ctx, rocprim::inclusive_scan(tmp.p, bytes, unit_count.as<uint32_t>(), run_start.as<uint32_t>(), (size_t)n, rocprim::maximum<uint32_t>(), s));
    }
    hipLaunchKernelGGL(k_lbvh_unit_heads, grid, block, 0, s, A.f, n, run_start.as<uint32_t>(), unit_count.as<uint32_t>());
    if (int rc = prefix_sum(unit_count.as<uint32_t>(), unit_first.as<uint32_t>(), n, nu)) return rc;
    hipLaunchKernelGGL(k_lbvh_units, grid, block, 0, s, A.f, n, run_start.as<uint32_t>(), unit_count.as<uint32_t>(), unit_first.as<uint32_t>(), A.unit);
    A.n = nu;
    hipLaunchKernelGGL(k_lbvh_keys, grid, block, 0, s, A, lo[0], lo[1], lo[2], scale(0), scale(1), scale(2), axes);
    size_t tmp_bytes = 0;
    HIP_TRY(ctx, rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys[0].as<unsigned long long>(), keys[1].as<unsigned long long>(),
                                           order[0].as<uint32_t>(), order[1].as<uint32_t>(), (size_t)nu, 0u, 63u, s));
    HIP_TRY(ctx, tmp.reserve(std::max<size_t>(1, tmp_bytes)));
    HIP_TRY(ctx, rocprim::radix_sort_pairs(tmp.p, tmp_bytes, keys[0].as<unsigned long long>(), keys[1].as<unsigned long long>(),
                                           order[0].as<uint32_t>(), order[1].as<uint32_t>(), (size_t)nu, 0u, 63u, s));
    A.keys = keys[1].as<unsigned long long>(); A.order = order[1].as<uint32_t>();
    hipLaunchKernelGGL(k_lbvh_unit_sizes, grid, block, 0, s, A);
    uint32_t n_placed = 0;
    if (int rc = prefix_sum(A.tsize, A.toff, nu, n_placed)) return rc;
    if (n_placed != n) return ctx->fail(RT_ERR_HIP, "BVH build: %u of %u triangles placed", n_placed, n);
    LbvhOut O{};
    O.node_base = node_base; O.rec_base = rec_base; O.tri_base = tri_base;
    HIP_TRY(ctx, out.tris->reserve((size_t)n * 48));
    O.tris = out.tris->as<float4>();
    uint32_t T = 0, R = 0, deepest = 0;
    if (n > max_leaf) {
        hipLaunchKernelGGL(k_lbvh_hierarchy, grid, block, 0, s, A);
        hipLaunchKernelGGL(k_lbvh_fit, grid, block, 0, s, A);
        hipLaunchKernelGGL(k_lbvh_mark_inner, grid, block, 0, s, A);
        if (int rc = prefix_sum(inner.as<uint32_t>(), inner_rank.as<uint32_t>(), nu - 1, T)) return rc;
        hipLaunchKernelGGL(k_lbvh_depth, grid, block, 0, s, A, inner.as<uint32_t>());
        HIP_TRY(ctx, hipMemcpyAsync(&deepest, maxd.p, 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(ctx, hipStreamSynchronize(s));
        // every kept inner node could be the root of a record; the wide tree has at most as many levels as the binary one
        if ((uint64_t)node_base + 2 + 2ull * T >= (1ull << 31) || (uint64_t)rec_base + T >= (1ull << 31)) return ctx->fail(RT_ERR_UNSUPPORTED, "scene has too many BVH nodes");
        out.n_nodes = 2 + 2 * T;
        HIP_TRY(ctx, out.nodes->reserve((size_t)out.n_nodes * 32)); HIP_TRY(ctx, out.recs->reserve((size_t)T * 64));
        HIP_TRY(ctx, hipMemsetAsync(out.nodes->p, 0, 64, s));
        O.nodes = out.nodes->as<float4>(); O.recs = out.recs->as<float4>();
        hipLaunchKernelGGL(k_lbvh_emit_binary, grid, block, 0, s, A, inner.as<uint32_t>(), inner_rank.as<uint32_t>(), O);
        std::vector<uint32_t> per_level(deepest + 2, 0u);
        per_level[0] = 1; // the root's record
        HIP_TRY(ctx, front.reserve((size_t)T * 4)); HIP_TRY(ctx, level_count.reserve(per_level.size() * 4));
        HIP_TRY(ctx, hipMemsetAsync(front.p, 0, 4, s)); // ... whose node is the root
        HIP_TRY(ctx, hipMemcpyAsync(level_count.p, per_level.data(), per_level.size() * 4, hipMemcpyHostToDevice, s));
        for (uint32_t level = 0; level < deepest; ++level)
            hipLaunchKernelGGL(k_lbvh_wide_level, grid, block, 0, s, A, inner.as<uint32_t>(), inner_rank.as<uint32_t>(), front.as<uint32_t>(),
                               level_count.as<uint32_t>(), level, O);
        HIP_TRY(ctx, hipMemcpyAsync(per_level.data(), level_count.p, per_level.size() * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(ctx, hipStreamSynchronize(s));
        for (uint32_t c : per_level) { R += c; out.wide_levels += c ? 1u : 0u; }
        out.n_recs = R;
        out.root_word = node_base + 2; // the pair of the root's children: the root is the first kept inner node
        out.root4 = rec_base;          // ... and the first record
    } else { // a handful of triangles: the root is a leaf
        out.n_nodes = 2; out.n_recs = 0;
        HIP_TRY(ctx, out.nodes->reserve(64)); HIP_TRY(ctx, out.recs->reserve(1));
        out.root_word = 0x80000000u | ((n - 1) << 28) | tri_base;
        out.root4 = out.root_word;
        deepest = 0;
        const float4 root[4] = {make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), make_float4(lo[0], hi[0], lo[1], hi[1]),
                                make_float4(lo[2], hi[2], __builtin_bit_cast(float, out.root_word), 0.0f)};
        HIP_TRY(ctx, hipMemcpyAsync(out.nodes->p, root, sizeof root, hipMemcpyHostToDevice, s));
    }
    hipLaunchKernelGGL(k_lbvh_emit_tris, grid, block, 0, s, A, O);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ctx->evk1, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    float ms = 0;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->evk0, ctx->evk1));
    out.ms = ms;
    out.depth = deepest;
    return RT_OK;
}

int rt_scene_upload(rt_ctx *ctx, const rt_scene_desc *sc)
{
    if (!ctx) return RT_ERR_INVALID;
    if (!sc || !sc->nodes || sc->n_nodes == 0) return ctx->fail(RT_ERR_INVALID, "scene has no nodes");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->have_scene = false;

    // ---- validate: the kernels index with what is checked here ----
    const uint32_t nn = sc->n_nodes;
    std::vector<DNode> nodes(nn);
    int max_mesh_depth = -1, max_wide_levels = 0; // levels of the deepest binary BVH, of the deepest four-wide one
    int ray_levels = 1;
    for (uint32_t i = 0; i < nn; ++i) {
        const rt_node &n = sc->nodes[i];
        if ((i == 0 && n.parent != -1) || (i > 0 && (n.parent < 0 || n.parent >= (int)i)))
            return ctx->fail(RT_ERR_INVALID, "node %u: nodes must be in depth-first order with parent < index", i);
        DNode &d = nodes[i];
        memset(&d, 0, sizeof d);
        memcpy(d.itm, n.itm, sizeof d.itm); memcpy(d.tm, n.tm, sizeof d.tm);
        d.parent = n.parent; d.prim = n.prim; d.mesh = n.mesh; d.material = n.material;
        d.depth = i == 0 ? 0 : nodes[n.parent].depth + 1;
        if (d.depth >= kMaxSceneDepth)
            return ctx->fail(RT_ERR_UNSUPPORTED, "node %u: hierarchy deeper than %d levels", i, kMaxSceneDepth);
        if (i > 0) {
            // depth-first order: the parent must be the previous node or one of its ancestors
            int a = (int)i - 1;
            while (a >= 0 && a != n.parent) a = nodes[a].parent;
            if (a != n.parent) return ctx->fail(RT_ERR_INVALID, "node %u: not in depth-first order", i);
            nodes[n.parent].has_children = 1;
            ray_levels = std::max(ray_levels, nodes[n.parent].depth + 1);
            memcpy(d.chain, nodes[n.parent].chain, sizeof d.chain);
        }
        d.chain[d.depth] = (int)i;
        if (n.prim < RT_PRIM_NONE || n.prim > RT_PRIM_MESH) return ctx->fail(RT_ERR_INVALID, "node %u: unknown primitive %d", i, n.prim);
        if (n.prim != RT_PRIM_NONE) {
            // the reference casts the node's material unchecked (src/main.cpp:212): a
            // missing one is undefined there and refused here
            if (n.material < 0 || n.material >= (int)sc->n_materials)
                return ctx->fail(RT_ERR_UNSUPPORTED, "node %u: object node without a valid material", i);
        }
        if (n.prim == RT_PRIM_MESH) {
            if (n.mesh < 0 || n.mesh >= (int)sc->n_meshes) return ctx->fail(RT_ERR_INVALID, "node %u: mesh index out of range", i);
            // a mesh with sub-materials: the node's material is the first of that many consecutive ones
            const rt_mesh &m = sc->meshes[n.mesh];
            if (sc->face_materials && m.face_material && (uint64_t)n.material + m.n_sub_materials > sc->n_materials)
                return ctx->fail(RT_ERR_INVALID, "node %u: the mesh names %u sub-materials but only %u materials follow the node's", i,
                                 m.n_sub_materials, sc->n_materials - (uint32_t)n.material);
        }
    }
    if (sc->n_lights < 1 || !sc->lights || sc->lights[0].type != RT_LIGHT_POINT || !(sc->lights[0].size > 0.0f))
        return ctx->fail(RT_ERR_UNSUPPORTED, "the path integrator needs lights[0] to be a point light with size > 0 (src/main.cpp:165)");

    // ---- meshes: BVH nodes and triangle records of all meshes go into two pools; ids in the data words become pool ids ----
    std::vector<DMesh> dmeshes(sc->n_meshes);
    bool any_face_materials = false;
    std::vector<std::unique_ptr<DeviceBuf>> bufs;
    auto upload = [&](const void *src, size_t bytes, const void **dst) -> hipError_t {
        auto b = std::make_unique<DeviceBuf>();
        hipError_t e = b->reserve(bytes);
        if (e != hipSuccess) return e;
        if (bytes) e = hipMemcpy(b->p, src, bytes, hipMemcpyHostToDevice);
        *dst = b->p;
        bufs.push_back(std::move(b));
        return e;
    };
    std::vector<uint32_t> node_base(sc->n_meshes), tri_base(sc->n_meshes);
    size_t pool_nodes = 2, pool_tris = 0; // pool node 0/1 unused: pairs of children start at even ids
    // A scene whose meshes come without a BVH (or RT_GPU_BVH=1) gets them built here, on the GPU (device/lbvh.h)
    bool gpu_bvh = ctx->gpu_bvh;
    for (uint32_t mi = 0; mi < sc->n_meshes; ++mi) gpu_bvh = gpu_bvh || !sc->meshes[mi].bvh_nodes || !sc->meshes[mi].bvh_elements;
    // A scene that brings the reference's BVH keeps it for everything that has to give the reference's answers and
    // counts (depth / id planes, query entry points, the counting variant).  The frames' extension and shadow rays,
    // which go through the four-wide walk and are held to the image tolerance, get a tree of their own, built here on
    // the GPU: it holds the same triangles under the same test, and is the better tree (device/lbvh.h: frames of the
    // 1M-triangle scene take 7 % less through it than through the reference's tree collapsed to four-wide records).
    const bool wide_gpu = !gpu_bvh && ctx->wide_from_gpu && sc->n_meshes > 0;
    std::vector<GpuBvh> built(gpu_bvh || wide_gpu ? sc->n_meshes : 0);
    size_t pool_recs = 0;
    double build_ms = 0;
    for (uint32_t mi = 0; mi < sc->n_meshes; ++mi) {
        const rt_mesh &m = sc->meshes[mi];
        if (!m.v || !m.vn || !m.vt || !m.f || !m.fn || !m.ft || m.nf == 0 || (!gpu_bvh && (!m.bvh_nodes || !m.bvh_elements || m.n_bvh_nodes < 2)))
            return ctx->fail(RT_ERR_INVALID, "mesh %u: missing arrays (positions, normals, tex coords and faces are all required)", mi);
        node_base[mi] = (uint32_t)pool_nodes; tri_base[mi] = (uint32_t)pool_tris;
        if (gpu_bvh) {
            if (m.nf >= (1u << 28)) return ctx->fail(RT_ERR_UNSUPPORTED, "scene has more than 2^28 triangles");
            for (size_t k = 0; k < (size_t)m.nf * 3; ++k)
                if (m.f[k] >= m.nv) return ctx->fail(RT_ERR_INVALID, "mesh %u: face index out of range", mi);
            if (int rc = build_bvh_gpu(ctx, m, node_base[mi], (uint32_t)pool_recs, tri_base[mi], kLbvhMaxLeafTris, built[mi])) return rc;
            pool_nodes += built[mi].n_nodes;
            pool_recs += built[mi].n_recs;
            build_ms += built[mi].ms;
        } else {
            pool_nodes += ((size_t)m.n_bvh_nodes + 1) / 2 * 2;
            if (wide_gpu) {
                if (m.nf >= (1u << 28)) return ctx->fail(RT_ERR_UNSUPPORTED, "scene has more than 2^28 triangles");
                for (size_t k = 0; k < (size_t)m.nf * 3; ++k)
                    if (m.f[k] >= m.nv) return ctx->fail(RT_ERR_INVALID, "mesh %u: face index out of range", mi);
                if (int rc = build_bvh_gpu(ctx, m, 0u, (uint32_t)pool_recs, tri_base[mi], kLbvhWideLeafTris, built[mi])) return rc; // (its binary nodes are not kept)
                pool_recs += built[mi].n_recs;
                build_ms += built[mi].ms;
            }
        }
        pool_tris += m.nf;
        if (pool_nodes >= ((size_t)1 << 31) || pool_tris >= ((size_t)1 << 28)) return ctx->fail(RT_ERR_UNSUPPORTED, "scene has more than 2^28 triangles");
    }
    std::vector<float4> bvh(gpu_bvh ? 0 : pool_nodes * 2, make_float4(0, 0, 0, 0));
    std::vector<float4> tris(gpu_bvh ? 0 : pool_tris * 3);
    std::vector<rt_bvh_node> roots(sc->n_meshes);
    std::vector<float4> bvh4; // records of four float4 (device/lbvh.h: wide_record_pack)
    std::vector<uint32_t> root4(sc->n_meshes, 0);
    for (uint32_t mi = 0; mi < sc->n_meshes; ++mi) {
        const rt_mesh &m = sc->meshes[mi];
        for (size_t k = 0; k < (size_t)m.nf * 3; ++k)
            if (m.f[k] >= m.nv || m.fn[k] >= m.nvn || m.ft[k] >= m.nvt) return ctx->fail(RT_ERR_INVALID, "mesh %u: face index out of range", mi);
        // BVH: walk it once - checks indices, finds the depth, proves it is a tree
        const uint32_t nb = node_base[mi], tb = tri_base[mi];
        uint32_t depth = 0;
        if (gpu_bvh) {
            const GpuBvh &g = built[mi];
            depth = g.depth;
            max_wide_levels = std::max(max_wide_levels, (int)g.wide_levels);
            memcpy(roots[mi].bounds, g.bounds, sizeof g.bounds);
            roots[mi].data = g.root_word;
            root4[mi] = g.root4;
        } else {
        size_t visited = 0;
        std::vector<std::pair<uint32_t, uint32_t>> stack{{1u, 0u}};
        while (!stack.empty()) {
            auto [id, lvl] = stack.back();
            stack.pop_back();
            if (id == 0 || id >= m.n_bvh_nodes || ++visited > m.n_bvh_nodes) return ctx->fail(RT_ERR_INVALID, "mesh %u: BVH node index out of range or cyclic", mi);
            depth = std::max(depth, lvl);
            const rt_bvh_node &b = m.bvh_nodes[id];
            uint32_t data;
            if (b.data & 0x80000000u) {
                const uint32_t off = b.data & 0x0FFFFFFFu, cnt = ((b.data >> 28) & 7u) + 1;
                if ((size_t)off + cnt > m.nf) return ctx->fail(RT_ERR_INVALID, "mesh %u: BVH leaf out of range", mi);
                data = (b.data & 0xF0000000u) | (tb + off);
                for (uint32_t k = off; k < off + cnt; ++k) {
                    const uint32_t face = m.bvh_elements[k];
                    if (face >= m.nf) return ctx->fail(RT_ERR_INVALID, "mesh %u: BVH element out of range", mi);
                    const float *v0 = &m.v[m.f[face * 3] * 3], *v1 = &m.v[m.f[face * 3 + 1] * 3], *v2 = &m.v[m.f[face * 3 + 2] * 3];
                    tris[(size_t)(tb + k) * 3] = make_float4(v0[0], v0[1], v0[2], v1[0]);
                    tris[(size_t)(tb + k) * 3 + 1] = make_float4(v1[1], v1[2], v2[0], v2[1]);
                    tris[(size_t)(tb + k) * 3 + 2] = make_float4(v2[2], __builtin_bit_cast(float, face), 0.0f, 0.0f);
                }
            } else {
                const uint32_t c = b.data & 0x7FFFFFFFu;
                if (c & 1u) return ctx->fail(RT_ERR_UNSUPPORTED, "mesh %u: BVH child pairs must start at even indices", mi);
                data = nb + c;
                stack.push_back({c, lvl + 1});
                stack.push_back({c + 1, lvl + 1});
            }
            // bounds = min.xyz, max.xyz; stored interleaved per axis (see bvh_box_fast)
            bvh[(size_t)(nb + id) * 2] = make_float4(b.bounds[0], b.bounds[3], b.bounds[1], b.bounds[4]);
            bvh[(size_t)(nb + id) * 2 + 1] = make_float4(b.bounds[2], b.bounds[5], __builtin_bit_cast(float, data), 0.0f);
            if (id == 1) { roots[mi] = b; roots[mi].data = data; }
        }
        // the four-wide form: a record per inner node reached in steps of two levels from the root
        if (wide_gpu) { // (the frames' rays have a tree of their own, built above)
            root4[mi] = built[mi].root4;
            max_wide_levels = std::max(max_wide_levels, (int)built[mi].wide_levels);
        } else {
            max_wide_levels = std::max(max_wide_levels, ((int)depth + 1) / 2); // a record per two levels
            const auto pooled = [&](uint32_t id) { return __builtin_bit_cast(uint32_t, bvh[(size_t)(nb + id) * 2 + 1].z); }; // data word, pool numbering
            const auto is_leaf = [&](uint32_t id) { return (m.bvh_nodes[id].data & 0x80000000u) != 0; };
            std::vector<std::pair<uint32_t, size_t>> todo; // (binary node, its record)
            const auto record_of = [&](uint32_t id) -> uint32_t { // data word of binary node `id` in the four-wide form
                if (is_leaf(id)) return pooled(id);
                const size_t rec = bvh4.size() / 4;
                bvh4.resize(bvh4.size() + 4);
                todo.push_back({id, rec});
                return (uint32_t)rec;
            };
            root4[mi] = record_of(1);
            while (!todo.empty()) { // depth-first: a record's children follow it closely
                const auto [id, rec] = todo.back();
                todo.pop_back();
                const uint32_t c = m.bvh_nodes[id].data & 0x7FFFFFFFu;
                uint32_t slots[4], ns = 0;
                for (uint32_t ch = c; ch < c + 2; ++ch) {
                    if (is_leaf(ch)) slots[ns++] = ch;
                    else { const uint32_t g = m.bvh_nodes[ch].data & 0x7FFFFFFFu; slots[ns++] = g; slots[ns++] = g + 1; }
                }
                float boxes[4][6];
                uint32_t words[4] = {kWideNoChild, kWideNoChild, kWideNoChild, kWideNoChild};
                for (uint32_t k = 0; k < ns; ++k) {
                    words[k] = record_of(slots[k]); // may grow bvh4: index, not pointer
                    memcpy(boxes[k], m.bvh_nodes[slots[k]].bounds, sizeof boxes[k]);
                }
                float4 packed[4];
                wide_record_pack(boxes, words, ns, packed);
                for (int i = 0; i < 4; ++i) bvh4[rec * 4 + i] = packed[i];
            }
            if (bvh4.size() / 4 >= ((size_t)1 << 31)) return ctx->fail(RT_ERR_UNSUPPORTED, "scene has too many BVH nodes");
        }
        } // host-built BVH
        max_mesh_depth = std::max(max_mesh_depth, (int)depth);
        DMesh &d = dmeshes[mi];
        d.n_nodes = gpu_bvh ? built[mi].n_nodes : m.n_bvh_nodes; d.depth = depth; d.root = nb + 1;
        HIP_TRY(ctx, upload(m.fn, (size_t)m.nf * 12, (const void **)&d.fn));
        HIP_TRY(ctx, upload(m.ft, (size_t)m.nf * 12, (const void **)&d.ft));
        HIP_TRY(ctx, upload(m.vn, (size_t)m.nvn * 12, (const void **)&d.vn));
        HIP_TRY(ctx, upload(m.vt, (size_t)m.nvt * 12, (const void **)&d.vt));
        d.fmtl = nullptr;
        if (sc->face_materials && m.face_material) {
            for (uint32_t k = 0; k < m.nf; ++k)
                if (m.face_material[k] >= m.n_sub_materials) return ctx->fail(RT_ERR_INVALID, "mesh %u: face %u names sub-material %u of %u", mi, k, m.face_material[k], m.n_sub_materials);
            HIP_TRY(ctx, upload(m.face_material, (size_t)m.nf * 4, (const void **)&d.fmtl));
            any_face_materials = true;
        }
        auto rec = std::make_unique<DeviceBuf>();
        HIP_TRY(ctx, rec->reserve(std::max<size_t>(1, (size_t)m.nf * 64)));
        if (m.nf) hipLaunchKernelGGL(k_face_records, dim3((unsigned)std::min<size_t>(((size_t)m.nf + kBlock - 1) / kBlock, (size_t)ctx->n_cu * 16)), dim3(kBlock), 0,
                                     ctx->stream, d.fn, d.vn, d.fmtl, m.nf, rec->as<float4>());
        HIP_TRY(ctx, hipGetLastError());
        d.face_rec = rec->as<float4>();
        bufs.push_back(std::move(rec));
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const float4 *d_bvh_pool = nullptr, *d_tri_pool = nullptr, *d_bvh4_pool = nullptr, *d_tri4_pool = nullptr;
    if (sc->n_meshes && gpu_bvh) { // the meshes' parts, each built in pool numbering, side by side in the three pools
        auto pool = [&](size_t bytes, const void **dst) -> hipError_t {
            auto b = std::make_unique<DeviceBuf>();
            hipError_t e = b->reserve(std::max<size_t>(128, bytes));
            *dst = b->p;
            bufs.push_back(std::move(b));
            return e;
        };
        HIP_TRY(ctx, pool(pool_nodes * 32, (const void **)&d_bvh_pool));
        HIP_TRY(ctx, pool(pool_recs * 64, (const void **)&d_bvh4_pool));
        HIP_TRY(ctx, pool(pool_tris * 48, (const void **)&d_tri_pool));
        HIP_TRY(ctx, hipMemset((void *)d_bvh_pool, 0, 64));
        size_t rec_at = 0;
        for (uint32_t mi = 0; mi < sc->n_meshes; ++mi) {
            GpuBvh &g = built[mi];
            HIP_TRY(ctx, hipMemcpy((char *)d_bvh_pool + (size_t)node_base[mi] * 32, g.nodes->p, (size_t)g.n_nodes * 32, hipMemcpyDeviceToDevice));
            if (g.n_recs) HIP_TRY(ctx, hipMemcpy((char *)d_bvh4_pool + rec_at * 64, g.recs->p, (size_t)g.n_recs * 64, hipMemcpyDeviceToDevice));
            HIP_TRY(ctx, hipMemcpy((char *)d_tri_pool + (size_t)tri_base[mi] * 48, g.tris->p, (size_t)sc->meshes[mi].nf * 48, hipMemcpyDeviceToDevice));
            rec_at += g.n_recs;
            g.nodes.reset(); g.recs.reset(); g.tris.reset();
        }
        for (DMesh &d : dmeshes) { d.bvh = d_bvh_pool; d.tris = d_tri_pool; }
        d_tri4_pool = d_tri_pool;
        ctx->bvh_build_ms = build_ms;
    } else if (sc->n_meshes) {
        HIP_TRY(ctx, upload(bvh.data(), bvh.size() * sizeof(float4), (const void **)&d_bvh_pool));
        HIP_TRY(ctx, upload(tris.data(), tris.size() * sizeof(float4), (const void **)&d_tri_pool));
        for (DMesh &d : dmeshes) { d.bvh = d_bvh_pool; d.tris = d_tri_pool; }
        if (wide_gpu) { // the four-wide records and the triangle records in their leaf order, mesh after mesh
            auto pool = [&](size_t bytes, const void **dst) -> hipError_t {
                auto b = std::make_unique<DeviceBuf>();
                hipError_t e = b->reserve(std::max<size_t>(128, bytes));
                *dst = b->p;
                bufs.push_back(std::move(b));
                return e;
            };
            HIP_TRY(ctx, pool(pool_recs * 64, (const void **)&d_bvh4_pool));
            HIP_TRY(ctx, pool(pool_tris * 48, (const void **)&d_tri4_pool));
            size_t rec_at = 0;
            for (uint32_t mi = 0; mi < sc->n_meshes; ++mi) {
                GpuBvh &g = built[mi];
                if (g.n_recs) HIP_TRY(ctx, hipMemcpy((char *)d_bvh4_pool + rec_at * 64, g.recs->p, (size_t)g.n_recs * 64, hipMemcpyDeviceToDevice));
                HIP_TRY(ctx, hipMemcpy((char *)d_tri4_pool + (size_t)tri_base[mi] * 48, g.tris->p, (size_t)sc->meshes[mi].nf * 48, hipMemcpyDeviceToDevice));
                rec_at += g.n_recs;
                g.nodes.reset(); g.recs.reset(); g.tris.reset();
            }
            ctx->bvh_build_ms = build_ms;
        } else {
            if (bvh4.empty()) bvh4.resize(4, make_float4(0, 0, 0, 0));
            HIP_TRY(ctx, upload(bvh4.data(), bvh4.size() * sizeof(float4), (const void **)&d_bvh4_pool));
            d_tri4_pool = d_tri_pool;
            ctx->bvh_build_ms = 0;
        }
    }
    // compact node records of the ray-list kernel
    std::vector<TNode> tnodes(nn);
    for (uint32_t i = 0; i < nn; ++i) {
        TNode &t = tnodes[i];
        memset(&t, 0, sizeof t);
        memcpy(t.itm, nodes[i].itm, sizeof t.itm);
        t.meta = (uint32_t)nodes[i].prim | ((uint32_t)nodes[i].depth << 8);
        if (nodes[i].prim == RT_PRIM_MESH) {
            const rt_bvh_node &r = roots[nodes[i].mesh];
            t.root = r.data;
            t.root4 = root4[nodes[i].mesh];
            t.box[0] = r.bounds[0]; t.box[1] = r.bounds[3]; t.box[2] = r.bounds[1]; t.box[3] = r.bounds[4]; t.box[4] = r.bounds[2]; t.box[5] = r.bounds[5];
        }
    }
    // ---- world-space boxes of the object nodes: rays that miss a node's box need not be taken into its space.
    // The reference tests every ray against every object (src/main.cpp:16-89); skipping one whose box the ray misses
    // changes no answer as long as the box is conservative, hence the padding (node matrices and their inverses are
    // single precision and agree to a few ulp only).
    for (uint32_t i = 0; i < nn; ++i) {
        DNode &d = nodes[i];
        for (int a = 0; a < 3; ++a) { d.wbox[a] = -FLT_MAX; d.wbox[3 + a] = FLT_MAX; }
        double lo[3] = {-1, -1, -1}, hi[3] = {1, 1, 1}; // unit sphere; unit square in z = 0
        if (d.prim == RT_PRIM_PLANE) lo[2] = hi[2] = 0;
        else if (d.prim == RT_PRIM_MESH) {
            const rt_bvh_node &root = roots[d.mesh];
            for (int a = 0; a < 3; ++a) { lo[a] = root.bounds[a]; hi[a] = root.bounds[3 + a]; }
        } else if (d.prim != RT_PRIM_SPHERE) continue;
        double wlo[3] = {1e300, 1e300, 1e300}, whi[3] = {-1e300, -1e300, -1e300};
        bool finite = true;
        for (int c = 0; c < 8; ++c) {
            double q[3] = {(c & 1) ? hi[0] : lo[0], (c & 2) ? hi[1] : lo[1], (c & 4) ? hi[2] : lo[2]};
            for (int k = d.depth; k >= 0; --k) { // node space -> world, through the ancestors' transformations
                const float *m = nodes[d.chain[k]].tm;
                const double x = q[0] * m[0] + q[1] * m[3] + q[2] * m[6] + m[9];
                const double y = q[0] * m[1] + q[1] * m[4] + q[2] * m[7] + m[10];
                const double z = q[0] * m[2] + q[1] * m[5] + q[2] * m[8] + m[11];
                q[0] = x; q[1] = y; q[2] = z;
            }
            for (int a = 0; a < 3; ++a) { finite = finite && std::isfinite(q[a]); wlo[a] = std::min(wlo[a], q[a]); whi[a] = std::max(whi[a], q[a]); }
        }
        if (!finite) continue;
        double extent = 0;
        for (int a = 0; a < 3; ++a) extent = std::max(extent, std::max(std::fabs(wlo[a]), std::fabs(whi[a])));
        const double padding = 1e-3 * extent + 1e-4;
        for (int a = 0; a < 3; ++a) { d.wbox[a] = (float)(wlo[a] - padding); d.wbox[3 + a] = (float)(whi[a] + padding); }
    }
    // ---- hierarchy over the object nodes' world boxes (TlasNode): median splits of the box centres, one object per leaf ----
    std::vector<TlasNode> tlas;
    {
        std::vector<int> boxed, unboxed;
        for (uint32_t i = 0; i < nn; ++i) {
            if (nodes[i].prim == RT_PRIM_NONE) continue;
            (nodes[i].wbox[0] == -FLT_MAX ? unboxed : boxed).push_back((int)i);
        }
        if (ctx->tlas_min_objects > 0 && (int)(boxed.size() + unboxed.size()) >= ctx->tlas_min_objects) {
            for (int i : unboxed) { // objects without a finite box: leaves every ray reaches
                TlasNode t{};
                for (int a = 0; a < 3; ++a) { t.box[a] = -FLT_MAX; t.box[3 + a] = FLT_MAX; }
                t.object = i; t.skip = (int32_t)tlas.size() + 1;
                tlas.push_back(t);
            }
            std::function<void(int, int)> build = [&](int lo, int hi) {
                const size_t me = tlas.size();
                tlas.push_back(TlasNode{});
                float b[6] = {FLT_MAX, FLT_MAX, FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX}, c[6] = {FLT_MAX, FLT_MAX, FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
                for (int k = lo; k < hi; ++k) {
                    const float *w = nodes[boxed[k]].wbox;
                    for (int a = 0; a < 3; ++a) {
                        b[a] = std::min(b[a], w[a]); b[3 + a] = std::max(b[3 + a], w[3 + a]);
                        const float mid = 0.5f * w[a] + 0.5f * w[3 + a];
                        c[a] = std::min(c[a], mid); c[3 + a] = std::max(c[3 + a], mid);
                    }
                }
                memcpy(tlas[me].box, b, sizeof b);
                if (hi - lo == 1) tlas[me].object = boxed[lo];
                else {
                    tlas[me].object = -1;
                    int axis = 0;
                    for (int a = 1; a < 3; ++a) if (c[3 + a] - c[a] > c[3 + axis] - c[axis]) axis = a;
                    const int mid = (lo + hi) / 2;
                    std::nth_element(boxed.begin() + lo, boxed.begin() + mid, boxed.begin() + hi, [&](int x, int y) {
                        const float cx = nodes[x].wbox[axis] + nodes[x].wbox[3 + axis], cy = nodes[y].wbox[axis] + nodes[y].wbox[3 + axis];
                        return cx < cy || (cx == cy && x < y);
                    });
                    build(lo, mid);
                    build(mid, hi);
                }
                tlas[me].skip = (int32_t)tlas.size();
            };
            if (!boxed.empty()) build(0, (int)boxed.size());
        }
    }
    // ---- textures ----
    const auto tex_ok = [&](int32_t t) { return t >= 0 && (uint32_t)t <= sc->n_textures; };
    if (!tex_ok(sc->background_texture) || !tex_ok(sc->environment_texture)) return ctx->fail(RT_ERR_INVALID, "background / environment texture index out of range");
    for (uint32_t i = 0; i < sc->n_materials; ++i)
        if (!tex_ok(sc->materials[i].diffuse_texture)) return ctx->fail(RT_ERR_INVALID, "material %u: diffuse texture index out of range", i);
    if (sc->n_textures > 0 && !sc->textures) return ctx->fail(RT_ERR_INVALID, "null texture array");
    std::vector<DTexture> dtex(sc->n_textures);
    for (uint32_t i = 0; i < sc->n_textures; ++i) {
        const rt_texture &t = sc->textures[i];
        DTexture &d = dtex[i];
        memset(&d, 0, sizeof d);
        d.type = t.type;
        memcpy(d.itm, t.itm, sizeof d.itm); memcpy(d.color1, t.color1, 12); memcpy(d.color2, t.color2, 12);
        if (t.type == RT_TEX_IMAGE) {
            if (t.width < 0 || t.height < 0 || (int64_t)t.width * t.height > ((int64_t)1 << 28) || (t.width > 0 && t.height > 0 && !t.pixels) ||
                ((t.width == 0) != (t.height == 0)))
                return ctx->fail(RT_ERR_INVALID, "texture %u: bad image size or null pixels", i);
            d.width = t.width; d.height = t.height;
            if (t.width > 0) HIP_TRY(ctx, upload(t.pixels, (size_t)t.width * t.height * 3, (const void **)&d.pixels));
        } else if (t.type != RT_TEX_CHECKER) return ctx->fail(RT_ERR_INVALID, "texture %u: unknown type %d", i, t.type);
    }
    DScene S{};
    // entries a lane's stack may hold: the binary walk pushes at most two nodes per level of the tree; the four-wide walk
    // of the ray-list kernel pushes up to three nodes per level of the wide tree
    S.stack_levels = max_mesh_depth < 0 ? 0 : std::max(max_mesh_depth + 2, 3 * max_wide_levels + 2);
    // LDS holds the first levels of every lane's stack (26 KiB per 256 threads lets six
    // workgroups share a CU's 160 KiB); deeper entries spill to a global array sized
    // for the largest grid any entry point launches
    S.lds_levels = std::min(S.stack_levels, ctx->lds_levels_max);
    const size_t ovf_levels = (size_t)(S.stack_levels - S.lds_levels);
    ctx->ovf_threads = (size_t)ctx->n_cu * 8 * kBlock;
    HIP_TRY(ctx, ctx->d_stack_ovf.reserve(std::max<size_t>(1, ovf_levels) * ctx->ovf_threads * sizeof(uint2)));
    S.stack_overflow = ctx->d_stack_ovf.as<uint2>();

    HIP_TRY(ctx, ctx->d_nodes.reserve(sizeof(DNode) * nn));
    HIP_TRY(ctx, hipMemcpy(ctx->d_nodes.p, nodes.data(), sizeof(DNode) * nn, hipMemcpyHostToDevice));
    for (uint32_t i = 0; i < nn; ++i) {
        tnodes[i].meta |= (uint32_t)(nodes[i].has_children ? 1u : 0u) << 16;
        memcpy(tnodes[i].wbox, nodes[i].wbox, sizeof tnodes[i].wbox);
    }
    HIP_TRY(ctx, ctx->d_tnodes.reserve(sizeof(TNode) * nn));
    HIP_TRY(ctx, hipMemcpy(ctx->d_tnodes.p, tnodes.data(), sizeof(TNode) * nn, hipMemcpyHostToDevice));
    S.tnodes = ctx->d_tnodes.as<TNode>();
    HIP_TRY(ctx, ctx->d_tlas.reserve(sizeof(TlasNode) * std::max<size_t>(1, tlas.size())));
    if (!tlas.empty()) HIP_TRY(ctx, hipMemcpy(ctx->d_tlas.p, tlas.data(), sizeof(TlasNode) * tlas.size(), hipMemcpyHostToDevice));
    S.tlas = tlas.empty() ? nullptr : ctx->d_tlas.as<TlasNode>();
    S.n_tlas = (int32_t)tlas.size();
    S.bvh_pool = d_bvh_pool; S.tri_pool = d_tri_pool; S.bvh4_pool = d_bvh4_pool; S.tri4_pool = d_tri4_pool;
    S.cancel = (const uint32_t *)&ctx->host_words[0];
    S.progress = &ctx->host_words[1];
    HIP_TRY(ctx, ctx->d_meshes.reserve(sizeof(DMesh) * std::max<size_t>(1, dmeshes.size())));
    if (!dmeshes.empty()) HIP_TRY(ctx, hipMemcpy(ctx->d_meshes.p, dmeshes.data(), sizeof(DMesh) * dmeshes.size(), hipMemcpyHostToDevice));
    HIP_TRY(ctx, ctx->d_materials.reserve(sizeof(rt_material) * std::max<uint32_t>(1, sc->n_materials)));
    if (sc->n_materials) HIP_TRY(ctx, hipMemcpy(ctx->d_materials.p, sc->materials, sizeof(rt_material) * sc->n_materials, hipMemcpyHostToDevice));
    HIP_TRY(ctx, ctx->d_lights.reserve(sizeof(rt_light) * sc->n_lights));
    HIP_TRY(ctx, hipMemcpy(ctx->d_lights.p, sc->lights, sizeof(rt_light) * sc->n_lights, hipMemcpyHostToDevice));

    HIP_TRY(ctx, ctx->d_textures.reserve(sizeof(DTexture) * std::max<size_t>(1, dtex.size())));
    if (!dtex.empty()) HIP_TRY(ctx, hipMemcpy(ctx->d_textures.p, dtex.data(), sizeof(DTexture) * dtex.size(), hipMemcpyHostToDevice));
    S.textures = ctx->d_textures.as<DTexture>();
    S.background_texture = sc->background_texture; S.environment_texture = sc->environment_texture;
    S.nodes = ctx->d_nodes.as<DNode>();
    S.meshes = ctx->d_meshes.as<DMesh>();
    S.materials = ctx->d_materials.as<rt_material>();
    S.lights = ctx->d_lights.as<rt_light>();
    S.n_nodes = (int)nn; S.n_lights = (int)sc->n_lights;
    S.face_materials = any_face_materials ? 1 : 0;
    memcpy(S.background, sc->background, 12);
    memcpy(S.environment, sc->environment, 12);
    ctx->mesh_bufs.swap(bufs);
    ctx->pmap_n[0] = ctx->pmap_n[1] = 0; // photon maps belong to the scene they were traced in
    ctx->pmap_flags = -1;
    ctx->scene = S;
    ++ctx->scene_generation;
    ctx->ray_levels = ray_levels;
    ctx->have_meshes = max_mesh_depth >= 0;
    ctx->have_scene = true;
    return RT_OK;
}

int rt_camera_set(rt_ctx *ctx, const rt_camera *cam)
{
    if (!ctx) return RT_ERR_INVALID;
    if (!cam) return ctx->fail(RT_ERR_INVALID, "null camera");
    if (cam->width < 1 || cam->height < 1 || (int64_t)cam->width * cam->height > (int64_t)1 << 28)
        return ctx->fail(RT_ERR_INVALID, "image size %dx%d out of range", cam->width, cam->height);
    ctx->camera = *cam;
    ctx->have_camera = true;
    return RT_OK;
}

int rt_get_counters(rt_ctx *ctx, rt_counters *out)
{
    if (!ctx || !out) return RT_ERR_INVALID;
    *out = ctx->counters;
    out->bvh_build_ms = ctx->bvh_build_ms;
    return RT_OK;
}

int rt_render(rt_ctx *ctx, const rt_render_params *prm, rt_frame_out *out)
{
    if (!ctx) return RT_ERR_INVALID;
    if (!ctx->have_scene || !ctx->have_camera) return ctx->fail(RT_ERR_STATE, "rt_render needs rt_scene_upload and rt_camera_set first");
    if (!out) return ctx->fail(RT_ERR_INVALID, "null frame output");
    if (int rc = validate_params(ctx, prm)) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const int W = ctx->camera.width, H = ctx->camera.height;
    const size_t npx = (size_t)W * H;

    DFrame F;
    fill_frame(ctx, prm, F);

    // halton tables (src/main.cpp:109-112): bases 2,3,5,7, `period` entries each
    float hal[256] = {0};
    const int bases[4] = {2, 3, 5, 7};
    for (int b = 0; b < 4; ++b) for (int i = 0; i < prm->halton_period; ++i) hal[b * 64 + i] = halton(i, bases[b]);
    HIP_TRY(ctx, ctx->d_halton.reserve(sizeof hal));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_halton.p, hal, sizeof hal, hipMemcpyHostToDevice, s));

    // pixel list of this shard in tile order (src/main.cpp:278-288), cached per image/tile geometry
    const int key[6] = {W, H, prm->tile_size, prm->tile_rank, prm->tile_world, 0};
    if (memcmp(key, ctx->active_key, sizeof key) != 0) {
        const int ts = prm->tile_size;
        const int tx = (W + ts - 1) / ts, ty = (H + ts - 1) / ts;
        ctx->active_host.clear();
        for (int t = prm->tile_rank; t < tx * ty; t += prm->tile_world) {
            const int x0 = (t % tx) * ts, y0 = (t / tx) * ts;
            const int tw = std::min(ts, W - x0), th = std::min(ts, H - y0);
            for (int j = y0; j < y0 + th; ++j) for (int i = x0; i < x0 + tw; ++i) ctx->active_host.push_back(j * W + i);
        }
        // the shard's whole list stays on the device (8 MB at 1080p: a millisecond of upload per frame otherwise); the
        // first sampling round reads it, the later rounds' shrinking lists alternate between the two d_active buffers
        HIP_TRY(ctx, ctx->d_active_all.reserve(sizeof(int32_t) * std::max<size_t>(1, ctx->active_host.size())));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_active_all.p, ctx->active_host.data(), sizeof(int32_t) * ctx->active_host.size(), hipMemcpyHostToDevice, s));
        HIP_TRY(ctx, hipStreamSynchronize(s));
        memcpy(ctx->active_key, key, sizeof key);
    }
    const uint32_t n0 = (uint32_t)ctx->active_host.size();
    for (auto &b : ctx->d_active) HIP_TRY(ctx, b.reserve(sizeof(int32_t) * std::max<uint32_t>(1, n0)));
    HIP_TRY(ctx, ctx->d_state.reserve(sizeof(PixelState) * npx));

    // output planes: the caller's device memory, or internal planes copied out afterwards
    FramePlanes P{};
    const bool dev_out = out->device_memory != 0;
    if (dev_out) {
        P.rgb = out->rgb; P.z = out->z; P.sample_count = out->sample_count; P.object_id = out->object_id; P.linear = out->linear;
    } else {
        if (out->rgb) { HIP_TRY(ctx, ctx->d_rgb.reserve(npx * 3)); P.rgb = ctx->d_rgb.as<uint8_t>(); }
        if (out->z) { HIP_TRY(ctx, ctx->d_z.reserve(npx * 4)); P.z = ctx->d_z.as<float>(); }
        if (out->sample_count) { HIP_TRY(ctx, ctx->d_cnt.reserve(npx * 4)); P.sample_count = ctx->d_cnt.as<int32_t>(); }
        if (out->object_id) { HIP_TRY(ctx, ctx->d_id.reserve(npx * 4)); P.object_id = ctx->d_id.as<int32_t>(); }
        if (out->linear) { HIP_TRY(ctx, ctx->d_lin.reserve(npx * 12)); P.linear = ctx->d_lin.as<float>(); }
    }

    // the caller's host planes, from the internal ones (also on a cancelled frame: the pixels finished so far are in them)
    const auto planes_to_host = [&]() -> int {
        if (dev_out) return RT_OK;
        // whole image: straight copies; a tile shard: stage and copy only the rows of its tiles
        const bool whole = prm->tile_world == 1;
        struct Plane { void *dst; const DeviceBuf *src; size_t px_bytes; };
        const Plane planes[5] = {{out->rgb, &ctx->d_rgb, 3}, {out->z, &ctx->d_z, 4}, {out->sample_count, &ctx->d_cnt, 4},
                                 {out->object_id, &ctx->d_id, 4}, {out->linear, &ctx->d_lin, 12}};
        std::vector<uint8_t> stage;
        for (const Plane &pl : planes) {
            if (!pl.dst) continue;
            if (whole) { HIP_TRY(ctx, hipMemcpyAsync(pl.dst, pl.src->p, npx * pl.px_bytes, hipMemcpyDeviceToHost, s)); continue; }
            stage.resize(npx * pl.px_bytes);
            HIP_TRY(ctx, hipMemcpyAsync(stage.data(), pl.src->p, npx * pl.px_bytes, hipMemcpyDeviceToHost, s));
            HIP_TRY(ctx, hipStreamSynchronize(s));
            for (int32_t pix : ctx->active_host)
                memcpy((uint8_t *)pl.dst + (size_t)pix * pl.px_bytes, stage.data() + (size_t)pix * pl.px_bytes, pl.px_bytes);
        }
        return hipStreamSynchronize(s) == hipSuccess ? RT_OK : ctx->fail(RT_ERR_HIP, "copy of the frame to the host failed");
    };

    HIP_TRY(ctx, hipMemsetAsync(ctx->d_counters.p, 0, sizeof(DCounters), s));
#ifdef RT_PROFILE_TAIL
    HIP_TRY(ctx, hipMemsetAsync((char *)ctx->d_counters.p + offsetof(DCounters, phase) + 12 * 8, 0xFF, 8, s)); // the minimum starts at the top
    HIP_TRY(ctx, hipMemsetAsync((char *)ctx->d_counters.p + offsetof(DCounters, phase) + 0 * 8, 0xFF, 8, s));
    HIP_TRY(ctx, hipMemsetAsync((char *)ctx->d_counters.p + offsetof(DCounters, phase) + 7 * 8, 0xFF, 8, s));
#endif
    const bool gathers = prm->mode == RT_MODE_WHITTED && prm->photon_flags != 0;
    const size_t lds = gathers ? gather_lds_bytes(ctx->scene) : stack_lds_bytes(ctx->scene);
    double trace_ms = 0.0;
    uint32_t launches = 0;
    uint64_t wf_paths = 0; // camera paths of the wavefront pipeline (its kernels count rays only)
    if (int rc = begin_cancellable(ctx)) return rc;
    ctx->prog_pixels_total.store(n0); ctx->prog_pixels_done.store(0); ctx->prog_round.store(0);
    ctx->prog_round_items.store(0); ctx->prog_round_done.store(0); ctx->prog_live_queue.store(0);
    ctx->prog_rendering.store(1);
    struct Done { rt_ctx *c; ~Done() { c->prog_rendering.store(0); c->prog_live_queue.store(0); } } done_guard{ctx};
    HIP_TRY(ctx, hipEventRecord(ctx->ev0, s));

    // Expensive pixels first (path frames): keys from one camera ray per pixel, then a stable descending radix sort.
    // A launch ends with a tail of a few milliseconds in which the last long paths run alone; it is shorter when the
    // long paths started first.  On the whole 1080p frame the tail is 3 % and the keys cost as much as they save
    // (0.6 ms); on the 1/8 shard of an 8-GPU run the launch is 1.0 ms shorter (22.4 -> 21.4 ms), so it is on for shards.
    const int32_t *first_list = ctx->d_active_all.as<int32_t>();
    const bool order = ctx->order_pixels < 0 ? prm->tile_world > 1 : ctx->order_pixels != 0;
    const bool wavefront_frame = prm->mode == RT_MODE_PATH && prm->max_bounces > 0 && (ctx->pipeline == 1 || (ctx->pipeline < 0 && ctx->have_meshes));
    if (order && prm->mode == RT_MODE_PATH && prm->max_bounces > 0 && n0 > 1 && !wavefront_frame) { // (the wavefront pipeline has all paths of a round in flight at once: nothing to order)
        HIP_TRY(ctx, ctx->d_active_sorted.reserve(sizeof(int32_t) * n0));
        for (auto &b : ctx->d_keys) HIP_TRY(ctx, b.reserve(sizeof(uint32_t) * n0));
        const void *fn = ctx->ray_levels <= 1 ? (const void *)k_order_keys<1> : ctx->ray_levels == 2 ? (const void *)k_order_keys<2> : (const void *)k_order_keys<5>;
        const dim3 grid((unsigned)std::min<uint64_t>(((uint64_t)n0 + kBlock - 1) / kBlock, ctx->ovf_threads / kBlock));
        DScene a_scene = ctx->scene;
        const float *a_hal = ctx->d_halton.as<float>();
        const int32_t *a_act = first_list;
        uint32_t a_n = n0;
        uint32_t *a_keys = ctx->d_keys[0].as<uint32_t>();
        void *args[] = {&a_scene, &F, &a_hal, &a_act, &a_n, &a_keys};
        HIP_TRY(ctx, hipLaunchKernel(fn, grid, dim3(kBlock), args, stack_lds_bytes(ctx->scene), s));
        size_t tmp_bytes = 0;
        HIP_TRY(ctx, rocprim::radix_sort_pairs_desc(nullptr, tmp_bytes, ctx->d_keys[0].as<uint32_t>(), ctx->d_keys[1].as<uint32_t>(),
                                                    first_list, ctx->d_active_sorted.as<int32_t>(), (size_t)n0, 0u, 24u, s));
        HIP_TRY(ctx, ctx->d_sort_tmp.reserve(std::max<size_t>(1, tmp_bytes)));
        HIP_TRY(ctx, rocprim::radix_sort_pairs_desc(ctx->d_sort_tmp.p, tmp_bytes, ctx->d_keys[0].as<uint32_t>(), ctx->d_keys[1].as<uint32_t>(),
                                                    first_list, ctx->d_active_sorted.as<int32_t>(), (size_t)n0, 0u, 24u, s));
        first_list = ctx->d_active_sorted.as<int32_t>();
    }

    // sampling rounds: the first takes min_spp samples of every pixel, later ones
    // up to 16 more of the pixels the stop test has not released yet
    // path frames of scenes with meshes go through the wavefront pipeline (ray-level regeneration keeps the waves of the
    // BVH walk full); scenes of spheres and planes, where every ray costs the same, through the lock-step path kernel
    const bool has_mesh = ctx->have_meshes;
    const bool wavefront = prm->mode == RT_MODE_PATH && prm->max_bounces > 0 && (ctx->pipeline == 1 || (ctx->pipeline < 0 && has_mesh));
    const uint64_t kMaxItems = wavefront ? 1ull << 27 : 1ull << 28; // bounds the sample buffer to 4 GiB (the wavefront's path state to 24 GiB)
    uint32_t n_active = n0;
    int cur = 0, k0 = 0;
    while (n_active > 0 && k0 < prm->max_spp) {
        const int nk = k0 == 0 ? prm->min_spp : std::min(16, prm->max_spp - k0);
        ctx->prog_round.fetch_add(1);
        const uint32_t chunk_px = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(n_active, kMaxItems / (uint64_t)nk));
        HIP_TRY(ctx, ctx->d_sbuf.reserve((size_t)chunk_px * nk * sizeof(float4)));
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_next_count.p, 0, sizeof(uint32_t), s));
        for (uint32_t first = 0; first < n_active; first += chunk_px) {
            const uint32_t cnt = std::min(chunk_px, n_active - first);
            const uint64_t items = (uint64_t)cnt * nk;
            const int32_t *act = (k0 == 0 ? first_list : ctx->d_active[cur].as<int32_t>()) + first;
            ctx->prog_round_done.store(0); ctx->prog_round_items.store(items);
            __atomic_store_n(&ctx->host_words[1], 0ull, __ATOMIC_RELEASE);
            ctx->prog_live_queue.store(wavefront || prm->max_bounces <= 0 ? 0 : 1);
            HIP_TRY(ctx, hipEventRecord(ctx->evk0, s));
            if (prm->mode == RT_MODE_WHITTED) {
                const void *fn = whitted_kernel(ctx->counting, ctx->ray_levels);
                for (int w = 0; w < 2; ++w) { ctx->scene.pmap[w] = ctx->pmap_n[w] ? ctx->pmap(w) : nullptr; ctx->scene.pmap_n[w] = ctx->pmap_n[w]; }
                int per_cu = 1;
                HIP_TRY(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, kBlock, lds));
                per_cu = std::max(1, std::min(per_cu, 8));
                const dim3 grid((unsigned)std::min<uint64_t>((items + kBlock - 1) / kBlock, (uint64_t)ctx->n_cu * per_cu));
                HIP_TRY(ctx, hipMemsetAsync(ctx->d_queue.p, 0, kQueueBytes, s));
                {   // items per draw from the work queue: about a sixteenth of what a wave will get (see the path kernel below)
                    const uint64_t per_wave = items / ((uint64_t)grid.x * (kBlock / 64)) / 16;
                    F.chunk = per_wave >= 512 ? 512u : per_wave >= 256 ? 256u : per_wave >= 128 ? 128u : 64u;
                    if (ctx->chunk_items > 0) F.chunk = (uint32_t)ctx->chunk_items;
                }
                DScene a_scene = ctx->scene;
                const float *a_hal = ctx->d_halton.as<float>();
                uint64_t a_items = items;
                int a_k0 = k0, a_nk = nk;
                float4 *a_sbuf = ctx->d_sbuf.as<float4>();
                DCounters *a_ctr = ctx->d_counters.as<DCounters>();
                unsigned long long *a_queue = ctx->d_queue.as<unsigned long long>();
                void *args[] = {&a_scene, &F, &a_hal, &act, &a_items, &a_k0, &a_nk, &a_sbuf, &P, &a_ctr, &a_queue};
                HIP_TRY(ctx, hipLaunchKernel(fn, grid, dim3(kBlock), args, lds, s));
            } else if (prm->max_bounces <= 0) {
                const void *fn = ctx->ray_levels <= 1 ? (const void *)k_probe<1> : ctx->ray_levels == 2 ? (const void *)k_probe<2> : (const void *)k_probe<5>;
                const dim3 grid((unsigned)std::min<uint64_t>((items + kBlock - 1) / kBlock, ctx->ovf_threads / kBlock));
                DScene a_scene = ctx->scene;
                const float *a_hal = ctx->d_halton.as<float>();
                uint64_t a_items = items;
                int a_k0 = k0, a_nk = nk;
                float4 *a_sbuf = ctx->d_sbuf.as<float4>();
                DCounters *a_ctr = ctx->d_counters.as<DCounters>();
                void *args[] = {&a_scene, &F, &a_hal, &act, &a_items, &a_k0, &a_nk, &a_sbuf, &P, &a_ctr};
                HIP_TRY(ctx, hipLaunchKernel(fn, grid, dim3(kBlock), args, lds, s));
            } else if (wavefront) {
                if (int rc = render_wavefront(ctx, F, act, items, k0, nk, P, lds, &trace_ms, &launches)) return rc;
                wf_paths += items;
            } else {
            // persistent launch: as many workgroups as the chip holds at once (more would only queue up)
            {
                // Items per draw from the work queue.  Large chunks keep a wave on neighbouring pixels (a wave is 1 %
                // busier per halving: 144.5 / 146.1 / 150.1 / 157.4 ms of work per wave for 512 / 256 / 128 / 64 on the
                // 1080p frame), small ones even out the end of the launch (a wave of a 1/8 shard draws five chunks of 512
                // in all, and waits 3.1 ms on average for the slowest wave; 1.1 ms with 256): about a sixteenth of what
                // a wave will get, as a power of two between 256 and 1024.
                const uint64_t per_wave = items / ((uint64_t)ctx->n_cu * 6 * (kBlock / 64)) / 16;
                F.chunk = per_wave >= 1024 ? 1024u : per_wave >= 512 ? 512u : 256u;
                if (ctx->chunk_items > 0) F.chunk = (uint32_t)ctx->chunk_items;
            }
            const void *fn = paths_kernel(ctx->counting, ctx->ray_levels, ctx->scene.background_texture > 0, ctx->scene.n_tlas > 0);
            int per_cu = 1;
            HIP_TRY(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, kBlock, lds));
            per_cu = std::max(1, std::min(per_cu, 8));
            const uint64_t want_blocks = (items + kBlock - 1) / kBlock;
            const dim3 grid((unsigned)std::min<uint64_t>(want_blocks, (uint64_t)ctx->n_cu * per_cu));
            HIP_TRY(ctx, hipMemsetAsync(ctx->d_queue.p, 0, kQueueBytes, s));
            {
                DScene a_scene = ctx->scene;
                const float *a_hal = ctx->d_halton.as<float>();
                uint64_t a_items = items;
                int a_k0 = k0, a_nk = nk;
                float4 *a_sbuf = ctx->d_sbuf.as<float4>();
                DCounters *a_ctr = ctx->d_counters.as<DCounters>();
                unsigned long long *a_queue = ctx->d_queue.as<unsigned long long>();
                void *args[] = {&a_scene, &F, &a_hal, &act, &a_items, &a_k0, &a_nk, &a_sbuf, &P, &a_ctr, &a_queue};
                HIP_TRY(ctx, hipLaunchKernel(fn, grid, dim3(kBlock), args, lds, s));
            }
            }
            HIP_TRY(ctx, hipGetLastError());
            HIP_TRY(ctx, hipEventRecord(ctx->evk1, s));
            hipLaunchKernelGGL(k_resolve, dim3((cnt + kBlock - 1) / kBlock), dim3(kBlock), 0, s, F, act, cnt, k0, nk,
                               ctx->d_sbuf.as<float4>(), ctx->d_state.as<PixelState>(), P,
                               ctx->d_active[cur ^ 1].as<int32_t>(), ctx->d_next_count.as<uint32_t>());
            HIP_TRY(ctx, hipGetLastError());
            if (!wavefront) { // the wavefront pipeline times its own ray launches
                HIP_TRY(ctx, hipEventSynchronize(ctx->evk1));
                float ms = 0;
                HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->evk0, ctx->evk1));
                trace_ms += ms;
                ++launches;
            }
        }
        uint32_t next = 0;
        HIP_TRY(ctx, hipMemcpyAsync(&next, ctx->d_next_count.p, sizeof next, hipMemcpyDeviceToHost, s));
        HIP_TRY(ctx, hipStreamSynchronize(s));
        ctx->prog_live_queue.store(0);
        if (cancelled(ctx)) {
            if (int rc = planes_to_host()) return rc;
            return ctx->fail(RT_ERR_CANCELLED, "cancelled by rt_cancel after %d samples per pixel", k0);
        }
        ctx->prog_pixels_done.store(n0 - next);
        n_active = next;
        cur ^= 1;
        k0 += nk;
    }
    HIP_TRY(ctx, hipEventRecord(ctx->ev1, s));

    DCounters dc{};
    HIP_TRY(ctx, hipMemcpyAsync(&dc, ctx->d_counters.p, sizeof dc, hipMemcpyDeviceToHost, s));
    if (int rc = planes_to_host()) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(s));
    float ms = 0;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    rt_counters &c = ctx->counters;
    c.closest_rays = dc.closest; c.shadow_rays = dc.shadow; c.camera_paths = dc.paths + wf_paths;
    c.box_tests = dc.box_tests; c.tri_tests = dc.tri_tests;
    c.gathers = dc.gathers; c.photon_visits = dc.photon_visits;
    for (int i = 0; i < 16; ++i) c.phase_cycles[i] = dc.phase[i];
    c.kernel_ms = ms; c.trace_kernel_ms = trace_ms; c.trace_launches = launches; c.trace_pipeline = wavefront ? 1u : 0u;
    return RT_OK;
}

// ---- frame exchange between GPUs ----
uint32_t rt_frame_records(int width, int height, int tile_size, int world)
{
    if (width < 1 || height < 1 || tile_size < 1 || world < 1) return 0;
    const uint64_t tiles = (uint64_t)((width + tile_size - 1) / tile_size) * ((height + tile_size - 1) / tile_size);
    return (uint32_t)((tiles + world - 1) / world * (uint64_t)tile_size * tile_size);
}

static int exchange_args(rt_ctx *ctx, const rt_frame_out *pl, const void *packed, int tile_size, int rank, int world, FramePlanes &P)
{
    if (!ctx->have_camera) return ctx->fail(RT_ERR_STATE, "rt_camera_set first: the frame exchange needs the image size");
    if (!pl || !packed || !pl->rgb || !pl->z || !pl->sample_count || !pl->object_id || !pl->device_memory)
        return ctx->fail(RT_ERR_INVALID, "the frame exchange needs device pointers for rgb, z, sample_count and object_id");
    if (tile_size < 1 || world < 1 || rank < -1 || rank >= world) return ctx->fail(RT_ERR_INVALID, "bad tile size / rank / world");
    P = FramePlanes{pl->rgb, pl->z, pl->sample_count, pl->object_id, nullptr};
    return RT_OK;
}

int rt_frame_pack(rt_ctx *ctx, const rt_frame_out *planes, int tile_size, int rank, int world, void *packed)
{
    if (!ctx) return RT_ERR_INVALID;
    FramePlanes P;
    if (int rc = exchange_args(ctx, planes, packed, tile_size, rank, world, P)) return rc;
    if (rank < 0) return ctx->fail(RT_ERR_INVALID, "bad rank");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int W = ctx->camera.width, H = ctx->camera.height;
    const uint32_t n = rt_frame_records(W, H, tile_size, world);
    hipLaunchKernelGGL(k_frame_pack, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, P, W, H, tile_size, rank, world, n, (uint4 *)packed);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_frame_unpack(rt_ctx *ctx, const void *packed_all, int tile_size, int skip_rank, int world, const rt_frame_out *planes)
{
    if (!ctx) return RT_ERR_INVALID;
    FramePlanes P;
    if (int rc = exchange_args(ctx, planes, packed_all, tile_size, skip_rank, world, P)) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int W = ctx->camera.width, H = ctx->camera.height;
    const uint32_t n = rt_frame_records(W, H, tile_size, world);
    hipLaunchKernelGGL(k_frame_unpack, dim3((n + kBlock - 1) / kBlock, (unsigned)world), dim3(kBlock), 0, ctx->stream, (const uint4 *)packed_all, W, H,
                       tile_size, skip_rank, world, n, P);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

// ---- batch queries ----
static dim3 batch_grid(const rt_ctx *ctx, int n)
{
    const size_t want = ((size_t)n + kBlock - 1) / kBlock;
    return dim3((unsigned)std::max<size_t>(1, std::min(want, ctx->ovf_threads / kBlock)));
}

// kernel duration of the batch query that just finished, for rt_get_counters
static int batch_time(rt_ctx *ctx)
{
    float ms = 0;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->evk0, ctx->evk1));
    ctx->counters = rt_counters{};
    ctx->counters.kernel_ms = ms; ctx->counters.trace_kernel_ms = ms; ctx->counters.trace_launches = 1;
    return RT_OK;
}

static int batch_common(rt_ctx *ctx, const float *rays6, int n)
{
    if (!ctx) return RT_ERR_INVALID;
    if (!ctx->have_scene) return ctx->fail(RT_ERR_STATE, "no scene uploaded");
    if (n < 0 || (n > 0 && !rays6)) return ctx->fail(RT_ERR_INVALID, "bad ray array");
    return RT_OK;
}

int rt_trace(rt_ctx *ctx, const float *rays6, int n, rt_hit *out)
{
    if (int rc = batch_common(ctx, rays6, n)) return rc;
    if (n == 0) return RT_OK;
    if (!out) return ctx->fail(RT_ERR_INVALID, "null output");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    HIP_TRY(ctx, ctx->d_q_in.reserve((size_t)n * 24));
    HIP_TRY(ctx, ctx->d_q_out.reserve((size_t)n * sizeof(rt_hit)));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_q_in.p, rays6, (size_t)n * 24, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipEventRecord(ctx->evk0, s));
    hipLaunchKernelGGL(k_trace, batch_grid(ctx, n), dim3(kBlock), stack_lds_bytes(ctx->scene), s, ctx->scene,
                       ctx->d_q_in.as<float>(), n, ctx->d_q_out.as<rt_hit>());
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ctx->evk1, s));
    HIP_TRY(ctx, hipMemcpyAsync(out, ctx->d_q_out.p, (size_t)n * sizeof(rt_hit), hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    return batch_time(ctx);
}

int rt_trace_shadow(rt_ctx *ctx, const float *rays6, const float *t_max, int n, int32_t *occluded)
{
    if (int rc = batch_common(ctx, rays6, n)) return rc;
    if (n == 0) return RT_OK;
    if (!t_max || !occluded) return ctx->fail(RT_ERR_INVALID, "null argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    HIP_TRY(ctx, ctx->d_q_in.reserve((size_t)n * 24));
    HIP_TRY(ctx, ctx->d_q_aux.reserve((size_t)n * 4));
    HIP_TRY(ctx, ctx->d_q_out.reserve((size_t)n * 4));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_q_in.p, rays6, (size_t)n * 24, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_q_aux.p, t_max, (size_t)n * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipEventRecord(ctx->evk0, s));
    hipLaunchKernelGGL(k_trace_shadow, batch_grid(ctx, n), dim3(kBlock), stack_lds_bytes(ctx->scene), s, ctx->scene,
                       ctx->d_q_in.as<float>(), ctx->d_q_aux.as<float>(), n, ctx->d_q_out.as<int32_t>());
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ctx->evk1, s));
    HIP_TRY(ctx, hipMemcpyAsync(occluded, ctx->d_q_out.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    return batch_time(ctx);
}

int rt_trace_brief(rt_ctx *ctx, const float *rays6, const float *t_max, int n, rt_hit_brief *out)
{
    if (int rc = batch_common(ctx, rays6, n)) return rc;
    if (n == 0) return RT_OK;
    if (!out) return ctx->fail(RT_ERR_INVALID, "null output");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    std::vector<float4> od((size_t)n * 2);
    for (int i = 0; i < n; ++i) {
        const float *r = rays6 + (size_t)i * 6;
        if (t_max && !(t_max[i] >= 0.0f)) return ctx->fail(RT_ERR_INVALID, "t_max[%d] must be >= 0", i);
        od[i] = make_float4(r[0], r[1], r[2], t_max ? t_max[i] : -1.0f);
        od[(size_t)n + i] = make_float4(r[3], r[4], r[5], 0.0f);
    }
    HIP_TRY(ctx, ctx->d_q_in.reserve((size_t)n * 32));
    HIP_TRY(ctx, ctx->d_q_out.reserve((size_t)n * 16));
    HIP_TRY(ctx, ctx->d_q_aux.reserve((size_t)n * 4));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_q_in.p, od.data(), (size_t)n * 32, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_counters.p, 0, sizeof(DCounters), s));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_queue.p, 0, kQueueBytes, s));
    const void *fn = ctx->query_wide && !ctx->counting ? rays_kernel_wide(scene_levels(ctx)) : rays_kernel(ctx->counting, scene_levels(ctx));
    const size_t lds = stack_lds_bytes(ctx->scene);
    int per_cu = 1;
    HIP_TRY(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, kBlock, lds));
    per_cu = std::max(1, std::min(per_cu, 8));
    const uint64_t want_blocks = ((uint64_t)n + kBlock - 1) / kBlock;
    const dim3 grid((unsigned)std::min<uint64_t>(want_blocks, (uint64_t)ctx->n_cu * per_cu));
    {
        DScene a_scene = ctx->scene;
        static_assert(sizeof(RayListWideIO) == sizeof(RayListIO), "the two list policies are one kernel argument");
        RayListIO a_io{ctx->d_q_in.as<float4>(), ctx->d_q_in.as<float4>() + n, ctx->d_q_out.as<float4>(), ctx->d_q_aux.as<float>()};
        uint32_t a_n = (uint32_t)n;
        DCounters *a_ctr = ctx->d_counters.as<DCounters>();
        unsigned long long *a_queue = ctx->d_queue.as<unsigned long long>();
        uint32_t a_chunk = ray_portion((uint64_t)n, grid.x);
        void *args[] = {&a_scene, &a_io, &a_n, &a_ctr, &a_queue, &a_chunk};
        HIP_TRY(ctx, hipEventRecord(ctx->evk0, s));
        HIP_TRY(ctx, hipLaunchKernel(fn, grid, dim3(kBlock), args, lds, s));
        HIP_TRY(ctx, hipEventRecord(ctx->evk1, s));
    }
    std::vector<float4> h((size_t)n);
    std::vector<float> hv((size_t)n);
    HIP_TRY(ctx, hipMemcpyAsync(h.data(), ctx->d_q_out.p, (size_t)n * 16, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipMemcpyAsync(hv.data(), ctx->d_q_aux.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    DCounters dc{};
    HIP_TRY(ctx, hipMemcpyAsync(&dc, ctx->d_counters.p, sizeof dc, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    float ms = 0;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->evk0, ctx->evk1));
    rt_counters &c = ctx->counters;
    c = rt_counters{};
    c.closest_rays = dc.closest; c.shadow_rays = dc.shadow; c.box_tests = dc.box_tests; c.tri_tests = dc.tri_tests;
    c.kernel_ms = ms; c.trace_kernel_ms = ms; c.trace_launches = 1;
    c.trace_pipeline = ctx->query_wide && !ctx->counting ? 2u : 0u;
    for (int i = 0; i < 16; ++i) c.phase_cycles[i] = dc.phase[i];
    for (int i = 0; i < n; ++i) {
        rt_hit_brief &o = out[i];
        if (t_max) { o = rt_hit_brief{0.0f, h[i].x != 0.0f ? 1 : 0, 0u, 0, 0.0f, 0.0f}; continue; }
        const uint32_t ff = __builtin_bit_cast(uint32_t, h[i].z);
        o.t = h[i].x; o.node = (int32_t)__builtin_bit_cast(uint32_t, h[i].y);
        o.face = ff & 0x7FFFFFFFu; o.front = (int32_t)(ff >> 31); o.u = h[i].w; o.v = hv[i];
    }
    return RT_OK;
}

int rt_trace_brief_noalloc(rt_ctx *ctx, const float *d_rays6, const float *d_t_max, int n, rt_hit_brief *d_out)
{
    if (!ctx) return RT_ERR_INVALID;
    if (!ctx->have_scene) return ctx->fail(RT_ERR_STATE, "no scene uploaded");
    if (n < 0 || (n > 0 && (!d_rays6 || !d_out))) return ctx->fail(RT_ERR_INVALID, "bad ray / answer array");
    if (n > RT_NOALLOC_MAX_RAYS) return ctx->fail(RT_ERR_INVALID, "rt_trace_brief_noalloc takes at most %d rays per call (got %d)", RT_NOALLOC_MAX_RAYS, n);
    if (n == 0) return RT_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    // the lanes' stack levels beyond LDS live in the static scratch: the grid is cut to what it holds
    const DScene &S0 = ctx->scene;
    const size_t ovf_levels = std::max<size_t>(1, (size_t)(S0.stack_levels - S0.lds_levels));
    const size_t max_blocks = kStaticStackEntries / (ovf_levels * kBlock);
    if (max_blocks < 1) return ctx->fail(RT_ERR_UNSUPPORTED, "rt_trace_brief_noalloc: %zu stack levels beyond LDS exceed the static scratch", ovf_levels);
    const void *fn = ctx->scene.n_tlas > 0 ? (const void *)k_rays<false, 0, BriefIO>
                     : ctx->ray_levels <= 1 ? (const void *)k_rays<false, 1, BriefIO> : ctx->ray_levels == 2 ? (const void *)k_rays<false, 2, BriefIO> : (const void *)k_rays<false, 5, BriefIO>;
    const size_t lds = stack_lds_bytes(ctx->scene);
    int per_cu = 1;
    HIP_TRY(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, kBlock, lds));
    per_cu = std::max(1, std::min(per_cu, 8));
    const uint64_t want_blocks = ((uint64_t)n + kBlock - 1) / kBlock;
    const dim3 grid((unsigned)std::min<uint64_t>(std::min<uint64_t>(want_blocks, (uint64_t)ctx->n_cu * per_cu), max_blocks));
    unsigned char *base = ctx->static_scratch; // never the array's name: that is a host-side stand-in
    // the array keeps the last call's contents: queue word and counters must start at zero (the stack is written before it is read)
    HIP_TRY(ctx, hipMemsetAsync(base, 0, kStaticStackOff, s));
    DScene a_scene = ctx->scene;
    a_scene.stack_overflow = (uint2 *)(base + kStaticStackOff);
    BriefIO a_io{d_rays6, d_t_max, d_out};
    uint32_t a_n = (uint32_t)n;
    DCounters *a_ctr = (DCounters *)(base + kStaticCountersOff);
    unsigned long long *a_queue = (unsigned long long *)(base + kStaticQueueOff);
    uint32_t a_chunk = ray_portion((uint64_t)n, grid.x);
    void *args[] = {&a_scene, &a_io, &a_n, &a_ctr, &a_queue, &a_chunk};
    HIP_TRY(ctx, hipEventRecord(ctx->evk0, s));
    HIP_TRY(ctx, hipLaunchKernel(fn, grid, dim3(kBlock), args, lds, s));
    HIP_TRY(ctx, hipEventRecord(ctx->evk1, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    return batch_time(ctx);
}

int rt_trace_paths(rt_ctx *ctx, const float *rays6, int n, uint64_t seed, const rt_render_params *prm, float *radiance3)
{
    if (int rc = batch_common(ctx, rays6, n)) return rc;
    if (int rc = validate_params(ctx, prm)) return rc;
    if (n == 0) return RT_OK;
    if (!radiance3) return ctx->fail(RT_ERR_INVALID, "null output");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    DFrame F;
    memset(&F, 0, sizeof F);
    F.max_bounces = prm->max_bounces; F.bias = prm->bias; F.seed = seed;
    HIP_TRY(ctx, ctx->d_q_in.reserve((size_t)n * 24));
    HIP_TRY(ctx, ctx->d_q_out.reserve((size_t)n * 12));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_q_in.p, rays6, (size_t)n * 24, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_counters.p, 0, sizeof(DCounters), s));
    hipLaunchKernelGGL(k_trace_paths, batch_grid(ctx, n), dim3(kBlock), stack_lds_bytes(ctx->scene), s, ctx->scene, F,
                       ctx->d_q_in.as<float>(), n, ctx->d_q_out.as<float>(), ctx->d_counters.as<DCounters>());
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(radiance3, ctx->d_q_out.p, (size_t)n * 12, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    return RT_OK;
}

} // extern "C"

// ---- photon maps ----
namespace {

const void *photon_kernel(bool write, int ray_levels)
{
    if (ray_levels <= 1) return write ? (const void *)k_photon_paths<true, 1> : (const void *)k_photon_paths<false, 1>;
    if (ray_levels == 2) return write ? (const void *)k_photon_paths<true, 2> : (const void *)k_photon_paths<false, 2>;
    return write ? (const void *)k_photon_paths<true, 5> : (const void *)k_photon_paths<false, 5>;
}

// the kd-tree walk keeps one stack entry per tree level in the lanes' BVH stacks: make them deep enough
int ensure_stack_levels(rt_ctx *ctx, int levels)
{
    DScene &S = ctx->scene;
    if (levels <= S.stack_levels) return RT_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    S.stack_levels = levels;
    // the lookup kernels put the arrays of found photons behind the stacks: leave room for three workgroups per CU
    const int room = (int)((160 * 1024 / 3 - 512 - (int)kHeapLdsBytes) / (kBlock * 8));
    S.lds_levels = std::min(S.stack_levels, std::min(ctx->lds_levels_max, std::max(room, 2)));
    const size_t ovf_levels = (size_t)(S.stack_levels - S.lds_levels);
    HIP_TRY(ctx, ctx->d_stack_ovf.reserve(std::max<size_t>(1, ovf_levels) * ctx->ovf_threads * sizeof(uint2)));
    S.stack_overflow = ctx->d_stack_ovf.as<uint2>();
    return RT_OK;
}

// what a photon lookup keeps outside LDS, per slot of the largest grid: the last level of its heap and the ids of the photons found
int ensure_gather_heaps(rt_ctx *ctx)
{
    const size_t slots = ctx->ovf_threads / kBlock * kHeapSlotStride;
    HIP_TRY(ctx, ctx->d_gather_heaps.reserve(slots * (kHeapGlobalEntries * sizeof(uint2) + RT_GATHER_MAX_PHOTONS * sizeof(uint32_t))));
    ctx->scene.gather_heaps = ctx->d_gather_heaps.as<uint2>();
    ctx->scene.gather_ids = (uint32_t *)(ctx->scene.gather_heaps + slots * kHeapGlobalEntries);
    return RT_OK;
}

int validate_photon_params(rt_ctx *ctx, const rt_photon_params *p, int which)
{
    if (!p) return ctx->fail(RT_ERR_INVALID, "null photon params");
    if (which != RT_MAP_GLOBAL && which != RT_MAP_CAUSTICS) return ctx->fail(RT_ERR_INVALID, "map must be RT_MAP_GLOBAL or RT_MAP_CAUSTICS");
    if (p->flags & ~(RT_PM_DIRECT | RT_PM_INDIRECT | RT_PM_CAUSTICS | RT_PM_MONTE_CARLO)) return ctx->fail(RT_ERR_INVALID, "unknown photon flag bits 0x%x", p->flags);
    if (p->max_segments < 1 || p->max_segments > 4096) return ctx->fail(RT_ERR_INVALID, "max_segments must be 1..4096");
    return RT_OK;
}

bool map_enabled(int flags, int which)
{
    return which == RT_MAP_CAUSTICS ? (flags & RT_PM_CAUSTICS) != 0 : (flags & (RT_PM_DIRECT | RT_PM_INDIRECT | RT_PM_MONTE_CARLO)) != 0;
}

// paths [path_begin, path_begin + n_paths) -> photons at out[base ..], in path order; *stored = photons these paths store
int emit_chunk(rt_ctx *ctx, const rt_photon_params *pp, int which, uint64_t path_begin, uint32_t n_paths, rt_photon *d_out,
               uint64_t base, uint64_t capacity, uint64_t *stored)
{
    hipStream_t s = ctx->stream;
    const uint32_t n_tiles = (n_paths + kScanTile - 1) / kScanTile;
    HIP_TRY(ctx, ctx->d_ph_counts.reserve((size_t)n_paths * 4));
    HIP_TRY(ctx, ctx->d_ph_offsets.reserve((size_t)n_paths * 4));
    HIP_TRY(ctx, ctx->d_ph_tiles.reserve((size_t)n_tiles * 4));
    HIP_TRY(ctx, ctx->d_ph_total.reserve(8));
    const size_t lds = stack_lds_bytes(ctx->scene);
    DScene a_scene = ctx->scene;
    rt_photon_params a_pp = *pp;
    uint32_t *a_counts = ctx->d_ph_counts.as<uint32_t>();
    const uint32_t *a_offsets = ctx->d_ph_offsets.as<uint32_t>();
    DCounters *a_ctr = ctx->d_counters.as<DCounters>();
    unsigned long long *a_queue = ctx->d_queue.as<unsigned long long>();
    for (int pass = 0; pass < 2; ++pass) {
        const void *fn = photon_kernel(pass == 1, ctx->ray_levels);
        int per_cu = 1;
        HIP_TRY(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, kBlock, lds));
        per_cu = std::max(1, std::min(per_cu, 8));
        const dim3 grid((unsigned)std::min<uint64_t>(((uint64_t)n_paths + kBlock - 1) / kBlock, (uint64_t)ctx->n_cu * per_cu));
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_queue.p, 0, kQueueBytes, s));
        void *args[] = {&a_scene, &a_pp, &which, &path_begin, &n_paths, &a_counts, &a_offsets, &base, &d_out, &capacity, &a_ctr, &a_queue};
        HIP_TRY(ctx, hipLaunchKernel(fn, grid, dim3(kBlock), args, lds, s));
        if (pass == 0) {
            hipLaunchKernelGGL(k_scan_tiles, dim3(n_tiles), dim3(kBlock), 0, s, ctx->d_ph_counts.as<uint32_t>(), ctx->d_ph_offsets.as<uint32_t>(),
                               n_paths, ctx->d_ph_tiles.as<uint32_t>());
            hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(kBlock), 0, s, ctx->d_ph_tiles.as<uint32_t>(), n_tiles, ctx->d_ph_total.as<unsigned long long>());
            hipLaunchKernelGGL(k_scan_add, dim3(n_tiles), dim3(kBlock), 0, s, ctx->d_ph_offsets.as<uint32_t>(), n_paths, ctx->d_ph_tiles.as<uint32_t>());
            HIP_TRY(ctx, hipGetLastError());
            unsigned long long total = 0;
            HIP_TRY(ctx, hipMemcpyAsync(&total, ctx->d_ph_total.p, 8, hipMemcpyDeviceToHost, s));
            HIP_TRY(ctx, hipStreamSynchronize(s));
            *stored = total;
            if (total == 0 || base >= capacity) break; // nothing to write
        }
    }
    HIP_TRY(ctx, hipGetLastError());
    return RT_OK;
}

} // namespace

extern "C" {

int rt_photons_emit(rt_ctx *ctx, const rt_photon_params *pp, int which, uint64_t path_begin, uint64_t path_count,
                    rt_photon *out, uint32_t capacity, uint64_t *n_stored, int device_memory)
{
    if (!ctx) return RT_ERR_INVALID;
    if (!ctx->have_scene) return ctx->fail(RT_ERR_STATE, "no scene uploaded");
    if (int rc = validate_photon_params(ctx, pp, which)) return rc;
    if (!n_stored || (capacity > 0 && !out)) return ctx->fail(RT_ERR_INVALID, "null output");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    *n_stored = 0;
    if (!map_enabled(pp->flags, which) || path_count == 0) return RT_OK;
    // a call of its own starts with a clear flag (a request left over from an earlier call, or made while nothing ran,
    // would make the kernels' work queues hand out nothing); inside rt_photons_fill the flag is that call's
    if (ctx->cancellable_depth == 0) if (int rc = begin_cancellable(ctx)) return rc;
    rt_photon *d_out = out;
    if (!device_memory) {
        HIP_TRY(ctx, ctx->d_ph_raw.reserve(std::max<size_t>(1, capacity) * sizeof(rt_photon)));
        d_out = ctx->d_ph_raw.as<rt_photon>();
    }
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_counters.p, 0, sizeof(DCounters), s));
    HIP_TRY(ctx, hipEventRecord(ctx->evk0, s));
    // photon offsets inside a launch are 32-bit: a launch's paths times the photons a path can store must stay below 2^32
    const uint64_t kMaxChunk = std::min<uint64_t>(1ull << 24, 0xFFFFFFFFull / (uint64_t)pp->max_segments);
    uint64_t stored = 0;
    for (uint64_t first = 0; first < path_count; first += kMaxChunk) {
        const uint32_t n = (uint32_t)std::min<uint64_t>(kMaxChunk, path_count - first);
        uint64_t got = 0;
        if (int rc = emit_chunk(ctx, pp, which, path_begin + first, n, d_out, stored, capacity, &got)) return rc;
        stored += got;
    }
    HIP_TRY(ctx, hipEventRecord(ctx->evk1, s));
    const uint64_t written = std::min<uint64_t>(stored, capacity);
    if (!device_memory && written) HIP_TRY(ctx, hipMemcpyAsync(out, d_out, written * sizeof(rt_photon), hipMemcpyDeviceToHost, s));
    DCounters dc{};
    HIP_TRY(ctx, hipMemcpyAsync(&dc, ctx->d_counters.p, sizeof dc, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    if (cancelled(ctx)) return ctx->fail(RT_ERR_CANCELLED, "cancelled by rt_cancel while tracing photon paths"); // the counts are those of a stopped queue
    *n_stored = stored;
    float ms = 0;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->evk0, ctx->evk1));
    ctx->counters = rt_counters{};
    ctx->counters.closest_rays = dc.closest; // both passes: the rays actually traced
    ctx->counters.camera_paths = dc.paths;
    ctx->counters.kernel_ms = ms; ctx->counters.trace_kernel_ms = ms; ctx->counters.trace_launches = 2;
    return RT_OK;
}

int rt_photons_build(rt_ctx *ctx, int which, const rt_photon *photons, uint32_t n, int device_memory)
{
    if (!ctx) return RT_ERR_INVALID;
    if (!ctx->have_scene) return ctx->fail(RT_ERR_STATE, "no scene uploaded");
    if (which != RT_MAP_GLOBAL && which != RT_MAP_CAUSTICS) return ctx->fail(RT_ERR_INVALID, "map must be RT_MAP_GLOBAL or RT_MAP_CAUSTICS");
    if (n > 0 && !photons) return ctx->fail(RT_ERR_INVALID, "null photon array");
    if (n > (1u << 30)) return ctx->fail(RT_ERR_INVALID, "at most 2^30 photons per map");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->pmap_n[which] = 0;
    ctx->pmap_flags = -1;
    if (n == 0) return RT_OK;
    hipStream_t st = ctx->stream;
    DeviceBuf staged;
    const rt_photon *d_in = photons;
    if (!device_memory) {
        HIP_TRY(ctx, staged.reserve((size_t)n * sizeof(rt_photon)));
        HIP_TRY(ctx, hipMemcpyAsync(staged.p, photons, (size_t)n * sizeof(rt_photon), hipMemcpyHostToDevice, st));
        d_in = staged.as<rt_photon>();
    }
    HIP_TRY(ctx, ctx->d_pmap[which].reserve(((size_t)n + 1) * sizeof(rt_photon)));
    // working arrays of the build: positions as three arrays plus the photon ids, the scratch of the partition steps,
    // the boxes of the nodes that have children (heap index <= n / 2), six words for the root box and one flag
    DeviceBuf work;
    const size_t n1 = ((size_t)n + 1 + 63) & ~(size_t)63, n_boxes = ((size_t)n / 2 + 2) * 6;
    const size_t crowd_bytes = 3 * 256 * sizeof(unsigned long long); // per group of a crowd launch: two words and one per workgroup
    HIP_TRY(ctx, work.reserve((6 * n1 + n_boxes + 64) * sizeof(float) + crowd_bytes));
    rtd::KdArrays A{};
    A.x = work.as<float>(); A.y = A.x + n1; A.z = A.y + n1;
    A.id = (uint32_t *)(A.z + n1); A.lpos = A.id + n1; A.rpos = A.lpos + n1;
    A.boxes = (float *)(A.rpos + n1);
    uint32_t *d_box_bits = (uint32_t *)(A.boxes + n_boxes), *d_bad = d_box_bits + 6;
    unsigned long long *d_crowd = (unsigned long long *)(d_box_bits + 64);
    A.in = d_in; A.out = ctx->pmap(which); A.n = n;
    A.scale = 1.0f / (float)(int)n; // ScalePhotonPowers(1 / NumPhotons()) (src/photonmap.h:82)
    const uint32_t init_words[7] = {0x80000000u, 0x80000000u, 0x80000000u, 0x80000000u, 0x80000000u, 0x80000000u, 0u};
    HIP_TRY(ctx, hipEventRecord(ctx->evk0, st));
    // One launch per level.  While a level has fewer segments than the chip has compute units and they are long, a
    // crowd of workgroups shares each segment; then a workgroup per segment, sized to the level's largest segment (its
    // leftmost); segments of kKdTiny photons or fewer are finished by one lane each.
    // A crowd's workgroups meet at barriers of their own and must all be resident; if something else holds compute
    // units (another process, another stream) they can wait in vain, notice, and leave a half-partitioned segment.
    // The build then starts over from the caller's photons with a workgroup per segment on every level: slower
    // (about ten times for ten million photons), the same tree.
    const bool test_abort = getenv("RT_KD_TEST_ABORT") != nullptr; // (test hook: treat the first attempt as given up)
    uint32_t grid_max = 1;
    while (grid_max * 2 <= (uint32_t)std::min(ctx->n_cu, 256)) grid_max *= 2; // every workgroup of a crowd launch must be resident
    for (int attempt = 0; attempt < 2; ++attempt) {
        const bool crowds = attempt == 0;
        HIP_TRY(ctx, hipMemcpyAsync(d_box_bits, init_words, sizeof init_words, hipMemcpyHostToDevice, st));
        {
            const uint32_t blocks = (uint32_t)std::min<size_t>(((size_t)n + 255) / 256, (size_t)ctx->n_cu * 8);
            hipLaunchKernelGGL(rtd::k_kd_init, dim3(blocks), dim3(256), 0, st, A, d_box_bits, d_bad);
            hipLaunchKernelGGL(rtd::k_kd_root_box, dim3(1), dim3(64), 0, st, A, (const uint32_t *)d_box_bits);
        }
        uint32_t bad = 0;
        HIP_TRY(ctx, hipMemcpyAsync(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));
        if (bad) return ctx->fail(RT_ERR_INVALID, "a photon has a non-finite position"); // the quick-select would not terminate
        for (int level = 0; level < rtd::kKdMaxLevels && n >= 2; ++level) {
            const uint32_t first = 1u << level;
            if (first > n) break;
            const uint32_t nodes = std::min<uint32_t>(first, n - first + 1);
            int s0, e0;
            if (!rtd::kd_segment(n, first, s0, e0)) break; // nothing left with two or more photons
            const int largest = e0 - s0 + 1;
            if (largest <= rtd::kKdTiny) {
                hipLaunchKernelGGL(rtd::k_kd_tiny, dim3((nodes + 63) / 64), dim3(64), 0, st, A, first, nodes);
                break;
            }
            const uint32_t crowd = grid_max >> std::min(level, 31);
            if (crowds && crowd >= 2 && largest > rtd::kKdAloneBelow) {
                HIP_TRY(ctx, hipMemsetAsync(d_crowd, 0, crowd_bytes, st));
                hipLaunchKernelGGL((rtd::k_kd_level_crowd<1024, 8>), dim3(first * crowd), dim3(1024), 0, st, A, first, crowd, d_crowd, d_bad);
            }
            else if (largest > 16384) hipLaunchKernelGGL((rtd::k_kd_level<1024, 8>), dim3(nodes), dim3(1024), 0, st, A, first, nodes);
            else if (largest > 2048) hipLaunchKernelGGL((rtd::k_kd_level<256, 8>), dim3(nodes), dim3(256), 0, st, A, first, nodes);
            else hipLaunchKernelGGL((rtd::k_kd_level<64, 4>), dim3(nodes), dim3(64), 0, st, A, first, nodes);
        }
        HIP_TRY(ctx, hipMemcpyAsync(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipStreamSynchronize(st));
        if (crowds && test_abort) bad = 1;
        if (!bad) break;
        if (!crowds) return ctx->fail(RT_ERR_HIP, "kd-tree build failed without shared segments");
    }
    HIP_TRY(ctx, hipEventRecord(ctx->evk1, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    float build_ms = 0;
    HIP_TRY(ctx, hipEventElapsedTime(&build_ms, ctx->evk0, ctx->evk1));
    ctx->counters = rt_counters{};
    ctx->counters.kernel_ms = build_ms;
    if (int rc = ensure_gather_heaps(ctx)) return rc;
    int levels = 1;
    while ((1u << levels) <= n) ++levels; // tree height
    if (int rc = ensure_stack_levels(ctx, levels + 1)) return rc;
    ctx->pmap_n[which] = n;
    return RT_OK;
}

int rt_photons_fill(rt_ctx *ctx, const rt_photon_params *pp)
{
    if (!ctx) return RT_ERR_INVALID;
    if (!ctx->have_scene) return ctx->fail(RT_ERR_STATE, "no scene uploaded");
    if (int rc = validate_photon_params(ctx, pp, RT_MAP_GLOBAL)) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int rc = begin_cancellable(ctx)) return rc;
    struct Nest { rt_ctx *c; Nest(rt_ctx *x) : c(x) { ++c->cancellable_depth; } ~Nest() { --c->cancellable_depth; } } nest(ctx);
    double emit_ms = 0.0;
    uint64_t rays = 0, paths_total = 0;
    for (int which = 0; which < 2; ++which) {
        const uint32_t cap = which == RT_MAP_CAUSTICS ? pp->n_caustic : pp->n_global;
        if (!map_enabled(pp->flags, which) || cap == 0) {
            if (int rc = rt_photons_build(ctx, which, nullptr, 0, 0)) return rc;
            continue;
        }
        DeviceBuf raw;
        HIP_TRY(ctx, raw.reserve((size_t)cap * sizeof(rt_photon)));
        uint64_t stored = 0, path = 0, round = 1u << 16;
        int barren = 0;
        while (stored < cap) {
            if (cancelled(ctx)) return ctx->fail(RT_ERR_CANCELLED, "cancelled by rt_cancel while filling the photon maps");
            uint64_t got = 0;
            if (int rc = rt_photons_emit(ctx, pp, which, path, round, raw.as<rt_photon>() + stored, (uint32_t)(cap - stored), &got, 1)) return rc;
            emit_ms += ctx->counters.kernel_ms; rays += ctx->counters.closest_rays; paths_total += round;
            path += round;
            stored += std::min<uint64_t>(got, cap - stored);
            // next round: what is still missing at the yield seen so far, with a little margin
            const double yield = (double)std::max<uint64_t>(stored, 1) / (double)path;
            round = (uint64_t)std::min<double>(1 << 24, std::max<double>(1 << 16, 1.05 * (double)(cap - stored) / yield + 1024));
            barren = got == 0 ? barren + 1 : 0;
            if (barren >= 8 && stored == 0) return ctx->fail(RT_ERR_UNSUPPORTED, "no photon stored after %llu paths: the scene has no surface that takes photons for this map", (unsigned long long)path);
        }
        if (int rc = rt_photons_build(ctx, which, raw.as<rt_photon>(), cap, 1)) return rc;
    }
    ctx->pmap_flags = pp->flags;
    ctx->counters = rt_counters{};
    ctx->counters.closest_rays = rays; ctx->counters.camera_paths = paths_total;
    ctx->counters.kernel_ms = emit_ms; ctx->counters.trace_kernel_ms = emit_ms;
    return RT_OK;
}

int rt_debug_samplers(rt_ctx *ctx, int which, const rt_sampler_query *queries, int n, rt_sampler_answer *out)
{
    if (!ctx) return RT_ERR_INVALID;
    if (!ctx->have_scene) return ctx->fail(RT_ERR_STATE, "no scene uploaded");
    if (which < RT_SAMPLER_LIGHT || which > RT_SAMPLER_PHOTON_EMIT) return ctx->fail(RT_ERR_INVALID, "unknown sampler %d", which);
    if (n < 0 || (n > 0 && (!queries || !out))) return ctx->fail(RT_ERR_INVALID, "bad query arrays");
    if (n == 0) return RT_OK;
    uint32_t n_materials = (uint32_t)(ctx->d_materials.bytes / sizeof(rt_material));
    for (int i = 0; i < n; ++i)
        if (queries[i].material < 0 || (uint32_t)queries[i].material >= n_materials) return ctx->fail(RT_ERR_INVALID, "query %d: material index out of range", i);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    HIP_TRY(ctx, ctx->d_q_in.reserve((size_t)n * sizeof(rt_sampler_query)));
    HIP_TRY(ctx, ctx->d_q_out.reserve((size_t)n * sizeof(rt_sampler_answer)));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_q_in.p, queries, (size_t)n * sizeof(rt_sampler_query), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_debug_samplers, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, ctx->scene, which,
                       ctx->d_q_in.as<rt_sampler_query>(), n, ctx->d_q_out.as<rt_sampler_answer>());
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(out, ctx->d_q_out.p, (size_t)n * sizeof(rt_sampler_answer), hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    return RT_OK;
}

int rt_debug_rng(rt_ctx *ctx, const uint64_t *seq, const uint64_t *seed, int n_streams, int n_draws, uint32_t *words, float *floats)
{
    if (!ctx) return RT_ERR_INVALID;
    if (n_streams < 0 || n_draws < 0 || (n_streams > 0 && n_draws > 0 && (!seq || !seed || !words || !floats))) return ctx->fail(RT_ERR_INVALID, "bad arrays");
    if (n_streams == 0 || n_draws == 0) return RT_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const size_t total = (size_t)n_streams * n_draws;
    HIP_TRY(ctx, ctx->d_q_in.reserve((size_t)n_streams * 16));
    HIP_TRY(ctx, ctx->d_q_out.reserve(total * 8));
    uint64_t *d_seq = ctx->d_q_in.as<uint64_t>(), *d_seed = d_seq + n_streams;
    uint32_t *d_words = ctx->d_q_out.as<uint32_t>();
    float *d_floats = (float *)(d_words + total);
    HIP_TRY(ctx, hipMemcpyAsync(d_seq, seq, (size_t)n_streams * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(d_seed, seed, (size_t)n_streams * 8, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_debug_rng, dim3((n_streams + kBlock - 1) / kBlock), dim3(kBlock), 0, s, d_seq, d_seed, n_streams, n_draws, d_words, d_floats);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(words, d_words, total * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipMemcpyAsync(floats, d_floats, total * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    return RT_OK;
}

int rt_photons_download(rt_ctx *ctx, int which, rt_photon *dst, uint32_t *n)
{
    if (!ctx) return RT_ERR_INVALID;
    if (which != RT_MAP_GLOBAL && which != RT_MAP_CAUSTICS) return ctx->fail(RT_ERR_INVALID, "map must be RT_MAP_GLOBAL or RT_MAP_CAUSTICS");
    if (!n) return ctx->fail(RT_ERR_INVALID, "null count");
    *n = ctx->pmap_n[which];
    if (!dst || *n == 0) return RT_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpy(dst, ctx->pmap(which), (size_t)*n * sizeof(rt_photon), hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_photons_irradiance(rt_ctx *ctx, int which, const float *pos3, const float *normal3, int n, float radius, float *irradiance3)
{
    if (!ctx) return RT_ERR_INVALID;
    if (!ctx->have_scene) return ctx->fail(RT_ERR_STATE, "no scene uploaded");
    if (which != RT_MAP_GLOBAL && which != RT_MAP_CAUSTICS) return ctx->fail(RT_ERR_INVALID, "map must be RT_MAP_GLOBAL or RT_MAP_CAUSTICS");
    if (n < 0 || (n > 0 && (!pos3 || !normal3 || !irradiance3))) return ctx->fail(RT_ERR_INVALID, "bad query arrays");
    if (!(radius > 0.0f)) return ctx->fail(RT_ERR_INVALID, "radius must be > 0");
    if (n == 0) return RT_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    HIP_TRY(ctx, ctx->d_q_in.reserve((size_t)n * 12));
    HIP_TRY(ctx, ctx->d_q_aux.reserve((size_t)n * 12));
    HIP_TRY(ctx, ctx->d_q_out.reserve((size_t)n * 12));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_q_in.p, pos3, (size_t)n * 12, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_q_aux.p, normal3, (size_t)n * 12, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_counters.p, 0, sizeof(DCounters), s));
    HIP_TRY(ctx, hipEventRecord(ctx->evk0, s));
    hipLaunchKernelGGL(k_photon_irradiance, batch_grid(ctx, n), dim3(kBlock), gather_lds_bytes(ctx->scene), s, ctx->scene,
                       ctx->pmap(which), ctx->pmap_n[which], ctx->d_q_in.as<float>(), ctx->d_q_aux.as<float>(), n, radius,
                       ctx->d_q_out.as<float>(), ctx->d_counters.as<DCounters>());
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ctx->evk1, s));
    HIP_TRY(ctx, hipMemcpyAsync(irradiance3, ctx->d_q_out.p, (size_t)n * 12, hipMemcpyDeviceToHost, s));
    DCounters dc{};
    HIP_TRY(ctx, hipMemcpyAsync(&dc, ctx->d_counters.p, sizeof dc, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    if (int rc = batch_time(ctx)) return rc;
    ctx->counters.gathers = dc.gathers; ctx->counters.photon_visits = dc.photon_visits;
    return RT_OK;
}

} // extern "C"
