// Host-side C++ mirror of the reference's Renderer / RenderImage pair
// (src/renderer.h:38-106, 205-238) on top of the C ABI of rt_mi355x.h: the same
// method names, argument meaning and error behaviour (bool results, a line on
// stderr, no exceptions), so code written against `renderer.LoadScene(...);
// renderer.BeginRender(); ... GetRenderImage().GetPixels()` reads the same here.
// Header-only; link with librt_mi355x.so.
//
// Differences a caller can see, all forced by where the work runs:
//   * BeginRender() starts ONE host thread that drives the GPU for the whole frame
//     (the reference starts hardware_concurrency() tile threads, src/main.cpp:599-606);
//     the pixels of the RenderImage are valid when the frame is complete.  While it runs,
//     GetNumRenderedPixels() / IsRenderDone() follow the GPU's progress (rt_progress): finished
//     pixels plus, during the first sampling round, the share of its camera paths that have ended.
//   * StopRender() asks the GPU to stop (rt_cancel) and waits for the worker thread; a stopped
//     frame leaves LastStatus() == RT_ERR_CANCELLED.
//   * One GPU context serves one call at a time: TraceRay / TraceShadowRay / TraceRays and the
//     photon-map calls wait for a frame in flight (they take the same mutex as the worker).
//   * Rays and hits are plain structs of floats, not cyVector types.
#ifndef RT_HIP_RENDERER_HPP
#define RT_HIP_RENDERER_HPP

#include "rt_mi355x.h"

#include <atomic>
#include <chrono>
#include <cfloat>
#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <limits>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace rt {

constexpr float BIGFLOAT = FLT_MAX; // src/scene.h: "no hit" depth

struct Color24 { uint8_t r, g, b; };
struct Ray { float p[3], dir[3]; };
// the fields of the reference's HitInfo that its intersectors fill (src/scene.h:78-93)
struct HitInfo {
    float p[3] = {0, 0, 0};
    float z = BIGFLOAT;
    int node = -1;      // depth-first index of the node that was hit, -1 none
    float N[3] = {0, 0, 0};
    float uvw[3] = {0.5f, 0.5f, 0.5f};
    bool front = true;
    bool light = false; // a spherical light was the closest thing hit
    int lightIndex = -1;
};

// src/renderer.h:38-106
class RenderImage {
    std::vector<Color24> img;
    std::vector<float> zbuffer;
    std::vector<uint8_t> zbufferImg;
    std::vector<int> sampleCount;
    std::vector<uint8_t> sampleCountImg;
    std::vector<int32_t> objectId; // extra plane: node index of the first sample's camera-ray hit
    int width = 0, height = 0;
    std::atomic<int> numRenderedPixels{0};

public:
    void Init(int w, int h)
    {
        width = w; height = h;
        const size_t size = (size_t)w * h;
        img.assign(size, Color24{0, 0, 0});
        zbuffer.assign(size, BIGFLOAT);
        zbufferImg.assign(size, 0);
        sampleCount.assign(size, 0);
        sampleCountImg.assign(size, 0);
        objectId.assign(size, -1);
        ResetNumRenderedPixels();
    }
    int GetWidth() const { return width; }
    int GetHeight() const { return height; }
    Color24 *GetPixels() { return img.data(); }
    float *GetZBuffer() { return zbuffer.data(); }
    uint8_t *GetZBufferImage() { return zbufferImg.data(); }
    int *GetSampleCount() { return sampleCount.data(); }
    uint8_t *GetSampleCountImage() { return sampleCountImg.data(); }
    int32_t *GetObjectIds() { return objectId.data(); }

    void ResetNumRenderedPixels() { numRenderedPixels = 0; }
    int GetNumRenderedPixels() const { return numRenderedPixels; }
    bool IsRenderDone() const { return numRenderedPixels >= width * height; }
    void IncrementNumRenderPixel(int n) { numRenderedPixels += n; }

    // 8-bit views of the depth and sample-count planes (src/renderer.h:84-105): values spread over 0..255 between the
    // smallest and largest entry that is not the plane's "nothing here" value; depth is shown near = bright.
    void ComputeZBufferImage() { SpreadToBytes(zbuffer, BIGFLOAT, true, zbufferImg); }
    int ComputeSampleCountImage() { return SpreadToBytes(sampleCount, 0, false, sampleCountImg); }

private:
    template <typename T> static T SpreadToBytes(const std::vector<T> &plane, T nothing, bool nearIsBright, std::vector<uint8_t> &bytes)
    {
        T lowest = std::numeric_limits<T>::max(), highest = T(0);
        for (size_t i = 0; i < plane.size(); ++i)
            if (plane[i] != nothing) { lowest = std::min(lowest, plane[i]); highest = std::max(highest, plane[i]); }
        const float span = float(highest - lowest);
        for (size_t i = 0; i < plane.size(); ++i) {
            int level = 0;
            if (plane[i] != nothing) {
                const float part = float(plane[i] - lowest) / span;
                level = int((nearIsBright ? 1 - part : part) * 255);
                level = std::max(0, std::min(255, level));
            }
            bytes[i] = (uint8_t)level;
        }
        return highest;
    }
};

class HipRenderer {
    rt_ctx *ctx = nullptr;
    rt_scene_host *host = nullptr;
    rt_camera camera{};
    RenderImage renderImage;
    std::string sceneFile;
    std::atomic<bool> isRendering{false};
    std::atomic<bool> stopRequested{false}; // StopRender has been called for the frame the worker was started for
    std::thread worker;
    mutable std::mutex gpu; // one call at a time on the context (rt_cancel and rt_progress excepted)
    rt_render_params params{};
    std::atomic<int> lastStatus{RT_OK};
    bool gpuBvh = false, faceMaterials = false;

    bool Fail(const char *what, const char *msg) const
    {
        std::fprintf(stderr, "ERROR: %s: %s\n", what, msg ? msg : "");
        return false;
    }

public:
    // the constants the reference compiles in (src/main.cpp:93,107,164,220,273-274,369)
    explicit HipRenderer(int device = 0)
    {
        params.mode = RT_MODE_PATH;
        params.min_spp = 128; params.max_spp = 128; params.delta_max = 0.01f;
        params.max_bounces = 50; params.bias = 0.002f; params.halton_period = 16;
        params.tile_size = 16; params.tile_rank = 0; params.tile_world = 1; params.seed = 7;
        params.photon_flags = 0; params.gather_radius = 3.0f;
        if (rt_init(device, &ctx) != RT_OK) { Fail("rt_init", rt_last_error(nullptr)); ctx = nullptr; }
    }
    ~HipRenderer()
    {
        StopRender();
        if (host) rt_scene_host_free(host);
        if (ctx) rt_shutdown(ctx);
    }
    HipRenderer(const HipRenderer &) = delete;
    HipRenderer &operator=(const HipRenderer &) = delete;

    bool Ok() const { return ctx != nullptr; }
    rt_render_params &Params() { return params; }   // spp, bounces, seed, tile shard
    rt_camera &GetCamera() { return camera; }       // call SetCamera() after changing it
    RenderImage &GetRenderImage() { return renderImage; }
    std::string const &SceneFileName() const { return sceneFile; }
    int LastStatus() const { return lastStatus; }

    // Renderer::LoadScene (src/renderer.cpp, src/xmlload.cpp:35-78): parse, build, upload
    bool LoadScene(char const *sceneFilename)
    {
        if (!ctx) return Fail("LoadScene", "no GPU context");
        StopRender();
        char err[512] = "";
        rt_scene_host *h = nullptr;
        if (rt_scene_load_xml(sceneFilename, &h, err, sizeof err) != RT_OK) return Fail(sceneFilename, err);
        if (host) rt_scene_host_free(host);
        host = h;
        sceneFile = sceneFilename;
        rt_scene_host_set_face_materials(host, faceMaterials ? 1 : 0);
        if (gpuBvh) rt_scene_host_set_gpu_bvh(host, 1);
        if ((lastStatus = rt_scene_upload(ctx, rt_scene_host_desc(host))) != RT_OK) return Fail("rt_scene_upload", rt_last_error(ctx));
        camera = *rt_scene_host_camera(host);
        return SetCamera();
    }
    // Scenes loaded from here on go to the GPU without the reference's BVH: the library builds its own there (milliseconds
    // instead of the seconds cyBVH::Build takes for millions of triangles; same depths and object ids, frames a few per
    // cent slower).  Call before LoadScene.
    void UseGpuBvh(bool on) { gpuBvh = on; }
    // Objects that bring their own materials (the reference's MultiMtl): false (default) shades every face with the first
    // member, as the reference does - its intersectors never set HitInfo::mtlID; true: each face with the member it names.
    void UseFaceMaterials(bool on) { faceMaterials = on; }

    bool SetCamera()
    {
        if (!ctx) return false;
        if ((lastStatus = rt_camera_set(ctx, &camera)) != RT_OK) return Fail("rt_camera_set", rt_last_error(ctx));
        renderImage.Init(camera.width, camera.height);
        return true;
    }

    // One frame, blocking: what the tile threads of the reference produce together (src/main.cpp:271-385)
    bool RenderFrame()
    {
        if (!ctx || !host) return Fail("RenderFrame", "no scene loaded");
        std::lock_guard<std::mutex> lock(gpu);
        if (stopRequested) { lastStatus = RT_ERR_CANCELLED; return false; } // stopped before the frame began
        renderImage.ResetNumRenderedPixels();
        rt_frame_out out{};
        out.rgb = (uint8_t *)renderImage.GetPixels();
        out.z = renderImage.GetZBuffer();
        out.sample_count = (int32_t *)renderImage.GetSampleCount();
        out.object_id = renderImage.GetObjectIds();
        if ((lastStatus = rt_render(ctx, &params, &out)) != RT_OK) return lastStatus == RT_ERR_CANCELLED ? false : Fail("rt_render", rt_last_error(ctx));
        renderImage.ResetNumRenderedPixels();
        renderImage.IncrementNumRenderPixel(renderImage.GetWidth() * renderImage.GetHeight());
        return true;
    }
    // Pixels rendered so far, for a progress bar (the viewer polls RenderImage::GetNumRenderedPixels from its idle
    // loop, src/viewport.cpp:568-600): call it from the polling thread; it also refreshes the RenderImage's counter.
    int PollRenderedPixels()
    {
        if (!ctx || !isRendering) return renderImage.GetNumRenderedPixels();
        rt_progress_info pi{};
        if (rt_progress(ctx, &pi) != RT_OK || !pi.rendering) return renderImage.GetNumRenderedPixels();
        uint64_t px = pi.pixels_done;
        if (pi.round == 1 && pi.round_items > 0) px += (pi.pixels_total - pi.pixels_done) * pi.round_items_done / pi.round_items;
        const int total = renderImage.GetWidth() * renderImage.GetHeight();
        const int shown = (int)std::min<uint64_t>(px, (uint64_t)(total > 0 ? total - 1 : 0)); // "done" only when the worker says so
        const int have = renderImage.GetNumRenderedPixels();
        if (shown > have) renderImage.IncrementNumRenderPixel(shown - have);
        return renderImage.GetNumRenderedPixels();
    }
    // "Generates one or more rendering threads and begins rendering. Returns immediately." (src/renderer.h:228)
    void BeginRender()
    {
        StopRender();
        stopRequested = false;
        isRendering = true;
        worker = std::thread([this] { RenderFrame(); isRendering = false; });
    }
    // "Stops the current rendering process. It should wait till rendering threads stop." (src/renderer.h:229)
    void StopRender()
    {
        if (worker.joinable()) {
            // rt_render clears the flag of an earlier call when it starts, so a request that arrives while the worker
            // is still on its way into rt_render would be lost: the worker looks at stopRequested before it calls, and
            // the request is repeated until the worker is out (no new work for the GPU: it returns within milliseconds)
            stopRequested = true;
            while (isRendering && ctx) {
                rt_cancel(ctx);
                std::this_thread::sleep_for(std::chrono::microseconds(200));
            }
            worker.join();
        }
        isRendering = false;
    }
    // waits for the frame in flight instead of stopping it
    void WaitRender()
    {
        if (worker.joinable()) worker.join();
        isRendering = false;
    }
    bool IsRendering() const { return isRendering; }

    // Renderer::TraceRay / TraceShadowRay (src/main.cpp:115-125); both hit sides, as the reference's callers ask
    bool TraceRay(Ray const &ray, HitInfo &hInfo) const
    {
        rt_hit h;
        std::lock_guard<std::mutex> lock(gpu);
        if (!ctx || rt_trace(ctx, ray.p, 1, &h) != RT_OK) return false; // p and dir are contiguous: 6 floats
        if (h.node == -1) return false;
        hInfo.z = h.z; hInfo.node = h.node; hInfo.front = h.front != 0;
        hInfo.light = h.node < -1; hInfo.lightIndex = h.node < -1 ? -2 - h.node : -1;
        for (int i = 0; i < 3; ++i) { hInfo.p[i] = h.p[i]; hInfo.N[i] = h.n[i]; hInfo.uvw[i] = h.uvw[i]; }
        return true;
    }
    bool TraceShadowRay(Ray const &ray, float t_max) const
    {
        int32_t occluded = 0;
        std::lock_guard<std::mutex> lock(gpu);
        if (!ctx || rt_trace_shadow(ctx, ray.p, &t_max, 1, &occluded) != RT_OK) return false;
        return occluded != 0;
    }
    // ---- photon maps: Renderer::Get/SetPhotonMap, Get/SetCausticsMap (src/renderer.h:221-224) and the emission
    // loops of the project-12 snapshot's main() (project12.zip src/main.cpp:262-470).  The reference hands the
    // renderer a PhotonMap it filled itself; here the library fills and owns the maps, FillPhotonMaps replaces
    // the loops, and GetPhotonMap / GetCausticsMap return a host copy in PhotonMap::GetPhotons() order.
    bool FillPhotonMaps(rt_photon_params const &pp)
    {
        if (!ctx || !host) return Fail("FillPhotonMaps", "no scene loaded");
        std::lock_guard<std::mutex> lock(gpu);
        if ((lastStatus = rt_photons_fill(ctx, &pp)) != RT_OK) return Fail("rt_photons_fill", rt_last_error(ctx));
        params.photon_flags = pp.flags; // the shader reads what was filled
        return true;
    }
    // Renderer::SetPhotonMap with photons the caller traced (emission order, unscaled powers)
    bool SetPhotonMap(rt_photon const *photons, uint32_t n) { return SetMap(RT_MAP_GLOBAL, photons, n); }
    bool SetCausticsMap(rt_photon const *photons, uint32_t n) { return SetMap(RT_MAP_CAUSTICS, photons, n); }
    std::vector<rt_photon> GetPhotonMap() const { return GetMap(RT_MAP_GLOBAL); }
    std::vector<rt_photon> GetCausticsMap() const { return GetMap(RT_MAP_CAUSTICS); }

private:
    bool SetMap(int which, rt_photon const *photons, uint32_t n)
    {
        if (!ctx) return Fail("SetPhotonMap", "no GPU context");
        std::lock_guard<std::mutex> lock(gpu);
        if ((lastStatus = rt_photons_build(ctx, which, photons, n, 0)) != RT_OK) return Fail("rt_photons_build", rt_last_error(ctx));
        return true;
    }
    std::vector<rt_photon> GetMap(int which) const
    {
        uint32_t n = 0;
        std::vector<rt_photon> out;
        std::lock_guard<std::mutex> lock(gpu);
        if (!ctx || rt_photons_download(ctx, which, nullptr, &n) != RT_OK || n == 0) return out;
        out.resize(n);
        if (rt_photons_download(ctx, which, out.data(), &n) != RT_OK) out.clear();
        return out;
    }

public:
    // many rays at once: one launch instead of one per ray
    bool TraceRays(const Ray *rays, int n, rt_hit *hits) const
    {
        std::lock_guard<std::mutex> lock(gpu);
        return ctx && rt_trace(ctx, rays[0].p, n, hits) == RT_OK;
    }
};

static_assert(sizeof(Ray) == 6 * sizeof(float), "Ray must be six packed floats");
static_assert(sizeof(Color24) == 3, "Color24 must be three bytes");

} // namespace rt
#endif
