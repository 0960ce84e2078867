// Drives include/hip_renderer.hpp the way the reference's main() drives its Renderer
// (src/main.cpp:388-621): LoadScene, BeginRender, poll IsRendering, read the RenderImage.
//   hip_renderer_demo scene.xml W H spp seed out.bin [n_global n_caustic]
// With the two photon counts the frame is the Whitted-style shader of the project-12 snapshot reading photon maps
// (indirect light + caustics, 3 bounces) that FillPhotonMaps traces first.
// out.bin = rgb (3 B/px) | z (f32) | sample counts (i32) | object ids (i32) | one hit record + one shadow flag
#include "hip_renderer.hpp"

#include <chrono>
#include <cstdlib>
#include <cstring>

int main(int argc, char **argv)
{
    if (argc != 7 && argc != 9) { std::fprintf(stderr, "usage: %s scene.xml W H spp seed out.bin [n_global n_caustic]\n", argv[0]); return 2; }
    rt::HipRenderer renderer(0);
    if (!renderer.Ok()) return 3;
    if (renderer.LoadScene("/nonexistent/scene.xml")) return 4; // must fail politely, like the reference
    if (!renderer.LoadScene(argv[1])) return 5;
    renderer.GetCamera().width = std::atoi(argv[2]);
    renderer.GetCamera().height = std::atoi(argv[3]);
    if (!renderer.SetCamera()) return 6;
    renderer.Params().min_spp = renderer.Params().max_spp = std::atoi(argv[4]);
    renderer.Params().seed = std::strtoull(argv[5], nullptr, 10);

    if (argc == 9) {
        rt_photon_params pp{};
        pp.flags = RT_PM_INDIRECT | RT_PM_CAUSTICS;
        pp.n_global = (uint32_t)std::atoi(argv[7]); pp.n_caustic = (uint32_t)std::atoi(argv[8]);
        pp.max_segments = 64; pp.seed = renderer.Params().seed;
        if (!renderer.FillPhotonMaps(pp)) return 9;
        if (renderer.GetPhotonMap().size() != pp.n_global || renderer.GetCausticsMap().size() != pp.n_caustic) return 10;
        renderer.Params().mode = RT_MODE_WHITTED;
        renderer.Params().max_bounces = 3;
    }

    renderer.BeginRender();
    while (renderer.IsRendering()) std::this_thread::sleep_for(std::chrono::milliseconds(1));
    renderer.StopRender();
    rt::RenderImage &im = renderer.GetRenderImage();
    if (!im.IsRenderDone() || renderer.LastStatus() != RT_OK) return 7;
    im.ComputeZBufferImage();
    im.ComputeSampleCountImage();

    // the camera's central ray through the query interface
    const rt_camera &c = renderer.GetCamera();
    rt::Ray r;
    for (int i = 0; i < 3; ++i) { r.p[i] = c.pos[i]; r.dir[i] = c.dir[i]; }
    rt::HitInfo h;
    const bool hit = renderer.TraceRay(r, h);
    const bool shadowed = renderer.TraceShadowRay(r, 1e30f);

    FILE *f = std::fopen(argv[6], "wb");
    if (!f) return 8;
    const size_t n = (size_t)im.GetWidth() * im.GetHeight();
    std::fwrite(im.GetPixels(), 3, n, f);
    std::fwrite(im.GetZBuffer(), 4, n, f);
    std::fwrite(im.GetSampleCount(), 4, n, f);
    std::fwrite(im.GetObjectIds(), 4, n, f);
    const float rec[4] = {hit ? h.z : -1.0f, (float)h.node, h.front ? 1.0f : 0.0f, shadowed ? 1.0f : 0.0f};
    std::fwrite(rec, 4, 4, f);
    std::fclose(f);
    std::printf("rendered %dx%d, centre ray: hit=%d node=%d z=%g shadowed=%d\n", im.GetWidth(), im.GetHeight(), hit, h.node, h.z, shadowed);
    return 0;
}
