"""GPU parity tests: the HIP path, called through the C ABI, against
  (1) the committed golden fixtures (outputs of the reference itself), and
  (2) the CPU oracle on other seeds / sizes.
Bar: primary-hit object ids, z and every closest-hit / any-hit query bit-exact
(integer and IEEE add/mul/div/sqrt work only); shaded 8-bit images PSNR >= 45 dB
at equal spp (the shading path goes through sinf/cosf/powf, whose device
implementations differ from glibc's in the last ulp)."""
import numpy as np
import pytest

import oraclelib as O
from raytracer_amd import abi, lib

pytestmark = pytest.mark.gpu

PSNR_MIN = 45.0


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def ctx():
    c = lib.Context(0)
    yield c
    c.close()


def test_library_is_the_in_tree_hip_build():
    import os
    assert os.path.samefile(lib.LIB_PATH, os.path.join(O.ROOT, "raytracer_amd", "librt_mi355x.so"))
    with open("/proc/self/maps") as fp:
        lib.load()
        assert "librt_mi355x.so" in fp.read()


@pytest.mark.parametrize("name", ["kat_trace_a", "kat_trace_nested", "kat_trace_mesh"])
def test_trace_queries_vs_reference(ctx, name):
    g = O.load_golden(name)
    s = O.scene_from_golden(g)
    ctx.upload(s.desc)
    hits = ctx.trace(g["kat_rays"])
    ref_hit = g["kat_flags"][:, 0].astype(bool)
    assert np.array_equal(hits["node"], g["kat_id"]), "hit node ids"
    assert np.array_equal(bits(hits["z"]), bits(g["kat_hit"][:, 0])), "z"
    assert np.array_equal(hits["front"][ref_hit], g["kat_flags"][ref_hit, 1])
    on_obj = ref_hit & (g["kat_id"] >= 0)
    assert np.array_equal(bits(hits["p"][on_obj]), bits(g["kat_hit"][on_obj, 1:4])), "hit point"
    assert np.array_equal(bits(hits["n"][on_obj]), bits(g["kat_hit"][on_obj, 4:7])), "normal"
    # uvw of spheres goes through atan2/asin: tolerance, not bits
    np.testing.assert_allclose(hits["uvw"][on_obj], g["kat_hit"][on_obj, 7:10], rtol=0, atol=2e-6)
    occ = ctx.trace_shadow(g["kat_rays"], g["kat_tmax"])
    assert np.array_equal(occ, g["kat_flags"][:, 2]), "shadow rays"


@pytest.mark.parametrize("name", ["kat_trace_a", "kat_trace_nested", "kat_trace_mesh"])
def test_ray_list_kernel_vs_reference(ctx, name):
    """rt_trace_brief (persistent ray-list kernel with ray-level regeneration) against the reference's answers."""
    g = O.load_golden(name)
    s = O.scene_from_golden(g)
    ctx.upload(s.desc)
    rays = np.tile(g["kat_rays"], (5, 1))  # more rays than lanes of a workgroup: lanes take several rays in turn
    ref_id, ref_z = np.tile(g["kat_id"], 5), np.tile(g["kat_hit"][:, 0], 5)
    got = ctx.trace_brief(rays)
    assert np.array_equal(got["node"], ref_id), "hit node ids"
    assert np.array_equal(bits(got["t"]), bits(ref_z)), "z"
    ref_hit = np.tile(g["kat_flags"][:, 0].astype(bool), 5)
    assert np.array_equal(got["front"][ref_hit], np.tile(g["kat_flags"][:, 1], 5)[ref_hit])
    full = ctx.trace(rays)  # the brief answer is the winner the full query interpolates from
    on_mesh = np.array([s.desc.nodes[i].prim == abi.RT_PRIM_MESH if i >= 0 else False for i in got["node"]])
    if on_mesh.any():
        assert (got["face"][on_mesh] < s.desc.meshes[0].nf).all()
        w = 1.0 - got["u"][on_mesh] - got["v"][on_mesh]
        assert (w >= -1e-6).all() and np.isfinite(full["n"][on_mesh]).all()
    occ = ctx.trace_brief(rays, np.tile(g["kat_tmax"], 5))
    assert np.array_equal(occ["node"], np.tile(g["kat_flags"][:, 2], 5)), "shadow rays"
    c = ctx.counters()
    assert c.shadow_rays == len(rays) and c.kernel_ms > 0
    with pytest.raises(lib.RtError):
        ctx.trace_brief(rays[:4], np.array([1.0, -1.0, 2.0, 3.0], np.float32))


@pytest.mark.parametrize("name", ["kat_trace_a", "kat_trace_mesh"])
def test_ray_list_lengths_vs_reference(ctx, name):
    """Ragged list lengths around the portions the waves take from the list (64 .. 1024 rays, chosen from the length):
    every prefix must give the reference's answers, closest hits and shadow rays."""
    g = O.load_golden(name)
    s = O.scene_from_golden(g)
    ctx.upload(s.desc)
    reps = 40
    rays = np.tile(g["kat_rays"], (reps, 1))
    ref_id, ref_z = np.tile(g["kat_id"], reps), np.tile(g["kat_hit"][:, 0], reps)
    tmax, ref_occ = np.tile(g["kat_tmax"], reps), np.tile(g["kat_flags"][:, 2], reps)
    for n in (1, 2, 63, 64, 65, 127, 129, 1000, 4097, 65537, len(rays)):
        n = min(n, len(rays))
        got = ctx.trace_brief(rays[:n])
        assert np.array_equal(got["node"], ref_id[:n]), f"hit node ids of {n} rays"
        assert np.array_equal(bits(got["t"]), bits(ref_z[:n])), f"z of {n} rays"
        occ = ctx.trace_brief(rays[:n], tmax[:n])
        assert np.array_equal(occ["node"], ref_occ[:n]), f"shadow rays, {n} of them"
    assert len(ctx.trace_brief(rays[:0])["node"]) == 0


def test_path_radiance_vs_reference(ctx):
    g = O.load_golden("kat_paths_a")
    s = O.scene_from_golden(g)
    ctx.upload(s.desc)
    rad = ctx.trace_paths(g["path_rays"], 3, abi.default_params())
    ref = g["path_radiance"]
    close = np.all(np.abs(rad - ref) <= 1e-4 * np.maximum(1.0, np.abs(ref)), axis=1)
    # a path diverges completely when a last-ulp difference flips a branch; that must stay rare
    assert close.mean() >= 0.995, f"only {close.mean():.4f} of the paths agree"
    assert abs(rad.mean() - ref.mean()) <= 0.01 * ref.mean()


def render_gpu(ctx, s, prm, w, h):
    ctx.upload(s.desc)
    ctx.set_camera(s.camera, w, h)
    return ctx.render(prm)


@pytest.mark.parametrize("name", ["image_a_160x90_s16", "image_a_96x54_adaptive", "image_nested_96x54_s8",
                                  "image_spheres_96x54_s8", "image_mesh_96x54_s8"])
def test_images_vs_reference(ctx, name):
    g = O.load_golden(name)
    s = O.scene_from_golden(g)
    seed, mn, mx, w, h = [int(x) for x in g["params"]]
    prm = abi.default_params(seed=seed, min_spp=mn, max_spp=mx, delta_max=float(g["delta"]))
    fr = render_gpu(ctx, s, prm, w, h)
    pinhole = float(g["scene_camera"][11]) == 0.0
    if pinhole:
        assert np.array_equal(fr.object_id, g["object_id"]), "primary-hit ids"
        assert np.array_equal(bits(fr.z), bits(g["z"])), "z-buffer"
    else:
        # a thin-lens camera puts cosf/sinf into the ray origin: ids may differ on silhouettes
        assert (fr.object_id == g["object_id"]).mean() > 0.995
    same_n = (fr.sample_count == g["sample_count"]).mean()
    assert same_n >= (1.0 if mn == mx else 0.97), f"sample counts agree on {same_n:.4f}"
    p = O.psnr(fr.rgb, g["rgb"])
    assert p >= PSNR_MIN, f"PSNR {p:.2f} dB vs the reference image"


@pytest.mark.parametrize("seed,w,h,spp", [(11, 320, 180, 32), (12, 200, 113, 8)])
def test_scene_a_vs_oracle(ctx, seed, w, h, spp):
    s = O.scene_from_golden(O.load_golden("scene_a"))
    prm = abi.default_params(seed=seed, min_spp=spp, max_spp=spp)
    gpu = render_gpu(ctx, s, prm, w, h)
    cpu = O.render(s, prm, w, h)
    assert np.array_equal(gpu.object_id, cpu.object_id)
    assert np.array_equal(bits(gpu.z), bits(cpu.z))
    assert np.array_equal(gpu.sample_count, cpu.sample_count)
    p = O.psnr(gpu.rgb, cpu.rgb)
    assert p >= PSNR_MIN, f"PSNR {p:.2f} dB"
    c = ctx.counters()
    assert c.camera_paths == w * h * spp
    # ray counts follow the paths; a handful of diverged paths may change them slightly
    assert abs(int(c.closest_rays) - int(cpu.stats.closest)) <= 2e-3 * cpu.stats.closest
    assert abs(int(c.shadow_rays) - int(cpu.stats.shadow)) <= 2e-3 * cpu.stats.shadow


def test_mesh_scene_vs_oracle_with_counters(ctx):
    g = O.load_golden("scene_mesh")
    s = O.scene_from_golden(g)
    prm = abi.default_params(seed=21, min_spp=4, max_spp=4)
    ctx.set_counting(True)
    try:
        gpu = render_gpu(ctx, s, prm, 160, 90)
        c = ctx.counters()
    finally:
        ctx.set_counting(False)
    cpu = O.render(s, prm, 160, 90)
    assert np.array_equal(gpu.object_id, cpu.object_id)
    assert np.array_equal(bits(gpu.z), bits(cpu.z))
    assert O.psnr(gpu.rgb, cpu.rgb) >= PSNR_MIN
    # the GPU walks the BVH in the reference's order, so the visit counts match the oracle's
    assert abs(int(c.box_tests) - int(cpu.stats.box_tests)) <= 5e-3 * cpu.stats.box_tests
    assert abs(int(c.tri_tests) - int(cpu.stats.tri_tests)) <= 5e-3 * cpu.stats.tri_tests


WHITTED = ["whitted_a_96x54_b0", "whitted_a_96x54_b5", "whitted_a_64x36_adaptive", "whitted_spheres_96x54_b0",
           "whitted_mesh_96x54_b5", "whitted_lights_96x54_b5"]  # the last: ambient + directional + spherical light


@pytest.mark.parametrize("name", WHITTED)
def test_whitted_images_vs_reference(ctx, name):
    """RT_MODE_WHITTED (k_whitted) against images of the reference's project-12 snapshot: primary-hit ids and
    z bit-exact; the shaded image within the PSNR bar (cosf/sinf feed the shadow-fan and lobe directions, powf
    the highlights, so single samples can flip at shadow edges)."""
    g = O.load_golden(name)
    s = O.scene_from_golden(g)
    seed, mn, mx, w, h, bounces = [int(x) for x in g["params"]]
    prm = abi.default_params(seed=seed, min_spp=mn, max_spp=mx, delta_max=float(g["delta"]), max_bounces=bounces,
                             mode=abi.RT_MODE_WHITTED)
    fr = render_gpu(ctx, s, prm, w, h)
    assert np.array_equal(fr.object_id, g["object_id"]), "primary-hit ids"
    assert np.array_equal(bits(fr.z), bits(g["z"])), "z-buffer"
    same_n = (fr.sample_count == g["sample_count"]).mean()
    assert same_n >= (1.0 if mn == mx else 0.97), f"sample counts agree on {same_n:.4f}"
    p = O.psnr(fr.rgb, g["rgb"])
    assert p >= PSNR_MIN, f"PSNR {p:.2f} dB vs the reference image"
    # most pixels agree to float rounding in linear radiance
    close = np.all(np.abs(fr.linear - g["linear"]) <= 1e-4 * np.maximum(1.0, np.abs(g["linear"])), axis=-1)
    assert close.mean() >= 0.97, f"only {close.mean():.4f} of the pixels agree in linear radiance"


@pytest.mark.parametrize("bounces,w,h,spp", [(0, 240, 135, 4), (3, 160, 90, 2)])
def test_whitted_spheres_vs_oracle(ctx, bounces, w, h, spp):
    """The sphere/plane configuration (67 nodes) through the Whitted-style shader against the oracle, with ray counts."""
    from raytracer_amd import scenes
    hs = lib.HostScene.from_string(scenes.spheres_scene_xml(width=w, height=h))
    prm = abi.default_params(seed=7, min_spp=spp, max_spp=spp, max_bounces=bounces, mode=abi.RT_MODE_WHITTED)
    gpu = render_gpu(ctx, hs, prm, w, h)
    cpu = O.render(hs, prm, w, h)
    assert np.array_equal(gpu.object_id, cpu.object_id)
    assert np.array_equal(bits(gpu.z), bits(cpu.z))
    assert O.psnr(gpu.rgb, cpu.rgb) >= PSNR_MIN
    c = ctx.counters()
    assert c.camera_paths == w * h * spp
    assert abs(int(c.closest_rays) - int(cpu.stats.closest)) <= 5e-3 * cpu.stats.closest
    assert abs(int(c.shadow_rays) - int(cpu.stats.shadow)) <= 5e-3 * cpu.stats.shadow


@pytest.mark.parametrize("bounces,w,h,spp", [(0, 200, 112, 2), (2, 128, 72, 2)])
def test_whitted_nested_transforms_vs_oracle(ctx, bounces, w, h, spp):
    """Rotated, non-uniformly scaled and nested nodes through the Whitted-style shader: the kernel skips objects
    whose padded world-space box no ray of a wave reaches, the oracle tests every object as the reference does, and
    depth and ids must not differ in a single pixel."""
    from raytracer_amd import scenes
    hs = lib.HostScene.from_string(scenes.nested_scene_xml(w, h).replace('<dof value="0.6"/>', ""))
    prm = abi.default_params(seed=11, min_spp=spp, max_spp=spp, max_bounces=bounces, mode=abi.RT_MODE_WHITTED)
    gpu = render_gpu(ctx, hs, prm, w, h)
    cpu = O.render(hs, prm, w, h)
    assert np.array_equal(gpu.object_id, cpu.object_id)
    assert np.array_equal(bits(gpu.z), bits(cpu.z))
    assert O.psnr(gpu.rgb, cpu.rgb) >= PSNR_MIN
    c = ctx.counters()
    assert abs(int(c.closest_rays) - int(cpu.stats.closest)) <= 5e-3 * cpu.stats.closest
    assert abs(int(c.shadow_rays) - int(cpu.stats.shadow)) <= 5e-3 * cpu.stats.shadow


def test_whitted_tile_shards_and_determinism(ctx):
    """Whitted-style frames with photon maps: rendering twice gives the same bits, and three tile shards tile the frame."""
    s = O.scene_from_golden(O.load_golden("scene_a"))
    w, h = 200, 120
    ctx.upload(s.desc)
    ctx.set_camera(s.camera, w, h)
    pp = abi.PhotonParams(flags=abi.RT_PM_INDIRECT | abi.RT_PM_CAUSTICS, n_global=5000, n_caustic=1000, max_segments=64, seed=3)
    ctx.photons_fill(pp)
    kw = dict(seed=3, min_spp=2, max_spp=6, delta_max=0.05, max_bounces=3, mode=abi.RT_MODE_WHITTED, photon_flags=pp.flags)
    whole = ctx.render(abi.default_params(**kw))
    again = ctx.render(abi.default_params(**kw))
    merged = lib.Frame(w, h)
    for rank in range(3):
        ctx.render(abi.default_params(tile_rank=rank, tile_world=3, **kw), merged)
    for k in ("rgb", "z", "sample_count", "object_id"):
        assert np.array_equal(getattr(whole, k), getattr(again, k)), f"{k} not deterministic"
        assert np.array_equal(getattr(whole, k), getattr(merged, k)), f"{k}: shards do not tile the frame"
    assert np.array_equal(bits(whole.linear), bits(again.linear))
    assert len(np.unique(whole.sample_count)) > 1  # the variance stop test released pixels at different counts


def test_whitted_rejects_deep_recursion(ctx):
    s = O.scene_from_golden(O.load_golden("scene_a"))
    ctx.upload(s.desc)
    ctx.set_camera(s.camera, 8, 8)
    with pytest.raises(lib.RtError) as e:
        ctx.render(abi.default_params(mode=abi.RT_MODE_WHITTED, max_bounces=9, min_spp=1, max_spp=1))
    assert e.value.code == abi.RT_ERR_INVALID


def photon_words(a):
    return np.ascontiguousarray(a).view(np.uint32).reshape(-1, 8)


@pytest.mark.parametrize("name", ["photons_a_indirect_caustics", "photons_a_direct"])
def test_photon_maps_vs_reference(ctx, name):
    """Photon emission (k_photon_paths), the kd-tree build and the irradiance estimate (k_photon_irradiance)
    against the project-12 snapshot's maps.  Emission goes through double sin/cos/pow whose device forms can
    differ from glibc's in the last bit of a double, which is invisible in all but a vanishing share of the
    photons; everything else is integer or IEEE float work and must be bit-exact."""
    g = O.load_golden(name)
    s = O.scene_from_golden(g)
    seed, ng, nc, seg, flags = [int(x) for x in g["params"]]
    pp = abi.PhotonParams(flags=flags, n_global=ng, n_caustic=nc, max_segments=seg, seed=seed)
    ctx.upload(s.desc)
    for which, key in ((abi.RT_MAP_GLOBAL, "global"), (abi.RT_MAP_CAUSTICS, "caustic")):
        ref_raw, ref_tree = g[f"pm_{key}_raw"], g[f"pm_{key}_tree"]
        paths = int(g["pm_paths"][which])
        if len(ref_raw) == 0:
            got, stored = ctx.photons_emit(pp, which, 0, 1000, 16)
            assert stored == 0 and len(got) == 0
            continue
        got, stored = ctx.photons_emit(pp, which, 0, paths, len(ref_raw))
        assert stored >= len(ref_raw) and len(got) == len(ref_raw)
        same = (photon_words(got) == ref_raw).all(axis=1)
        assert same.mean() >= 0.999, f"{key}: only {same.mean():.4f} of the emitted photons are bit-identical"
        # split emission (what several GPUs do): concatenation in path order is the same list
        parts, total = [], 0
        for b in range(0, paths, 1500):
            part, st = ctx.photons_emit(pp, which, b, min(1500, paths - b), len(ref_raw))
            parts.append(part[:st])
            total += st
        assert total == stored
        assert np.array_equal(photon_words(np.concatenate(parts))[:len(ref_raw)], photon_words(got))
        # the tree from the reference's own photon list must be the reference's tree, node for node
        ctx.photons_build(which, ref_raw.view(abi.PHOTON_DTYPE).reshape(-1))
        assert np.array_equal(photon_words(ctx.photons_download(which)), ref_tree), f"{key}: balanced kd-tree"
        col = 0 if which == abi.RT_MAP_GLOBAL else 3
        e = ctx.photons_irradiance(which, g["pm_query"][:, :3], g["pm_query"][:, 3:], float(g["radius"]))
        assert np.array_equal(bits(e), bits(g["pm_irradiance"][:, col:col + 3])), f"{key}: irradiance estimates"
    # rt_photons_fill: rounds of paths until the maps are full
    ctx.photons_fill(pp)
    for which, key in ((abi.RT_MAP_GLOBAL, "global"), (abi.RT_MAP_CAUSTICS, "caustic")):
        tree = ctx.photons_download(which)
        assert len(tree) == len(g[f"pm_{key}_tree"])
        if len(tree):
            same = (photon_words(tree) == g[f"pm_{key}_tree"]).all(axis=1)
            assert same.mean() >= 0.99, f"{key}: filled map differs from the reference's ({same.mean():.4f})"


@pytest.mark.parametrize("name", ["whitted_pm_a_64x36_b5", "whitted_pm_a_64x36_b0", "whitted_pd_a_64x36_b5"])
def test_whitted_photon_images_vs_reference(ctx, name):
    g = O.load_golden(name)
    s = O.scene_from_golden(g)
    seed, mn, mx, w, h, bounces, ng, nc, seg, flags = [int(x) for x in g["params"]]
    pp = abi.PhotonParams(flags=flags, n_global=ng, n_caustic=nc, max_segments=seg, seed=seed)
    ctx.upload(s.desc)
    ctx.set_camera(s.camera, w, h)
    prm = abi.default_params(seed=seed, min_spp=mn, max_spp=mx, delta_max=float(g["delta"]), max_bounces=bounces,
                             mode=abi.RT_MODE_WHITTED, photon_flags=flags, gather_radius=3.0)
    with pytest.raises(lib.RtError) as e:  # maps not built yet
        ctx.render(prm)
    assert e.value.code == abi.RT_ERR_STATE
    ctx.photons_fill(pp)
    ctx.set_counting(True)
    try:
        fr = ctx.render(prm)
        c = ctx.counters()
    finally:
        ctx.set_counting(False)
    assert np.array_equal(fr.object_id, g["object_id"]), "primary-hit ids"
    assert np.array_equal(bits(fr.z), bits(g["z"])), "z-buffer"
    p = O.psnr(fr.rgb, g["rgb"])
    assert p >= PSNR_MIN, f"PSNR {p:.2f} dB vs the reference image"
    close = np.all(np.abs(fr.linear - g["linear"]) <= 1e-4 * np.maximum(1.0, np.abs(g["linear"])), axis=-1)
    assert close.mean() >= 0.97, f"only {close.mean():.4f} of the pixels agree in linear radiance"
    # the oracle counts the lookups the same way
    O.fill_photon_maps(s, pp)
    cpu = O.render(s, prm, w, h)
    assert abs(int(c.gathers) - int(cpu.stats.gathers)) <= 5e-3 * cpu.stats.gathers
    assert abs(int(c.photon_visits) - int(cpu.stats.photon_visits)) <= 1e-2 * cpu.stats.photon_visits


def test_pipelines_agree(monkeypatch):
    """The lock-step path kernel and the wavefront pipeline (ray-list kernel with ray-level regeneration + shading
    kernel, compacted lists between them) run the same per-path arithmetic: every output plane and the ray counts
    must be identical, whichever of the two a scene gets by default."""
    ctxs = {}
    for mode in ("mega", "wave"):
        monkeypatch.setenv("RT_PIPELINE", mode)
        ctxs[mode] = lib.Context(0)
    monkeypatch.delenv("RT_PIPELINE")
    try:
        for name, (w, h, mn, mx) in {"scene_a": (200, 120, 6, 6), "scene_mesh": (160, 90, 3, 20),
                                      "image_nested_96x54_s8": (96, 54, 4, 4)}.items():
            s = O.scene_from_golden(O.load_golden(name))
            prm = abi.default_params(seed=17, min_spp=mn, max_spp=mx, delta_max=0.02)
            frames, counts = {}, {}
            for mode, c in ctxs.items():
                frames[mode] = render_gpu(c, s, prm, w, h)
                k = c.counters()
                counts[mode] = (k.closest_rays, k.shadow_rays, k.camera_paths)
            for other in ("wave",):
                for plane in ("rgb", "z", "sample_count", "object_id"):
                    assert np.array_equal(getattr(frames["mega"], plane), getattr(frames[other], plane)), (name, other, plane)
                assert np.array_equal(bits(frames["mega"].linear), bits(frames[other].linear)), (name, other)
                assert counts["mega"] == counts[other], (name, other)
    finally:
        for c in ctxs.values():
            c.close()


def test_a_frame_launched_from_its_profile_is_the_same_frame(monkeypatch):
    """The wavefront pipeline launches a round it has seen before (same view, same samples) from the list lengths it
    recorded then, without waiting for the device after every bounce: the second and third rendering of a frame must
    equal the first bit for bit, equal a context that never does this (RT_WF_REPLAY=0), and a frame that differs
    (another seed: other list lengths) must come out right after them - adaptive rounds and tile shards included."""
    monkeypatch.setenv("RT_PIPELINE", "wave")
    fast = lib.Context(0)
    monkeypatch.setenv("RT_WF_REPLAY", "0")
    plain = lib.Context(0)
    monkeypatch.delenv("RT_WF_REPLAY")
    monkeypatch.delenv("RT_PIPELINE")
    try:
        s = O.scene_from_golden(O.load_golden("scene_mesh"))
        w, h = 160, 90
        for c in (fast, plain):
            c.upload(s.desc)
            c.set_camera(s.camera, w, h)
        for kw in (dict(seed=5, min_spp=3, max_spp=20, delta_max=0.02), dict(seed=5, min_spp=4, max_spp=4, tile_rank=1, tile_world=3),
                   dict(seed=6, min_spp=3, max_spp=20, delta_max=0.02), dict(seed=5, min_spp=3, max_spp=20, delta_max=0.02)):
            want = plain.render(abi.default_params(**kw))
            for again in range(3):
                got = fast.render(abi.default_params(**kw))
                for plane in ("rgb", "z", "sample_count", "object_id"):
                    assert np.array_equal(getattr(got, plane), getattr(want, plane)), (kw, again, plane)
                assert np.array_equal(bits(got.linear), bits(want.linear)), (kw, again)
                k, kp = fast.counters(), plain.counters()
                assert (k.closest_rays, k.shadow_rays, k.camera_paths) == (kp.closest_rays, kp.shadow_rays, kp.camera_paths)
    finally:
        fast.close()
        plain.close()


def test_pixel_order_does_not_change_the_frame(monkeypatch):
    """Dealing a frame's pixels expensive-first (scheduling keys from one camera ray per pixel + radix sort; the
    default for tile shards) only changes when a pixel is rendered: every plane, the adaptive sample counts and
    the ray counts are those of tile order, for whole frames and for shards."""
    ctxs = {}
    for order in ("0", "1"):
        monkeypatch.setenv("RT_ORDER", order)
        ctxs[order] = lib.Context(0)
    monkeypatch.delenv("RT_ORDER")
    try:
        for name, (w, h, mn, mx) in {"scene_a": (200, 120, 4, 12), "scene_mesh": (160, 90, 3, 20)}.items():
            s = O.scene_from_golden(O.load_golden(name))
            for world in (1, 3):
                frames, counts = {}, {}
                for order, c in ctxs.items():
                    c.upload(s.desc)
                    c.set_camera(s.camera, w, h)
                    merged = lib.Frame(w, h)
                    total = [0, 0, 0]
                    for rank in range(world):
                        c.render(abi.default_params(seed=23, min_spp=mn, max_spp=mx, delta_max=0.02, tile_rank=rank, tile_world=world), merged)
                        k = c.counters()
                        total = [total[0] + k.closest_rays, total[1] + k.shadow_rays, total[2] + k.camera_paths]
                    frames[order], counts[order] = merged, total
                for plane in ("rgb", "z", "sample_count", "object_id"):
                    assert np.array_equal(getattr(frames["0"], plane), getattr(frames["1"], plane)), (name, world, plane)
                assert np.array_equal(bits(frames["0"].linear), bits(frames["1"].linear)), (name, world)
                assert counts["0"] == counts["1"], (name, world)
    finally:
        for c in ctxs.values():
            c.close()


def test_edge_cases(ctx):
    s = O.scene_from_golden(O.load_golden("scene_a"))
    # image smaller than a tile, odd sizes, one sample, no bounces
    for (w, h, mn, mx, bounces) in [(1, 1, 1, 1, 50), (17, 5, 1, 3, 50), (33, 31, 2, 2, 0), (16, 16, 4, 4, 1)]:
        prm = abi.default_params(seed=5, min_spp=mn, max_spp=mx, max_bounces=bounces, delta_max=0.05)
        gpu = render_gpu(ctx, s, prm, w, h)
        cpu = O.render(s, prm, w, h)
        assert np.array_equal(gpu.object_id, cpu.object_id), (w, h)
        assert np.array_equal(bits(gpu.z), bits(cpu.z)), (w, h)
        assert np.abs(gpu.rgb.astype(int) - cpu.rgb.astype(int)).mean() < 1.0, (w, h)
        if bounces == 0:
            assert not gpu.linear.any()


def test_full_size_properties(ctx):
    """1920x1080: properties that hold at any size - determinism, tile shards tile the
    image exactly (per-pixel seeding makes a pixel independent of who renders it)."""
    s = O.scene_from_golden(O.load_golden("scene_a"))
    w, h = 1920, 1080
    prm = abi.default_params(seed=1, min_spp=2, max_spp=2)
    whole = render_gpu(ctx, s, prm, w, h)
    again = ctx.render(prm)
    for k in ("rgb", "z", "sample_count", "object_id", "linear"):
        assert np.array_equal(getattr(whole, k), getattr(again, k)), f"{k} not deterministic"
    merged = lib.Frame(w, h)
    for rank in range(3):
        ctx.render(abi.default_params(seed=1, min_spp=2, max_spp=2, tile_rank=rank, tile_world=3), merged)
    for k in ("rgb", "z", "sample_count", "object_id"):
        assert np.array_equal(getattr(whole, k), getattr(merged, k)), f"{k}: shards do not tile the frame"
    assert (whole.sample_count == 2).all()
    assert (whole.object_id >= -2).all() and (whole.object_id < s.desc.n_nodes).all()
    # a sub-rectangle checked against the oracle at full resolution
    cpu = O.render(s, prm, w, h, y0=500, y1=516)
    assert np.array_equal(whole.object_id[500:516], cpu.object_id[500:516])
    assert np.array_equal(bits(whole.z[500:516]), bits(cpu.z[500:516]))


def test_error_paths(ctx):
    s = O.scene_from_golden(O.load_golden("scene_a"))
    ctx.upload(s.desc)
    ctx.set_camera(s.camera, 8, 8)
    with pytest.raises(lib.RtError) as e:
        ctx.render(abi.default_params(min_spp=4, max_spp=2))
    assert e.value.code == abi.RT_ERR_INVALID
    with pytest.raises(lib.RtError):
        ctx.render(abi.default_params(tile_rank=2, tile_world=2))
    bad = O.scene_from_golden(O.load_golden("scene_a"))
    bad.desc.nodes[2].material = -1
    with pytest.raises(lib.RtError) as e:
        ctx.upload(bad.desc)
    assert e.value.code == abi.RT_ERR_UNSUPPORTED and "material" in str(e.value)
    fresh = lib.Context(0)
    with pytest.raises(lib.RtError) as e:
        fresh.render(abi.default_params())
    assert e.value.code == abi.RT_ERR_STATE
    fresh.close()


def test_cpp_renderer_mirror_matches_binding(tmp_path, ctx):
    """include/hip_renderer.hpp (the C++ mirror of the reference's Renderer: LoadScene, BeginRender,
    IsRendering, StopRender, TraceRay, TraceShadowRay) must give the frame the ctypes binding gives."""
    import os
    import subprocess
    from raytracer_amd import scenes
    exe = os.path.join(O.ROOT, "raytracer_amd", "hip_renderer_demo")
    assert os.path.exists(exe), "build it with make -C raytracer_amd/csrc"
    # a scene file with an OBJ mesh next to it, a nested transform hierarchy would do as well
    scenes.write_obj(str(tmp_path / "mesh.obj"), scenes.heightfield(24, 12))
    xml = scenes.mesh_scene_xml("mesh.obj", 320, 180)
    (tmp_path / "scene.xml").write_text(xml)
    w, h, spp, seed = 96, 54, 4, 19
    r = subprocess.run([exe, "scene.xml", str(w), str(h), str(spp), str(seed), "out.bin"], cwd=tmp_path,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ERROR" in r.stderr  # the deliberately missing scene file is reported the reference's way
    raw = np.fromfile(tmp_path / "out.bin", np.uint8)
    n = w * h
    rgb = raw[:3 * n].reshape(h, w, 3)
    z = raw[3 * n:7 * n].view(np.float32).reshape(h, w)
    cnt = raw[7 * n:11 * n].view(np.int32).reshape(h, w)
    ids = raw[11 * n:15 * n].view(np.int32).reshape(h, w)
    rec = raw[15 * n:].view(np.float32)

    cwd = os.getcwd()
    os.chdir(tmp_path)  # OBJ paths are relative to the working directory, as in the reference
    try:
        hs = lib.HostScene.from_file("scene.xml")
    finally:
        os.chdir(cwd)
    ctx.upload(hs.desc)
    ctx.set_camera(hs.camera, w, h)
    fr = ctx.render(abi.default_params(seed=seed, min_spp=spp, max_spp=spp))
    assert np.array_equal(rgb, fr.rgb) and np.array_equal(bits(z), bits(fr.z))
    assert np.array_equal(cnt, fr.sample_count) and np.array_equal(ids, fr.object_id)
    cam = hs.camera
    ray = np.array([[*cam.pos, *cam.dir]], np.float32)
    hit = ctx.trace(ray)[0]
    assert (rec[0] == hit["z"] or (hit["node"] == -1 and rec[0] == -1.0)) and int(rec[1]) == hit["node"]
    assert int(rec[3]) == int(ctx.trace_shadow(ray, np.array([1e30], np.float32))[0])


def test_cpp_renderer_mirror_photon_maps(tmp_path, ctx):
    """HipRenderer::FillPhotonMaps / GetPhotonMap and a Whitted-style frame reading the maps, against the binding."""
    import os
    import subprocess
    from raytracer_amd import scenes
    exe = os.path.join(O.ROOT, "raytracer_amd", "hip_renderer_demo")
    (tmp_path / "scene.xml").write_text(scenes.nested_scene_xml(64, 36).replace('<dof value="0.6"/>', ""))
    w, h, spp, seed, ng, nc = 64, 36, 2, 23, 1500, 300
    r = subprocess.run([exe, "scene.xml", str(w), str(h), str(spp), str(seed), "out.bin", str(ng), str(nc)], cwd=tmp_path,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = np.fromfile(tmp_path / "out.bin", np.uint8)
    rgb = raw[:3 * w * h].reshape(h, w, 3)
    hs = lib.HostScene.from_file(str(tmp_path / "scene.xml"))
    ctx.upload(hs.desc)
    ctx.set_camera(hs.camera, w, h)
    pp = abi.PhotonParams(flags=abi.RT_PM_INDIRECT | abi.RT_PM_CAUSTICS, n_global=ng, n_caustic=nc, max_segments=64, seed=seed)
    ctx.photons_fill(pp)
    fr = ctx.render(abi.default_params(seed=seed, min_spp=spp, max_spp=spp, max_bounces=3, mode=abi.RT_MODE_WHITTED,
                                       photon_flags=pp.flags, gather_radius=3.0))
    assert np.array_equal(rgb, fr.rgb)
    assert fr.rgb.std() > 5  # a picture, not a constant


@pytest.mark.parametrize("name", ["image_textured_96x54_s8", "whitted_textured_96x54_b5"])
def test_textured_images_vs_reference(ctx, name, tmp_path):
    """Checkerboard and image textures (material diffuse colour, background, environment) on the GPU against the
    reference's images: path tracer at HEAD and the snapshot's Whitted-style shader."""
    from raytracer_amd import scenes
    g = O.load_golden(name)
    scenes.write_ppm(str(tmp_path / "tex.ppm"), g["image"])
    hs = lib.HostScene.from_string(bytes(g["xml"]).decode(), str(tmp_path))
    seed, mn, mx, w, h = [int(x) for x in g["params"][:5]]
    kw = dict(seed=seed, min_spp=mn, max_spp=mx, delta_max=float(g["delta"]))
    if name.startswith("whitted"):
        kw.update(mode=abi.RT_MODE_WHITTED, max_bounces=int(g["params"][5]))
    fr = render_gpu(ctx, hs, abi.default_params(**kw), w, h)
    assert np.array_equal(fr.object_id, g["object_id"]), "primary-hit ids"
    assert np.array_equal(bits(fr.z), bits(g["z"])), "z-buffer"
    p = O.psnr(fr.rgb, g["rgb"])
    assert p >= PSNR_MIN, f"PSNR {p:.2f} dB vs the reference image"
    miss = g["object_id"] == -1
    if name.startswith("whitted"):  # the background lookup has no transcendental in it: exact where every sample missed
        same = (bits(fr.linear[miss]) == bits(g["linear"][miss])).all(axis=-1)
        assert same.mean() > 0.9, f"background pixels bit-identical: {same.mean():.3f}"
    bad = abi.SceneDesc.from_buffer_copy(hs.desc)
    bad.background_texture = 99
    with pytest.raises(lib.RtError):
        ctx.upload(bad)


def test_multi_material_object_vs_reference(ctx, tmp_path, monkeypatch):
    """An OBJ object that brings its own materials (the reference's MultiMtl).  As the reference renders it - every face
    with the first member - against the snapshot's image; with face_materials (each face with the member it names)
    against the CPU oracle, in both renderers and through both pipelines of the path tracer.  A PNG texture on one
    member's diffuse colour (the library's `map_Kd`) travels the same way."""
    from test_host_side import multi_material_scene
    from raytracer_amd import scenes
    g = O.load_golden("whitted_multimtl_96x54_b5")
    hs = multi_material_scene(g, tmp_path)
    seed, mn, mx, w, h, bounces = [int(x) for x in g["params"]]
    prm = abi.default_params(seed=seed, min_spp=mn, max_spp=mx, delta_max=float(g["delta"]), max_bounces=bounces, mode=abi.RT_MODE_WHITTED)
    fr = render_gpu(ctx, hs, prm, w, h)
    assert np.array_equal(fr.object_id, g["object_id"]), "primary-hit ids"
    assert np.array_equal(bits(fr.z), bits(g["z"])), "z-buffer"
    p = O.psnr(fr.rgb, g["rgb"])
    assert p >= PSNR_MIN, f"PSNR {p:.2f} dB vs the reference image"

    # a texture on the second member, and every face with its own member
    (tmp_path / "multi.mtl").write_bytes(bytes(g["mtl"]).replace(b"Ns 90\n", b"Ns 90\nmap_Kd leaf.png\n"))
    (tmp_path / "leaf.png").write_bytes(scenes.png_bytes(scenes.texture_image(), 2, 8))
    hs = lib.HostScene.from_string(bytes(g["xml"]).decode(), str(tmp_path))
    assert hs.warnings() == [] and hs.desc.materials[3].diffuse_texture > 0
    hs.set_face_materials(True)
    cpu = O.render(hs, prm, w, h)
    gpu = render_gpu(ctx, hs, prm, w, h)
    assert np.array_equal(gpu.object_id, cpu.object_id) and np.array_equal(bits(gpu.z), bits(cpu.z))
    p = O.psnr(gpu.rgb, cpu.rgb)
    assert p >= PSNR_MIN, f"Whitted-style shader, per-face members: PSNR {p:.2f} dB vs the oracle"
    assert np.abs(gpu.linear - fr.linear).max(axis=2)[g["object_id"] == 1].mean() > 0.02  # not the first member's picture
    pt = abi.default_params(seed=seed, min_spp=16, max_spp=16)
    cpu = O.render(hs, pt, w, h)
    for pipe in ("wave", "mega"):
        monkeypatch.setenv("RT_PIPELINE", pipe)
        c2 = lib.Context(0)
        try:
            gpu = render_gpu(c2, hs, pt, w, h)
        finally:
            c2.close()
        assert np.array_equal(gpu.object_id, cpu.object_id) and np.array_equal(bits(gpu.z), bits(cpu.z))
        p = O.psnr(gpu.rgb, cpu.rgb)
        assert p >= PSNR_MIN, f"path tracer ({pipe}), per-face members: PSNR {p:.2f} dB vs the oracle"
    # a face that names a member past the materials the node brings is refused
    bad = abi.SceneDesc.from_buffer_copy(hs.desc)
    bad.n_materials -= 1
    with pytest.raises(lib.RtError):
        ctx.upload(bad)


def test_object_hierarchy_changes_nothing(monkeypatch, tmp_path):
    """The Whitted kernel's and the path kernel's scene walks through the hierarchy over the objects' world boxes (scenes of sixteen objects and
    more by default; RT_TLAS_MIN=1 here: every scene) against the walk over the flat node list (RT_TLAS_MIN=0): the
    hierarchy only decides which objects a wave looks at, so every plane is the same bit for bit - on the sphere grid,
    on nested transformations (objects met out of list order, local rays rebuilt through the ancestors), on the
    reference's own scene and on a mesh."""
    from raytracer_amd import scenes
    from test_host_side import multi_material_scene
    cases = [
        (lib.HostScene.from_string(scenes.spheres_scene_xml(seed=5, grid=6, width=160, height=90)), 0, 4),
        (lib.HostScene.from_string(scenes.spheres_scene_xml(seed=7, grid=8, width=128, height=72)), 3, 2),
        (lib.HostScene.from_string(scenes.nested_scene_xml(width=128, height=72)), 3, 2),
        (lib.HostScene.from_string(scenes.cornell_scene_xml(128, 72)), 5, 2),
        (multi_material_scene(O.load_golden("whitted_multimtl_96x54_b5"), tmp_path), 2, 2),
    ]
    # ... and the path kernel on the scenes without a mesh (a scene with one goes through the ray-list kernel)
    cases += [(cases[0][0], -50, 4), (cases[2][0], -50, 4), (cases[3][0], -50, 8)]
    # ... and the ray-list kernel's scene pass: two meshes among 64 spheres (rays that meet both meshes come back to the
    # pass after the first walk), through the path tracer (wavefront pipeline) and the Whitted-style shader
    many = lib.HostScene.from_string(scenes.mesh_and_spheres_scene_xml(grid=8, width=160, height=90))
    mesh = scenes.heightfield(nx=32, ny=16)
    for name in ("mesh_a.obj", "mesh_b.obj"):
        many.set_mesh(name, mesh["v"], mesh["vt"], mesh["f"])
    assert many.desc.n_nodes >= 68
    cases += [(many, -50, 4), (many, 3, 2), (multi_material_scene(O.load_golden("whitted_multimtl_96x54_b5"), tmp_path), -6, 4)]
    frames = {}
    for tlas_min in ("0", "1"):
        monkeypatch.setenv("RT_TLAS_MIN", tlas_min)
        c = lib.Context(0)
        try:
            for k, (hs, bounces, spp) in enumerate(cases):
                prm = abi.default_params(seed=11 + k, min_spp=spp, max_spp=spp, max_bounces=abs(bounces),
                                         mode=abi.RT_MODE_WHITTED if bounces >= 0 else abi.RT_MODE_PATH)
                frames[tlas_min, k] = render_gpu(c, hs, prm, hs.camera.width, hs.camera.height)
        finally:
            c.close()
    for k in range(len(cases)):
        a, b = frames["0", k], frames["1", k]
        assert np.array_equal(a.object_id, b.object_id) and np.array_equal(bits(a.z), bits(b.z)), f"scene {k}: ids / z"
        assert np.array_equal(bits(a.linear), bits(b.linear)), f"scene {k}: radiance"
        assert np.array_equal(a.rgb, b.rgb) and np.array_equal(a.sample_count, b.sample_count)
        assert (a.object_id >= 0).mean() > 0.3


# ---------------------------------------------------------------- samplers and random streams, value for value
def _unit(v):
    """cyVector's Normalize on float32 rows: component / sqrt((x*x + y*y) + z*z)."""
    f = np.float32
    v = np.asarray(v, f)
    l = np.sqrt((v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1]).astype(f) + (v[:, 2] * v[:, 2]).astype(f), dtype=f)
    return (v / l[:, None]).astype(f)


def test_random_streams_on_the_device(ctx):
    """PCG32 as the kernels compute it (64-bit multiply-add in 32-bit pieces, XSH-RR, float conversion with its clamp)
    against the reference's RNG: the streams of the fixture, word for word and float for float (src/rng.h:29-65)."""
    g = O.load_golden("rng")
    seqs = np.array([0, 1, 921599, 0xFFFFFFFFFF], np.uint64)
    seeds = np.array([0, 42, 0x9E3779B97F4A7C15, 7], np.uint64)
    words, floats = ctx.debug_rng(seqs, seeds, 64)
    assert np.array_equal(words, g["rng_u32"])
    assert np.array_equal(bits(floats), bits(g["rng_f32"]))


def test_samplers_on_the_device_vs_reference(ctx):
    """The light and material samplers the shading kernels call, evaluated on the device on the fixture's inputs
    (1024 surface points x the scene's six materials), against the reference's own GenerateSample / GetSampleInfo:
    the lobe chosen, the number of random numbers drawn and the probabilities of GetSampleInfo bit for bit wherever
    only IEEE arithmetic is involved; directions and weights that pass through the device's sinf / cosf / powf to a
    few ulp (those functions differ from glibc's in the last place)."""
    g = O.load_golden("kat_samplers_a")
    s = O.scene_from_golden(g)
    ctx.upload(s.desc)
    inp, n = g["smp_in"], len(g["smp_in"])
    q = np.zeros(n, abi.SAMPLER_QUERY_DTYPE)
    q["p"], q["n"], q["v"], q["dir"] = inp[:, 0:3], _unit(inp[:, 3:6]), -_unit(inp[:, 6:9]), inp[:, 9:12]
    q["front"] = g["smp_front"]
    q["seq"], q["seed"] = np.arange(1, n + 1), 5
    a = ctx.debug_samplers(abi.RT_SAMPLER_LIGHT, q)
    ref = g["smp_light_gen"]  # dir, mult, prob, dist, ok
    np.testing.assert_allclose(a["dir"], ref[:, 0:3], rtol=0, atol=3e-6)
    assert np.array_equal(bits(a["mult"]), bits(ref[:, 3:6]))
    np.testing.assert_allclose(a["prob"], ref[:, 6], rtol=2e-6)
    np.testing.assert_allclose(a["dist"], ref[:, 7], rtol=2e-6)
    exact = (bits(a["dir"]) == bits(ref[:, 0:3])).all(axis=1).mean()
    assert exact > 0.5, f"only {exact:.3f} of the light samples are bit-identical"
    np.testing.assert_allclose(a["pdf"], g["smp_light_pdf"], rtol=2e-6)
    lobes = set()
    for m in range(s.desc.n_materials):
        q["material"], q["seed"] = m, 5 ^ (0x1000 + m)
        a = ctx.debug_samplers(abi.RT_SAMPLER_MATERIAL, q)
        fl, ref = g["smp_mtl_flags"][:, m], g["smp_mtl_gen"][:, m]
        same_branch = (a["ok"] == fl[:, 0]) & (a["next_u32"] == fl[:, 2].astype(np.uint32)) & ((a["lobe"] == fl[:, 1]) | (fl[:, 0] == 0))
        assert same_branch.mean() >= 0.995, f"material {m}: {np.sum(~same_branch)} samples take another branch"
        ok = same_branch & (fl[:, 0] != 0)
        lobes |= set(fl[ok, 1].tolist())
        # (near the axis of a sharp lobe, sin(theta) = sqrt(1 - cos^2) amplifies the last bit of powf a thousandfold)
        np.testing.assert_allclose(a["dir"][ok], ref[ok, 0:3], rtol=0, atol=2e-4)
        assert (np.abs(a["dir"][ok] - ref[ok, 0:3]).max(axis=1) <= 3e-6).mean() >= 0.99
        np.testing.assert_allclose(a["mult"][ok], ref[ok, 3:6], rtol=2e-4, atol=1e-7)
        np.testing.assert_allclose(a["prob"][ok], ref[ok, 6], rtol=2e-4, atol=1e-7)
        np.testing.assert_allclose(a["pdf"], g["smp_mtl_pdf"][:, m], rtol=2e-4, atol=1e-7)
    assert lobes == {1, 2, 4}


def test_photon_samplers_on_the_device_vs_reference(ctx):
    """PointLight::RandomPhoton and the snapshot's scattering sampler on the device against the reference's (double
    sin / cos / pow narrowed once, as the snapshot computes them): draws consumed and branches equal, values to a few ulp."""
    g = O.load_golden("kat_photon_a")
    s = O.scene_from_golden(g)
    ctx.upload(s.desc)
    seed = int(g["seed"])
    n = len(g["ph_emit"])
    q = np.zeros(n, abi.SAMPLER_QUERY_DTYPE)
    q["seq"], q["seed"] = np.arange(n), seed ^ 0x50484F54
    a = ctx.debug_samplers(abi.RT_SAMPLER_PHOTON_EMIT, q)
    got = np.concatenate([a["origin"], a["dir"], a["mult"]], axis=1)
    same = (bits(got) == bits(g["ph_emit"])).all(axis=1)
    assert same.mean() >= 0.99, f"only {same.mean():.3f} of the emitted photons are bit-identical"
    np.testing.assert_allclose(got, g["ph_emit"], rtol=1e-5, atol=1e-5)
    inp, fl, gen = g["ph_smp_in"], g["ph_smp_flags"], g["ph_smp_gen"]
    n = len(inp)
    q = np.zeros(n, abi.SAMPLER_QUERY_DTYPE)
    q["p"], q["n"], q["v"] = inp[:, 0:3], _unit(inp[:, 3:6]), -_unit(inp[:, 6:9])
    q["front"] = g["ph_smp_front"]
    q["seq"] = np.arange(1, n + 1)
    for m in range(gen.shape[1]):
        q["material"], q["seed"] = m, seed ^ (0x2000 + m)
        a = ctx.debug_samplers(abi.RT_SAMPLER_PHOTON_MATERIAL, q)
        same_branch = (a["ok"] == fl[:, m, 0]) & (a["next_u32"] == (fl[:, m, 2].astype(np.int64) & 0xFFFFFFFF).astype(np.uint32)) & ((a["lobe"] == fl[:, m, 1]) | (fl[:, m, 0] == 0))
        assert same_branch.mean() >= 0.995, f"material {m}: {np.sum(~same_branch)} samples take another branch"
        ok = same_branch & (fl[:, m, 0] != 0)
        got = np.concatenate([a["dir"], a["mult"], a["prob"][:, None]], axis=1)
        np.testing.assert_allclose(got[ok], gen[ok, m], rtol=2e-4, atol=2e-4)
        assert (np.abs(got[ok] - gen[ok, m]).max(axis=1) <= 1e-5 * np.maximum(1.0, np.abs(gen[ok, m]).max(axis=1))).mean() >= 0.99
