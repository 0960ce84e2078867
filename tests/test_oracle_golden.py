"""Pin the CPU oracle against outputs of the reference itself.

Every fixture under tests/golden/ was produced by oracle/_ref/ref_harness, i.e.
by the reference's own code compiled from /root/reference/src with
-ffp-contract=off (oracle/make_golden.py).  The oracle must reproduce them bit
for bit: integer streams, hit records, sampler outputs, per-path radiance and
whole seeded images.  CPU only.
"""
import ctypes as C

import numpy as np
import pytest

import oraclelib as O
from raytracer_amd import abi


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def assert_bits_equal(a, b, what):
    a, b = bits(a), bits(b)
    bad = np.argwhere(a != b)
    assert len(bad) == 0, f"{what}: {len(bad)} of {a.size} values differ, first at {bad[0]}"


def test_rng_streams():
    g = O.load_golden("rng")
    L = O.lib()
    seqs = [(0, 0), (1, 42), (921599, 0x9E3779B97F4A7C15), (0xFFFFFFFFFF, 7)]
    r = O.Rng()
    for i, (seq, seed) in enumerate(seqs):
        L.orc_rng_set_sequence(C.byref(r), seq, seed)
        u = [L.orc_rng_u32(C.byref(r)) for _ in range(64)]
        assert np.array_equal(np.array(u, np.uint32), g["rng_u32"][i])
        L.orc_rng_set_sequence(C.byref(r), seq, seed)
        f = np.array([L.orc_rng_float(C.byref(r)) for _ in range(64)], np.float32)
        assert_bits_equal(f, g["rng_f32"][i], "RandomFloat")
    for i, s in enumerate([0, 1, 1001, 2002, 123456789012]):
        L.orc_rng_set_sequence1(C.byref(r), s)
        assert [L.orc_rng_u32(C.byref(r)) for _ in range(8)] == list(g["rng_onearg"][i])
    for i, d in enumerate([1, 5, 1000, 123456789]):
        L.orc_rng_set_sequence(C.byref(r), 3, 9)
        L.orc_rng_advance(C.byref(r), d)
        assert [L.orc_rng_u32(C.byref(r)) for _ in range(4)] == list(g["rng_advance"][i])


def test_halton_tables():
    g = O.load_golden("rng")
    L = O.lib()
    for b, base in enumerate([2, 3, 5, 7]):
        h = np.array([L.orc_halton(i, base) for i in range(16)], np.float32)
        assert_bits_equal(h, g["halton"][b], f"Halton base {base}")


@pytest.mark.parametrize("name", ["scene_a", "image_nested_96x54_s8", "scene_mesh"])
def test_camera_constants(name):
    g = O.load_golden(name)
    s = O.scene_from_golden(g)
    out = np.zeros(12, np.float32)
    O.lib().orc_camera_derive(C.byref(s.camera), out.ctypes.data_as(abi.f32p))
    assert_bits_equal(out, g["scene_camera_derived"], "camera constants")


@pytest.mark.parametrize("name", ["kat_trace_a", "kat_trace_nested", "kat_trace_mesh"])
def test_trace_ray_kat(name):
    g = O.load_golden(name)
    s = O.scene_from_golden(g)
    hits = O.trace(s, g["kat_rays"])
    ref_hit = g["kat_flags"][:, 0].astype(bool)
    assert np.array_equal(hits["node"], g["kat_id"]), "hit node ids"
    assert ref_hit.sum() > len(ref_hit) // 4
    assert_bits_equal(hits["z"], g["kat_hit"][:, 0], "z")
    on_obj = ref_hit & (g["kat_id"] >= 0)  # light hits leave p/N in light-local units, unused downstream
    assert np.array_equal(hits["front"][ref_hit], g["kat_flags"][ref_hit, 1])
    assert_bits_equal(hits["p"][on_obj], g["kat_hit"][on_obj, 1:4], "p")
    assert_bits_equal(hits["n"][on_obj], g["kat_hit"][on_obj, 4:7], "N")
    assert_bits_equal(hits["uvw"][on_obj], g["kat_hit"][on_obj, 7:10], "uvw")
    occ = O.trace_shadow(s, g["kat_rays"], g["kat_tmax"])
    assert np.array_equal(occ, g["kat_flags"][:, 2]), "shadow rays"


def test_sampler_kat():
    g = O.load_golden("kat_samplers_a")
    s = O.scene_from_golden(g)
    L = O.lib()
    n = len(g["smp_in"])
    nm = s.desc.n_materials
    seed = 5
    rng = O.Rng()
    for i in range(n):
        inp = g["smp_in"][i]
        P, N, rd, q = inp[0:3], inp[3:6], inp[6:9], inp[9:12]
        N32 = (N / np.sqrt((N * N).astype(np.float32).sum(dtype=np.float32), dtype=np.float32)).astype(np.float32)
        # normalise exactly as cyVector does: component / sqrt((x*x+y*y)+z*z)
        ln = np.float32(np.sqrt(np.float32(np.float32(N[0] * N[0] + N[1] * N[1]) + N[2] * N[2])))
        N32 = (N / ln).astype(np.float32)
        ld = np.float32(np.sqrt(np.float32(np.float32(rd[0] * rd[0] + rd[1] * rd[1]) + rd[2] * rd[2])))
        V = (-(rd / ld)).astype(np.float32)
        surf = O.Surface()
        surf.p[:] = [float(x) for x in P]; surf.n[:] = [float(x) for x in N32]; surf.v[:] = [float(x) for x in V]
        surf.front = int(g["smp_front"][i])
        smp = O.Sample()
        L.orc_rng_set_sequence(C.byref(rng), i + 1, seed)
        ok = L.orc_light_generate(s.desc.lights, C.byref(surf), C.byref(rng), C.byref(smp))
        got = np.array(list(smp.dir) + list(smp.mult) + [smp.prob, smp.dist, ok], np.float32)
        assert_bits_equal(got, g["smp_light_gen"][i], f"light GenerateSample {i}")
        qd = np.ascontiguousarray(q, np.float32)
        lp = L.orc_light_pdf(s.desc.lights, surf.p, qd.ctypes.data_as(abi.f32p))
        assert_bits_equal(np.float32(lp), g["smp_light_pdf"][i], f"light GetSampleInfo {i}")
        for m in range(nm):
            L.orc_rng_set_sequence(C.byref(rng), i + 1, seed ^ (0x1000 + m))
            ok = L.orc_mtl_generate(C.byref(s.desc.materials[m]), C.byref(surf), C.byref(rng), C.byref(smp))
            fl = g["smp_mtl_flags"][i, m]
            assert ok == fl[0], (i, m)
            assert L.orc_rng_u32(C.byref(rng)) == np.uint32(fl[2]), f"draw count {i},{m}"
            if ok:
                assert smp.lobe == fl[1]
                got = np.array(list(smp.dir) + list(smp.mult) + [smp.prob], np.float32)
                assert_bits_equal(got, g["smp_mtl_gen"][i, m, :7], f"material GenerateSample {i},{m}")
            mp = L.orc_mtl_pdf(C.byref(s.desc.materials[m]), C.byref(surf), qd.ctypes.data_as(abi.f32p))
            assert_bits_equal(np.float32(mp), g["smp_mtl_pdf"][i, m], f"material GetSampleInfo {i},{m}")


def test_path_radiance_kat():
    g = O.load_golden("kat_paths_a")
    s = O.scene_from_golden(g)
    rad = O.trace_paths(s, g["path_rays"], 3, abi.default_params())
    assert (g["path_radiance"].sum(axis=1) > 0).mean() > 0.5
    assert_bits_equal(rad, g["path_radiance"], "tracePath radiance")


@pytest.mark.parametrize("name", ["image_a_160x90_s16", "image_a_96x54_adaptive", "image_nested_96x54_s8",
                                  "image_spheres_96x54_s8", "image_mesh_96x54_s8"])
def test_seeded_images(name):
    g = O.load_golden(name)
    s = O.scene_from_golden(g)
    seed, mn, mx, w, h = [int(x) for x in g["params"]]
    prm = abi.default_params(seed=seed, min_spp=mn, max_spp=mx, delta_max=float(g["delta"]))
    fr = O.render(s, prm, w, h)
    assert np.array_equal(fr.object_id, g["object_id"]), "primary-hit ids"
    assert_bits_equal(fr.z, g["z"], "z-buffer")
    assert np.array_equal(fr.sample_count, g["sample_count"]), "sample counts"
    assert_bits_equal(fr.linear, g["linear"], "linear radiance")
    assert np.array_equal(fr.rgb, g["rgb"]), "8-bit image"
    if "adaptive" in name:
        assert len(np.unique(g["sample_count"])) > 3


WHITTED = ["whitted_a_96x54_b0", "whitted_a_96x54_b5", "whitted_a_64x36_adaptive", "whitted_spheres_96x54_b0",
           "whitted_mesh_96x54_b5", "whitted_lights_96x54_b5"]  # the last: ambient + directional + spherical light


@pytest.mark.parametrize("name", WHITTED)
def test_whitted_images(name):
    """RT_MODE_WHITTED against the project-12 snapshot of the reference (its Whitted-style shader with the
    photon maps switched off): direct light with adaptive soft shadows (b0), and the full recursion with
    Monte-Carlo indirect light, glossy reflection and refraction (b5) - bit for bit."""
    g = O.load_golden(name)
    s = O.scene_from_golden(g)
    seed, mn, mx, w, h, bounces = [int(x) for x in g["params"]]
    prm = abi.default_params(seed=seed, min_spp=mn, max_spp=mx, delta_max=float(g["delta"]), max_bounces=bounces,
                             mode=abi.RT_MODE_WHITTED)
    fr = O.render(s, prm, w, h)
    assert np.array_equal(fr.object_id, g["object_id"]), "primary-hit ids"
    assert_bits_equal(fr.z, g["z"], "z-buffer")
    assert np.array_equal(fr.sample_count, g["sample_count"]), "sample counts"
    assert_bits_equal(fr.linear, g["linear"], "linear radiance")
    assert np.array_equal(fr.rgb, g["rgb"]), "8-bit image"
    if "adaptive" in name:
        assert len(np.unique(g["sample_count"])) > 3
    if name.endswith("b5"):
        # the recursion changes the picture: mirrors / glass are not black any more
        b0 = abi.default_params(seed=seed, min_spp=mn, max_spp=mx, delta_max=float(g["delta"]), max_bounces=0,
                                mode=abi.RT_MODE_WHITTED)
        assert np.abs(O.render(s, b0, w, h).linear - fr.linear).max() > 0.05


def test_every_kind_of_light_shows_in_the_lights_golden():
    """The golden with three lights would pin nothing if two of them added nothing: without the ambient light, and
    without the directional one, the oracle's image of the same scene differs from the reference's clearly (the
    light loop of the snapshot's shader, project12.zip src/materials.cpp:24-47, src/lights.h:28-56)."""
    g = O.load_golden("whitted_lights_96x54_b5")
    seed, mn, mx, w, h, bounces = [int(x) for x in g["params"]]
    prm = abi.default_params(seed=seed, min_spp=mn, max_spp=mx, delta_max=float(g["delta"]), max_bounces=bounces, mode=abi.RT_MODE_WHITTED)
    kinds = g["scene_lights"][:, 0].astype(int).tolist()
    assert kinds == [abi.RT_LIGHT_POINT, abi.RT_LIGHT_AMBIENT, abi.RT_LIGHT_DIRECT]
    for drop in (1, 2):
        dark = dict(g)
        dark["scene_lights"] = g["scene_lights"].copy()
        dark["scene_lights"][drop, 1:4] = 0.0  # the light's intensity
        s = O.scene_from_golden(dark)
        diff = np.abs(O.render(s, prm, w, h).linear - g["linear"])
        assert (diff.max(axis=-1) > 0.02).mean() > 0.25, f"light {drop} changes too little of the image"


def photon_words(a):
    """32-byte photon records as [n, 8] words, the form the fixtures hold them in."""
    return np.ascontiguousarray(a).view(np.uint32).reshape(-1, 8)


def pm_params(g):
    seed, ng, nc, seg, flags = [int(x) for x in g["params"][-5:]] if len(g["params"]) > 5 else [int(x) for x in g["params"]]
    return abi.PhotonParams(flags=flags, n_global=ng, n_caustic=nc, max_segments=seg, seed=seed)


@pytest.mark.parametrize("name", ["photons_a_indirect_caustics", "photons_a_direct"])
def test_photon_maps(name):
    """Photon emission (one PCG stream per path), the balanced kd-tree and EstimateIrradiance<128> against the
    project-12 snapshot's PhotonMap / RandomPhoton / GenerateSample - bit for bit."""
    g = O.load_golden(name)
    s = O.scene_from_golden(g)
    seed, ng, nc, seg, flags = [int(x) for x in g["params"]]
    pp = abi.PhotonParams(flags=flags, n_global=ng, n_caustic=nc, max_segments=seg, seed=seed)
    for which, key in ((abi.RT_MAP_GLOBAL, "global"), (abi.RT_MAP_CAUSTICS, "caustic")):
        raw, paths = O.photons_emit(s, pp, which)
        ref_raw = g[f"pm_{key}_raw"]
        assert len(raw) == len(ref_raw)
        if len(raw) == 0:
            continue
        assert paths == int(g["pm_paths"][which]), "photon paths needed to fill the map"
        assert np.array_equal(photon_words(raw), ref_raw), f"{key}: emitted photons"
        tree = O.photon_build(raw)
        assert np.array_equal(photon_words(tree), g[f"pm_{key}_tree"]), f"{key}: balanced kd-tree"
        # chunked emission gives the same photons as one serial run (what the multi-GPU split relies on)
        parts = [O.photons_emit(s, pp, which, b, 1000, capacity=len(raw))[0] for b in range(0, paths, 1000)]
        assert np.array_equal(photon_words(np.concatenate(parts))[:len(raw)], ref_raw)
        col = 0 if which == abi.RT_MAP_GLOBAL else 3
        e = O.photon_irradiance(tree, g["pm_query"][:, :3], g["pm_query"][:, 3:], float(g["radius"]))
        assert_bits_equal(e, g["pm_irradiance"][:, col:col + 3], f"{key}: irradiance estimates")
        assert (e > 0).any()


@pytest.mark.parametrize("name", ["whitted_pm_a_64x36_b5", "whitted_pm_a_64x36_b0", "whitted_pd_a_64x36_b5"])
def test_whitted_photon_images(name):
    """The Whitted-style shader reading the photon maps (indirect + caustics as the snapshot ships, and direct
    light from the map) against images of the snapshot - bit for bit."""
    g = O.load_golden(name)
    s = O.scene_from_golden(g)
    seed, mn, mx, w, h, bounces, ng, nc, seg, flags = [int(x) for x in g["params"]]
    pp = abi.PhotonParams(flags=flags, n_global=ng, n_caustic=nc, max_segments=seg, seed=seed)
    maps = O.fill_photon_maps(s, pp)
    assert np.array_equal(photon_words(maps[0]), g["pm_global_tree"])
    if flags & abi.RT_PM_CAUSTICS:
        assert np.array_equal(photon_words(maps[1]), g["pm_caustic_tree"])
    prm = abi.default_params(seed=seed, min_spp=mn, max_spp=mx, delta_max=float(g["delta"]), max_bounces=bounces,
                             mode=abi.RT_MODE_WHITTED, photon_flags=flags, gather_radius=3.0)
    fr = O.render(s, prm, w, h)
    assert np.array_equal(fr.object_id, g["object_id"]), "primary-hit ids"
    assert_bits_equal(fr.z, g["z"], "z-buffer")
    assert_bits_equal(fr.linear, g["linear"], "linear radiance")
    assert np.array_equal(fr.rgb, g["rgb"]), "8-bit image"
    assert fr.stats.gathers > 0


def test_photon_emission_and_scatter_samplers():
    """PointLight::RandomPhoton and the snapshot's MtlBlinn::GenerateSample (the photon scattering sampler),
    value for value, including the number of random draws each call consumes."""
    g = O.load_golden("kat_photon_a")
    s = O.scene_from_golden(g)
    seed = int(g["seed"])
    L = O.lib()
    rng = O.Rng()
    for i, ref in enumerate(g["ph_emit"]):
        L.orc_rng_set_sequence(C.byref(rng), i, seed ^ 0x50484F54)
        r, c = np.zeros(6, np.float32), np.zeros(3, np.float32)
        L.orc_light_random_photon(C.byref(s.desc.lights[0]), C.byref(rng), r.ctypes.data_as(abi.f32p), c.ctypes.data_as(abi.f32p))
        assert_bits_equal(np.concatenate([r, c]), ref, f"RandomPhoton {i}")
    inp, front, gen, fl = g["ph_smp_in"], g["ph_smp_front"], g["ph_smp_gen"], g["ph_smp_flags"]
    f32 = np.float32

    def unit(v):  # Vec3f::Normalize: (x*x + y*y + z*z), sqrt, three divisions
        return (v / np.sqrt(f32(f32(v[0] * v[0]) + f32(v[1] * v[1])) + f32(v[2] * v[2]), dtype=f32)).astype(f32)

    lobes = set()
    for i in range(len(inp)):
        su = O.Surface()
        su.p[:] = inp[i, :3].tolist()
        su.n[:] = unit(inp[i, 3:6]).tolist()
        su.v[:] = (-unit(inp[i, 6:9])).tolist()
        su.front = int(front[i])
        for m in range(gen.shape[1]):
            L.orc_rng_set_sequence(C.byref(rng), i + 1, seed ^ (0x2000 + m))
            o = O.Sample()
            ok = L.orc_p12_mtl_generate(C.byref(s.desc.materials[m]), C.byref(su), C.byref(rng), C.byref(o))
            assert ok == fl[i, m, 0], (i, m)
            assert L.orc_rng_u32(C.byref(rng)) == int(fl[i, m, 2]) & 0xFFFFFFFF, f"draws consumed {i} {m}"
            if ok:
                assert o.lobe == fl[i, m, 1]
                lobes.add(o.lobe)
                assert_bits_equal(np.array(list(o.dir) + list(o.mult) + [o.prob], f32), gen[i, m], f"GenerateSample {i} {m}")
    assert lobes == {1, 2, 4}


def textured_scene(g, tmp_path):
    """The textured fixture's scene through our own loader: its XML text and image travel in the fixture."""
    from raytracer_amd import lib, scenes
    scenes.write_ppm(str(tmp_path / "tex.ppm"), g["image"])
    return lib.HostScene.from_string(bytes(g["xml"]).decode(), str(tmp_path))


@pytest.mark.parametrize("name", ["image_textured_96x54_s8", "whitted_textured_96x54_b5"])
def test_textures(name, tmp_path):
    """Checkerboard and image textures with texture transformations on a material's diffuse colour, the background and
    the environment: our loader's texture maps against the reference's, texture lookups and EvalEnvironment value for
    value, and the rendered images (path tracer at HEAD: background only; the snapshot's shader: all three) bit for bit."""
    g = O.load_golden(name)
    hs = textured_scene(g, tmp_path)
    d = hs.desc
    owners = [(0, 0, d.background_texture), (1, 0, d.environment_texture)] + \
             [(2, k, d.materials[k].diffuse_texture) for k in range(d.n_materials)]
    owners = [(kind, idx, t) for kind, idx, t in owners if t > 0]
    assert [(k, i) for k, i, _ in owners] == [tuple(x) for x in g["scene_tex_owner"]]
    L = O.lib()
    for row, (_, _, t) in enumerate(owners):
        tx = d.textures[t - 1]
        assert tx.type == g["scene_tex_type"][row]
        assert_bits_equal(np.array(tx.itm[:], np.float32), g["scene_tex_itm"][row], "texture transformation")
        if tx.type == abi.RT_TEX_CHECKER:
            assert_bits_equal(np.array(list(tx.color1) + list(tx.color2), np.float32), g["scene_tex_colors"][row], "checker colours")
        else:
            assert (tx.width, tx.height) == tuple(g["scene_tex_size"][row])
            px = np.ctypeslib.as_array(tx.pixels, shape=(tx.height, tx.width, 3))
            assert np.array_equal(px, g[f"scene_tex{row}_pixels"])
        got = np.zeros((256, 3), np.float32)
        for q, uvw in enumerate(g["scene_tex_kat_uvw"][row]):
            L.orc_texture_eval(C.byref(d), t, uvw.copy().ctypes.data_as(abi.f32p), got[q].ctypes.data_as(abi.f32p))
        assert_bits_equal(got, g["scene_tex_kat_rgb"][row], f"texture {row} lookups")
    env = np.zeros((256, 3), np.float32)
    for q, dirv in enumerate(g["scene_env_kat_dir"]):
        L.orc_environment_eval(C.byref(d), dirv.copy().ctypes.data_as(abi.f32p), env[q].ctypes.data_as(abi.f32p))
    assert_bits_equal(env, g["scene_env_kat_rgb"], "EvalEnvironment")
    seed, mn, mx, w, h = [int(x) for x in g["params"][:5]]
    kw = dict(seed=seed, min_spp=mn, max_spp=mx, delta_max=float(g["delta"]))
    if name.startswith("whitted"):
        kw.update(mode=abi.RT_MODE_WHITTED, max_bounces=int(g["params"][5]))
    fr = O.render(hs, abi.default_params(**kw), w, h)
    assert np.array_equal(fr.object_id, g["object_id"]), "primary-hit ids"
    assert_bits_equal(fr.z, g["z"], "z-buffer")
    assert_bits_equal(fr.linear, g["linear"], "linear radiance")
    assert np.array_equal(fr.rgb, g["rgb"]), "8-bit image"
    assert (g["object_id"] == -1).mean() > 0.05  # the background is in the picture


def test_multi_material_object(tmp_path):
    """An OBJ object with its own materials (a MultiMtl in the reference) rendered by the snapshot's shader: every face
    is shaded with the first member, because no intersector sets HitInfo::mtlID - bit for bit.  With face_materials the
    faces take the members they name: a different picture on the object, the same everywhere else."""
    from test_host_side import multi_material_scene
    g = O.load_golden("whitted_multimtl_96x54_b5")
    hs = multi_material_scene(g, tmp_path)
    seed, mn, mx, w, h, bounces = [int(x) for x in g["params"]]
    prm = abi.default_params(seed=seed, min_spp=mn, max_spp=mx, delta_max=float(g["delta"]), max_bounces=bounces, mode=abi.RT_MODE_WHITTED)
    fr = O.render(hs, prm, w, h)
    assert np.array_equal(fr.object_id, g["object_id"]), "primary-hit ids"
    assert_bits_equal(fr.z, g["z"], "z-buffer")
    assert_bits_equal(fr.linear, g["linear"], "linear radiance")
    assert np.array_equal(fr.rgb, g["rgb"]), "8-bit image"
    on_object = g["object_id"] == 1
    assert on_object.mean() > 0.05
    hs.set_face_materials(True)
    per_face = O.render(hs, prm, w, h)
    assert np.array_equal(per_face.object_id, g["object_id"])
    diff = np.abs(per_face.linear - fr.linear).max(axis=2)
    assert (diff[on_object] > 0.02).mean() > 0.3  # two thirds of the faces name another member
    for mode in (abi.RT_MODE_PATH,):  # the path tracer picks members the same way
        p2 = abi.default_params(seed=seed, min_spp=8, max_spp=8, mode=mode)
        a = O.render(hs, p2, w, h)
        hs.set_face_materials(False)
        b = O.render(hs, p2, w, h)
        assert (np.abs(a.linear - b.linear).max(axis=2)[on_object] > 0.02).mean() > 0.3
