"""The BVH the library builds on the GPU for meshes that come without one (Morton order, one thread per node), behind
the gate the depth / id planes set: it is another tree than the reference's, but it holds the same triangles, tested
with the same arithmetic - so closest-hit distances and object ids are the reference's bit for bit, and images agree to
the tolerance of the other image tests.  Checked against fixtures made by the reference itself, at 1 024, 1 000 000 and
10 000 000 triangles, through the query entry points, the lock-step kernels and the wavefront pipeline."""
import numpy as np
import pytest

import oraclelib as O
from raytracer_amd import abi, lib, scenes

PSNR_MIN = 45.0


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_scene_without_bvh_is_described_without_one():
    g = O.load_golden("scene_mesh")
    mesh = scenes.heightfield(nx=int(g["mesh_nx"]), ny=int(g["mesh_ny"]))
    hs = lib.HostScene.from_string(bytes(g["xml"]).decode())
    hs.set_gpu_bvh(True)
    hs.set_mesh("mesh_small.obj", mesh["v"], mesh["vt"], mesh["f"])
    m = hs.desc.meshes[0]
    assert m.nf == 1024 and not m.bvh_nodes and not m.bvh_elements and m.n_bvh_nodes == 0
    hs.set_gpu_bvh(False)  # the reference's BVH again, as if it had never been dropped
    from test_host_side import check_mesh
    check_mesh(hs.desc.meshes[0], g)


def mesh_scene(golden, name, nx, ny, gpu_bvh):
    mesh = scenes.heightfield(nx, ny)
    hs = lib.HostScene.from_string(bytes(golden["xml"]).decode())
    hs.set_gpu_bvh(gpu_bvh)
    hs.set_mesh(name, mesh["v"], mesh["vt"], mesh["f"])
    return hs


def check_rays(ctx, g):
    hits = ctx.trace(g["kat_rays"])
    ref_hit = g["kat_flags"][:, 0].astype(bool)
    assert np.array_equal(hits["node"], g["kat_id"]), "hit node ids"
    assert np.array_equal(bits(hits["z"]), bits(g["kat_hit"][:, 0])), "z"
    on_obj = ref_hit & (g["kat_id"] >= 0)
    assert np.array_equal(bits(hits["p"][on_obj]), bits(g["kat_hit"][on_obj, 1:4])), "hit point"
    assert np.array_equal(ctx.trace_shadow(g["kat_rays"], g["kat_tmax"]), g["kat_flags"][:, 2]), "shadow rays"
    brief = ctx.trace_brief(g["kat_rays"])
    assert np.array_equal(brief["node"], g["kat_id"]) and np.array_equal(bits(brief["t"]), bits(g["kat_hit"][:, 0]))
    assert np.array_equal(ctx.trace_brief(g["kat_rays"], g["kat_tmax"])["node"], g["kat_flags"][:, 2])


@pytest.mark.gpu
@pytest.mark.parametrize("lds_levels", [None, "3"])
def test_small_mesh_fixtures_with_gpu_bvh(monkeypatch, lds_levels):
    if lds_levels:
        monkeypatch.setenv("RT_LDS_LEVELS", lds_levels)  # the walk's stack overflows into global memory
    g = O.load_golden("kat_trace_mesh")
    hs = mesh_scene(O.load_golden("scene_mesh"), "mesh_small.obj", 32, 16, True)
    ctx = lib.Context(0)
    try:
        ctx.upload(hs.desc)
        assert ctx.counters().bvh_build_ms > 0
        check_rays(ctx, g)
        for name in ("image_mesh_96x54_s8", "whitted_mesh_96x54_b5"):
            g = O.load_golden(name)
            seed, mn, mx, w, h = [int(x) for x in g["params"][:5]]
            kw = dict(seed=seed, min_spp=mn, max_spp=mx, delta_max=float(g["delta"]))
            if name.startswith("whitted"):
                kw.update(mode=abi.RT_MODE_WHITTED, max_bounces=int(g["params"][5]))
            ctx.set_camera(hs.camera, w, h)
            fr = ctx.render(abi.default_params(**kw))
            assert np.array_equal(fr.object_id, g["object_id"]), name
            assert np.array_equal(bits(fr.z), bits(g["z"])), name
            assert O.psnr(fr.rgb, g["rgb"]) >= PSNR_MIN, name
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("pipeline", ["wave", "mega"])
def test_config_c_with_gpu_bvh(monkeypatch, pipeline):
    """The 1M-triangle frame: the reference's rays and its band of the frame, through both forms of the path tracer."""
    monkeypatch.setenv("RT_PIPELINE", pipeline)
    g = O.load_golden("config_c_1m")
    hs = mesh_scene(g, "mesh_1m.obj", int(g["mesh_nx"]), int(g["mesh_ny"]), True)
    ctx = lib.Context(0)
    try:
        ctx.upload(hs.desc)
        ms = ctx.counters().bvh_build_ms
        assert 0 < ms < 200, f"BVH of 1M triangles built in {ms:.1f} ms"
        check_rays(ctx, g)
        seed, mn, mx, w, h, y0, y1 = [int(x) for x in g["params"]]
        ctx.set_camera(hs.camera, w, h)
        fr = ctx.render(abi.default_params(seed=seed, min_spp=mn, max_spp=mx, delta_max=float(g["delta"])))
        assert np.array_equal(fr.object_id[y0:y1], g["object_id"]), "primary-hit ids"
        assert np.array_equal(bits(fr.z[y0:y1]), bits(g["z"])), "z-buffer"
        assert np.array_equal(fr.sample_count[y0:y1], g["sample_count"])
        p = O.psnr(fr.rgb[y0:y1], g["rgb"])
        assert p >= PSNR_MIN, f"PSNR {p:.2f} dB vs the reference's band"
    finally:
        ctx.close()


@pytest.mark.gpu
def test_config_e_rays_with_gpu_bvh():
    """Ten million triangles: built in well under a second, the reference's 4096 aimed rays answered bit for bit."""
    g = O.load_golden("config_e_10m")
    hs = mesh_scene(g, "mesh_10m.obj", int(g["mesh_nx"]), int(g["mesh_ny"]), True)
    ctx = lib.Context(0)
    try:
        ctx.upload(hs.desc)
        ms = ctx.counters().bvh_build_ms
        assert 0 < ms < 200, f"BVH of 10M triangles built in {ms:.1f} ms"
        check_rays(ctx, g)
    finally:
        ctx.close()


@pytest.mark.gpu
def test_objects_of_few_triangles_and_two_meshes(tmp_path):
    """A mesh of two triangles (its root is a leaf) next to a larger one, both built on the GPU, against the walk
    through the reference's BVHs: ids and depths equal.  (The quad is tilted: the reference's slab test never enters a
    box of no thickness, so a mesh that lies in an axis plane is invisible to it.)"""
    quad = "v -3 -3 0\nv 3 -3 0.5\nv 3 3 1\nv -3 3 0.5\nvt 0 0 0\nvt 1 0 0\nvt 1 1 0\nvt 0 1 0\nf 1/1 2/2 3/3\nf 1/1 3/3 4/4\n"
    (tmp_path / "quad.obj").write_text(quad)
    mesh = scenes.heightfield(24, 12, 6.0, 4.0)
    scenes.write_obj(tmp_path / "hills.obj", mesh)
    xml = scenes.mesh_scene_xml("hills.obj", 128, 72).replace(
        '<object type="obj" name="hills.obj"', '<object type="obj" name="quad.obj" material="smooth"><translate z="7"/></object>\n    <object type="obj" name="hills.obj"', 1)
    frames = []
    for gpu in (False, True):
        hs = lib.HostScene.from_string(xml, str(tmp_path))
        assert hs.desc.n_meshes == 2
        hs.set_gpu_bvh(gpu)
        ctx = lib.Context(0)
        try:
            ctx.upload(hs.desc)
            ctx.set_camera(hs.camera, 128, 72)
            frames.append(ctx.render(abi.default_params(seed=3, min_spp=4, max_spp=4)))
        finally:
            ctx.close()
    a, b = frames
    assert {1, 2} <= set(np.unique(a.object_id))
    assert np.array_equal(a.object_id, b.object_id) and np.array_equal(bits(a.z), bits(b.z))
    assert O.psnr(a.rgb, b.rgb) >= PSNR_MIN


@pytest.mark.gpu
@pytest.mark.parametrize("nx,ny,w,h", [(32, 16, 320, 180), (1000, 500, 960, 540)])
def test_gpu_built_tree_is_as_good_as_the_references(nx, ny, w, h):
    """Quality of the tree built on the GPU, measured the way the roofline counts work: box and triangle tests of a
    counted frame (the counting variant walks whatever binary tree the scene has) through the reference's mean-split
    tree and through the GPU's.  The sort keys give their bits to the axis whose cells are still the widest (a plain
    Morton code cuts the flat axis of a height field every third level: 1.3 - 1.6 x the box tests), which brings the
    tree to within a few per cent of the reference's: at most 1.15 x its box tests, no more triangle tests."""
    mesh = scenes.heightfield(nx, ny)
    ctx = lib.Context(0)
    counts = {}
    try:
        for gpu in (False, True):
            hs = lib.HostScene.from_string(scenes.mesh_scene_xml("mesh.obj", w, h))
            hs.set_gpu_bvh(gpu)
            hs.set_mesh("mesh.obj", mesh["v"], mesh["vt"], mesh["f"])
            ctx.upload(hs.desc)
            ctx.set_camera(hs.camera, w, h)
            ctx.set_counting(True)
            ctx.render(abi.default_params(min_spp=2, max_spp=2, seed=7))
            c = ctx.counters()
            ctx.set_counting(False)
            counts[gpu] = (c.box_tests, c.tri_tests, c.closest_rays + c.shadow_rays)
        assert counts[True][2] == counts[False][2] or abs(counts[True][2] - counts[False][2]) < 1e-3 * counts[False][2]
        assert counts[True][0] <= 1.15 * counts[False][0], f"box tests {counts[True][0]} against the reference tree's {counts[False][0]}"
        assert counts[True][1] <= 1.05 * counts[False][1], f"triangle tests {counts[True][1]} against {counts[False][1]}"
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("order", ["a lone triangle first", "shuffled faces", "every quad's halves far apart in the file"])
def test_gpu_bvh_whatever_the_order_of_the_faces(order):
    """The GPU builder pairs consecutive faces that share an edge (the halves of a quad) into units that share a leaf.
    Whether and how faces pair must not matter to any answer: the small mesh with its faces rearranged - a lone triangle
    in front (every later quad starts at an odd place), shuffled (hardly a pair left), the halves of every quad far
    apart (no pairs at all) - must give the rays' answers of the reference's own tree over the same faces, bit for bit,
    through the binary walk and through the four-wide one (ids equal, t within an ulp in all but two rays)."""
    g = O.load_golden("scene_mesh")
    mesh = scenes.heightfield(32, 16)
    f = np.asarray(mesh["f"], np.uint32)
    rng = np.random.default_rng(5)
    if order == "a lone triangle first":
        f = np.concatenate([f[-1:], f[:-1]])
    elif order == "shuffled faces":
        f = f[rng.permutation(len(f))]
    else:
        f = np.concatenate([f[0::2], f[1::2]])
    rays = O.load_golden("kat_trace_mesh")["kat_rays"]
    answers = {}
    for gpu_bvh, wide in ((False, "0"), (True, "0"), (True, "1")):
        hs = lib.HostScene.from_string(bytes(g["xml"]).decode())
        hs.set_gpu_bvh(gpu_bvh)
        hs.set_mesh("mesh_small.obj", mesh["v"], mesh["vt"], f)
        mp = pytest.MonkeyPatch()
        mp.setenv("RT_QUERY_WIDE", wide)
        ctx = lib.Context(0)
        mp.undo()
        try:
            ctx.upload(hs.desc)
            answers[(gpu_bvh, wide)] = ctx.trace_brief(rays)
        finally:
            ctx.close()
    ref, binary, four = answers[(False, "0")], answers[(True, "0")], answers[(True, "1")]
    assert (ref["node"] >= 0).sum() > 100
    assert np.array_equal(binary["node"], ref["node"]) and np.array_equal(bits(binary["t"]), bits(ref["t"]))
    assert np.array_equal(four["node"], ref["node"])
    d = np.abs(bits(four["t"]).astype(np.int64) - bits(ref["t"]).astype(np.int64))
    assert d.max() <= 1 and (d != 0).sum() <= 2
