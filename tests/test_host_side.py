"""Host side of the C ABI (no GPU): the library loads and exports what the header
declares, and its scene.xml / OBJ readers and BVH builder produce the very
arrays the reference's loader produced for the same text (golden fixtures hold
the reference's dump next to the scene text it was given)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import oraclelib as O
from raytracer_amd import abi, lib, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_library_exports_every_declared_symbol():
    L = lib.load()
    header = open(os.path.join(ROOT, "include", "rt_mi355x.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", header))
    assert declared == set(lib.SYMBOLS)
    for name in declared:
        assert hasattr(L, name), name
    assert L.rt_abi_version() == 6


def test_struct_sizes_match_header(tmp_path):
    """The ctypes mirrors in raytracer_amd/abi.py against sizeof() of the C header, compiled here."""
    import subprocess
    pairs = {"rt_node": abi.Node, "rt_bvh_node": abi.BvhNode, "rt_material": abi.Material, "rt_light": abi.Light,
             "rt_hit": abi.Hit, "rt_hit_brief": abi.HitBrief, "rt_render_params": abi.RenderParams, "rt_camera": abi.Camera,
             "rt_frame_out": abi.FrameOut, "rt_counters": abi.Counters, "rt_photon_params": abi.PhotonParams, "rt_progress_info": abi.ProgressInfo,
             "rt_scene_desc": abi.SceneDesc, "rt_mesh": abi.Mesh, "rt_texture": abi.Texture}
    src = "#include <stdio.h>\n#include \"rt_mi355x.h\"\nint main(void){\n"
    src += "".join(f'printf("{n} %zu\\n", sizeof({n}));\n' for n in pairs)
    src += 'printf("rt_photon %zu\\n", sizeof(rt_photon)); return 0; }\n'
    (tmp_path / "s.c").write_text(src)
    subprocess.run(["gcc", "-I", os.path.join(O.ROOT, "include"), str(tmp_path / "s.c"), "-o", str(tmp_path / "s")], check=True)
    out = dict(l.split() for l in subprocess.run([str(tmp_path / "s")], check=True, capture_output=True, text=True).stdout.splitlines())
    for n, t in pairs.items():
        assert C.sizeof(t) == int(out[n]), n
    assert abi.PHOTON_DTYPE.itemsize == int(out["rt_photon"]) == 32
    assert C.sizeof(abi.Node) == 112 and C.sizeof(abi.BvhNode) == 28


def nodes_of(desc):
    n = desc.n_nodes
    tm = np.array([list(desc.nodes[i].tm) for i in range(n)], np.float32)
    itm = np.array([list(desc.nodes[i].itm) for i in range(n)], np.float32)
    meta = np.array([[desc.nodes[i].parent, desc.nodes[i].prim, desc.nodes[i].mesh, desc.nodes[i].material]
                     for i in range(n)], np.int32)
    return tm, itm, meta


def check_scene_against_dump(hs, g):
    d = hs.desc
    tm, itm, meta = nodes_of(d)
    assert d.n_nodes == len(g["scene_node_parent"])
    assert np.array_equal(meta[:, 0], g["scene_node_parent"])
    assert np.array_equal(meta[:, 1], g["scene_node_prim"])
    assert np.array_equal(meta[:, 2], g["scene_node_mesh"])
    assert np.array_equal(meta[:, 3], g["scene_node_material"])
    assert np.array_equal(bits(tm), bits(g["scene_node_tm"])), "node transforms"
    assert np.array_equal(bits(itm), bits(g["scene_node_itm"])), "inverse node transforms"
    mats = np.array([list(m.diffuse) + list(m.specular) + [m.glossiness] + list(m.emission) + list(m.reflection) +
                     list(m.refraction) + list(m.absorption) + [m.ior]
                     for m in (d.materials[i] for i in range(d.n_materials))], np.float32)
    assert np.array_equal(bits(mats), bits(g["scene_materials"]))
    lights = np.array([[l.type] + list(l.intensity) + list(l.position) + list(l.direction) + [l.size, l.attenuation]
                       for l in (d.lights[i] for i in range(d.n_lights))], np.float32)
    assert np.array_equal(bits(lights), bits(g["scene_lights"]))
    c = hs.camera
    cam = np.array(list(c.pos) + list(c.dir) + list(c.up) + [c.fov, c.focaldist, c.dof, c.width, c.height, c.srgb], np.float32)
    assert np.array_equal(bits(cam), bits(g["scene_camera"]))
    assert np.array_equal(bits(list(d.background)), bits(g["scene_background"]))


@pytest.mark.parametrize("name", ["image_nested_96x54_s8", "image_spheres_96x54_s8"])
def test_xml_reader_matches_reference_loader(name):
    g = O.load_golden(name)
    hs = lib.HostScene.from_string(bytes(g["xml"]).decode())
    check_scene_against_dump(hs, g)


def test_generated_cornell_scene_is_the_references_scene():
    """raytracer_amd.scenes.cornell_scene_xml (what bench.py renders as `scene_a`) against the reference loader's dump of
    the scene file the reference ships."""
    g = O.load_golden("scene_a")
    hs = lib.HostScene.from_string(scenes.cornell_scene_xml())
    check_scene_against_dump(hs, g)


def mesh_arrays(m):
    A = np.ctypeslib.as_array
    return dict(v=A(m.v, (m.nv, 3)), vn=A(m.vn, (m.nvn, 3)), vt=A(m.vt, (m.nvt, 3)),
                f=A(m.f, (m.nf, 3)), fn=A(m.fn, (m.nf, 3)), ft=A(m.ft, (m.nf, 3)),
                bvh_nodes=A(C.cast(m.bvh_nodes, abi.u32p), (m.n_bvh_nodes, 7)), bvh_elements=A(m.bvh_elements, (m.nf,)))


def check_mesh(m, g):
    a = mesh_arrays(m)
    for k in ("v", "vn", "vt"):
        assert np.array_equal(bits(a[k]), bits(g["scene_mesh0_" + k])), k
    for k in ("f", "fn", "ft", "bvh_elements"):
        assert np.array_equal(a[k], g["scene_mesh0_" + k]), k
    ref = g["scene_mesh0_bvh_nodes"]
    assert a["bvh_nodes"].shape == ref.shape
    assert np.array_equal(a["bvh_nodes"][1:], ref[1:]), "BVH nodes"
    assert m.bvh_depth == abi.bvh_depth(ref)


def test_mesh_scene_from_obj_file(tmp_path):
    g = O.load_golden("scene_mesh")
    mesh = scenes.heightfield(nx=int(g["mesh_nx"]), ny=int(g["mesh_ny"]))
    scenes.write_obj(tmp_path / "mesh_small.obj", mesh)
    (tmp_path / "scene.xml").write_bytes(bytes(g["xml"]))
    hs = lib.HostScene.from_file(str(tmp_path / "scene.xml"))
    check_scene_against_dump(hs, g)
    check_mesh(hs.desc.meshes[0], g)


def test_mesh_scene_from_arrays():
    g = O.load_golden("scene_mesh")
    mesh = scenes.heightfield(nx=int(g["mesh_nx"]), ny=int(g["mesh_ny"]))
    hs = lib.HostScene.from_string(bytes(g["xml"]).decode())
    assert hs.desc.meshes[0].nf == 0  # file absent: mesh stays empty until set
    hs.set_mesh("mesh_small.obj", mesh["v"], mesh["vt"], mesh["f"])
    check_mesh(hs.desc.meshes[0], g)


def multi_material_scene(g, tmp_path):
    """The fixture's OBJ file, material library and scene text, through our own loader."""
    (tmp_path / "multi.obj").write_bytes(bytes(g["obj"]))
    (tmp_path / "multi.mtl").write_bytes(bytes(g["mtl"]))
    return lib.HostScene.from_string(bytes(g["xml"]).decode(), str(tmp_path))


def test_obj_with_materials_matches_reference_loader(tmp_path):
    """An OBJ file with `usemtl` groups and a material library, on a node without a material attribute: the reference
    regroups the faces by material and gives the node a MultiMtl made of the library's entries (xmlload.cpp:157,278-322).
    Our loader's faces, BVH, sub-material per face and the members' constants against the reference's dump."""
    g = O.load_golden("whitted_multimtl_96x54_b5")
    hs = multi_material_scene(g, tmp_path)
    check_scene_against_dump(hs, g)  # the members follow the scene file's materials; the node names the first
    m = hs.desc.meshes[0]
    check_mesh(m, g)
    assert m.n_sub_materials == 3 == int(g["scene_material_members"][2])
    fm = np.ctypeslib.as_array(m.face_material, (m.nf,))
    assert np.array_equal(fm, g["scene_mesh0_face_material"])
    assert set(fm) == {0, 1, 2} and np.all(np.diff(fm.astype(int)) >= 0)  # grouped: every material's faces are consecutive
    assert hs.desc.face_materials == 0  # as the reference renders: every face with the first member
    assert hs.warnings() == []


def test_png_decoder_matches_reference_decoder():
    """PNG files of every colour type and bit depth, plain and interlaced, all five filter types: the pixels our decoder
    delivers against those of the reference's decoder called as its texture loader calls it (texture.cpp:77-87)."""
    g = O.load_golden("png_cases")
    names = [k[5:] for k in g if k.startswith("file_")]
    assert len(names) == 32
    for n in names:
        got = lib.png_decode(g["file_" + n].tobytes())
        assert got.shape == g["rgb_" + n].shape and np.array_equal(got, g["rgb_" + n]), n
    # the generator in the repository still writes the very files the fixture holds
    cases = scenes.png_cases()
    assert all(cases[n] == g["file_" + n].tobytes() for n in names)
    # damaged files are refused, not decoded
    good = bytearray(g["file_c2_d8_plain"].tobytes())
    for cut in (0, 7, 20, len(good) // 2, len(good) - 13):
        with pytest.raises(lib.RtError):
            lib.png_decode(bytes(good[:cut]))
    bad = bytearray(good)
    bad[len(bad) // 2] ^= 0x40  # inside the image data: chunk checksum
    with pytest.raises(lib.RtError):
        lib.png_decode(bytes(bad))


def test_png_texture_file(tmp_path):
    """A texture named *.png reaches the scene with the pixels of the same image as a PPM file; a name that is neither
    is reported and ignored, as the reference does (texture.cpp:70-95)."""
    g = O.load_golden("image_textured_96x54_s8")
    img = g["image"]
    h, w, _ = img.shape
    (tmp_path / "tex.PNG").write_bytes(scenes.png_bytes(img, 2, 8, interlace=True))
    hs = lib.HostScene.from_string(scenes.textured_scene_xml("tex.PNG", 96, 54), str(tmp_path))
    d = hs.desc
    images = [d.textures[i] for i in range(d.n_textures) if d.textures[i].type == abi.RT_TEX_IMAGE]
    assert len(images) == 2  # the sphere's diffuse colour and the environment
    for tx in images:
        assert (tx.width, tx.height) == (w, h)
        assert np.array_equal(np.ctypeslib.as_array(tx.pixels, shape=(h, w, 3)), img)
    assert hs.warnings() == []
    hs2 = lib.HostScene.from_string(scenes.textured_scene_xml("tex.gif", 96, 54), str(tmp_path))
    assert sum(hs2.desc.textures[i].type == abi.RT_TEX_IMAGE for i in range(hs2.desc.n_textures)) == 0
    assert any("tex.gif" in x for x in hs2.warnings())


def test_bvh_build_entry_point():
    g = O.load_golden("scene_mesh")
    words, elems, depth = lib.bvh_build(g["scene_mesh0_v"], g["scene_mesh0_f"], 4)
    assert np.array_equal(words[1:], g["scene_mesh0_bvh_nodes"][1:])
    assert np.array_equal(elems, g["scene_mesh0_bvh_elements"])
    assert depth == abi.bvh_depth(g["scene_mesh0_bvh_nodes"])
    leaves = words[1:][(words[1:, 6] & 0x80000000) != 0]
    assert (((leaves[:, 6] >> 28) & 7) + 1).max() <= 4


def test_obj_reader_polygons_and_negative_indices(tmp_path):
    (tmp_path / "quad.obj").write_text(
        "# a quad and a pentagon, fan-triangulated\n"
        "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 2 0.5 1\n"
        "vt 0 0 0\nvt 1 0 0\nvt 1 1 0\nvt 0 1 0\nvt 0.5 0.5 0\n"
        "f 1/1 2/2 3/3 4/4\n"
        "f -5/-5 -4/-4 -1/-1 -3/-3 -2/-2\n")
    xml = scenes.mesh_scene_xml("quad.obj", 32, 18)
    (tmp_path / "s.xml").write_text(xml)
    hs = lib.HostScene.from_file(str(tmp_path / "s.xml"))
    a = mesh_arrays(hs.desc.meshes[0])
    assert a["f"].tolist() == [[0, 1, 2], [0, 2, 3], [0, 1, 4], [0, 4, 2], [0, 2, 3]]
    assert np.array_equal(a["ft"], a["f"]) and np.array_equal(a["fn"], a["f"])
    assert a["vn"].shape == (5, 3)


def test_error_reporting():
    with pytest.raises(lib.RtError) as e:
        lib.HostScene.from_file("/nonexistent/scene.xml")
    assert e.value.code == abi.RT_ERR_IO and "Failed to load" in str(e.value)
    for bad, msg in [("<xml><scene></scene></xml>", "camera"), ("<xml><camera/></xml>", "scene"),
                     ("<scene/>", "xml"), ("<xml><scene>", "missing")]:
        with pytest.raises(lib.RtError) as e:
            lib.HostScene.from_string(bad)
        assert e.value.code == abi.RT_ERR_IO and msg in str(e.value)


def test_init_without_gpu_fails_cleanly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    L = lib.load()
    h = C.c_void_p()
    rc = L.rt_init(0, C.byref(h))
    assert rc == abi.RT_ERR_NO_DEVICE and not h.value
    assert b"HIP device" in L.rt_last_error(None)


def test_wide_bvh_records_contain_their_childrens_boxes():
    """The 64-byte four-wide BVH record (device/lbvh.h: wide_record_pack) stores its children's boxes as 8-bit steps
    from the record's corner: a host build of the packing function (tests/cpp/wide_record_check.hip, built by the
    library's Makefile; no GPU call) packs 200 000 random records over twelve orders of magnitude of size, flat and
    point boxes among them, decodes them as the walker does and checks that every decoded box contains the real one and
    is at most two steps wider."""
    exe = os.path.join(ROOT, "raytracer_amd", "wide_record_check")
    if not os.path.exists(exe):
        pytest.skip("raytracer_amd/wide_record_check is not built (make -C raytracer_amd/csrc)")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.startswith("ok "), r.stdout + r.stderr
    assert int(r.stdout.split()[1]) > 400_000


def test_hot_kernels_keep_their_state_in_registers(tmp_path):
    """The frames' kernels are built for four waves per SIMD: at most 128 registers and nothing in scratch.  A change
    that makes the compiler put part of the ray walker's state into private memory costs a quarter of the frame and
    shows in no result, only in the code object's metadata - which is read here (llvm-objdump --offloading,
    llvm-readelf --notes)."""
    llvm = "/opt/rocm/lib/llvm/bin"
    if not os.path.exists(os.path.join(llvm, "llvm-objdump")):
        pytest.skip("no ROCm LLVM tools here")
    import shutil
    so = shutil.copy(os.path.join(ROOT, "raytracer_amd", "librt_mi355x.so"), tmp_path / "lib.so")
    subprocess.run([os.path.join(llvm, "llvm-objdump"), "--offloading", str(so)], check=True, capture_output=True, cwd=tmp_path)
    obj = [p for p in os.listdir(tmp_path) if "gfx950" in p]
    assert len(obj) == 1, os.listdir(tmp_path)
    notes = subprocess.run([os.path.join(llvm, "llvm-readelf"), "--notes", str(tmp_path / obj[0])], check=True, capture_output=True, text=True).stdout
    kernels = {}
    for block in notes.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block).group(1)
        kernels[name] = {k: int(re.search(rf"\.{k}:\s+(\d+)", block).group(1)) for k in ("private_segment_fixed_size", "vgpr_count", "vgpr_spill_count")}
    # the ray-list kernel for scenes whose objects are not nested (one local ray; the instantiations for deeper
    # hierarchies keep up to five and spill a register or two), and the shading kernel
    hot = [n for n in kernels if ("6k_raysILb0ELi1E" in n or "6k_raysILb0ELi0E" in n or "10k_wf_shadeILb" in n)]
    assert len(hot) >= 8, sorted(kernels)[:5]
    for n in hot:
        k = kernels[n]
        assert k["vgpr_count"] <= 128 and k["vgpr_spill_count"] == 0 and k["private_segment_fixed_size"] == 0, (n, k)
