#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the real reference (oracle/_ref/ref_harness).

TEST INFRASTRUCTURE ONLY.  Runs only in the build container, where
/root/reference exists and `make -C oracle ref` has built the harness from the
reference's own sources.  The fixtures hold inputs (flattened scenes, rays,
seeds, generated scene text) and the reference's outputs for them - no
reference source text.

    python oracle/make_golden.py            # regenerate everything
    python oracle/make_golden.py --whitted  # only the project-12 (Whitted-style shader) fixtures
    python oracle/make_golden.py --photons  # only the photon-map fixtures
    python oracle/make_golden.py --config-c       # only the full-size 1M-triangle band and ray fixture
    python oracle/make_golden.py --config-e       # only the full-size 10M-triangle band and ray fixture (minutes)
    python oracle/make_golden.py --photons-large  # only the tall photon map
    python oracle/make_golden.py --png            # only the PNG decoding cases
    python oracle/make_golden.py --multimtl       # only the OBJ-with-materials scene
    python oracle/make_golden.py --lights         # only the scene with ambient + directional + spherical light
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from raytracer_amd import scenes  # noqa: E402

HARNESS = os.path.join(HERE, "_ref", "ref_harness")
HARNESS_P12 = os.path.join(HERE, "_ref", "ref_harness_p12")  # the project-12 snapshot (Whitted-style shader)
REF_SCENE = "/root/reference/assets/scene.xml"
GOLD = os.path.join(ROOT, "tests", "golden")


def run(scene, cmd, *args, cwd=None, harness=None):
    """Run one harness command, return {name: array}."""
    with tempfile.TemporaryDirectory() as out:
        subprocess.run([harness or HARNESS, scene, out, cmd] + [str(a) for a in args], check=True, cwd=cwd,
                       stdout=subprocess.DEVNULL)
        res = {}
        with open(os.path.join(out, "manifest.txt")) as fp:
            for line in fp:
                name, dt, *shape = line.split()
                a = np.fromfile(os.path.join(out, name + ".bin"), dtype=np.dtype(dt))
                res[name] = a.reshape([int(s) for s in shape])
        return res


def save(name, **arrays):
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("%-32s %7.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


def scene_bundle(scene, cwd=None, harness=None):
    d = run(scene, "scene", cwd=cwd, harness=harness)
    return {"scene_" + k: v for k, v in d.items()}


def whitted():
    """Images of the project-12 snapshot's Whitted-style shader (no photon maps).
    params = seed, min spp, max spp, width, height, max_bounces."""
    if not os.path.exists(HARNESS_P12):
        sys.exit("build the snapshot harness first: make -C oracle ref12")
    H = HARNESS_P12

    def image(name, scene, bundle, seed, mn, mx, delta, w, h, bounces, cwd=None, **extra):
        img = run(scene, "render_whitted", seed, mn, mx, delta, w, h, bounces, cwd=cwd, harness=H)
        save(name, **bundle, **extra, params=np.array([seed, mn, mx, w, h, bounces], np.int64), delta=np.float32(delta), **img)

    sa = scene_bundle(REF_SCENE, harness=H)
    image("whitted_a_96x54_b0", REF_SCENE, sa, 7, 4, 4, 0.01, 96, 54, 0)
    image("whitted_a_96x54_b5", REF_SCENE, sa, 7, 4, 4, 0.01, 96, 54, 5)
    image("whitted_a_64x36_adaptive", REF_SCENE, sa, 3, 4, 16, 0.05, 64, 36, 5)
    with tempfile.TemporaryDirectory() as tmp:
        xml = scenes.spheres_scene_xml(seed=1001, grid=4, width=96, height=54)
        p = os.path.join(tmp, "spheres.xml")
        open(p, "w").write(xml)
        image("whitted_spheres_96x54_b0", p, scene_bundle(p, harness=H), 7, 8, 8, 0.01, 96, 54, 0,
              xml=np.frombuffer(xml.encode(), np.uint8))
        mesh = scenes.heightfield(nx=32, ny=16)
        scenes.write_obj(os.path.join(tmp, "mesh_small.obj"), mesh)
        xml = scenes.mesh_scene_xml("mesh_small.obj", width=96, height=54)
        p = os.path.join(tmp, "mesh.xml")
        open(p, "w").write(xml)
        image("whitted_mesh_96x54_b5", p, scene_bundle(p, cwd=tmp, harness=H), 7, 4, 4, 0.01, 96, 54, 5, cwd=tmp)
        lights(tmp, image)


def lights(tmp, image):
    """Ambient + directional + spherical light in one scene, through the snapshot's shader (its light loop,
    AmbientLight / DirectLight::Illuminate) with reflection and refraction on."""
    xml = scenes.lights_scene_xml(width=96, height=54)
    p = os.path.join(tmp, "lights.xml")
    open(p, "w").write(xml)
    image("whitted_lights_96x54_b5", p, scene_bundle(p, harness=HARNESS_P12), 7, 4, 4, 0.01, 96, 54, 5,
          xml=np.frombuffer(xml.encode(), np.uint8))


def photons():
    """Photon maps of the project-12 snapshot: emission, balanced kd-tree, irradiance estimates, and images of
    its Whitted-style shader reading the maps.  flags = RT_PM_* bits the harness was compiled with."""
    HPM = os.path.join(HERE, "_ref", "ref_harness_p12pm")  # indirect light + caustics from photon maps (as shipped)
    HPD = os.path.join(HERE, "_ref", "ref_harness_p12pd")  # direct light from the photon map
    for h in (HPM, HPD):
        if not os.path.exists(h):
            sys.exit("build the snapshot harnesses first: make -C oracle ref12")
    sa = scene_bundle(REF_SCENE, harness=HPM)
    # params = seed, n_global, n_caustic, max_segments, flags
    r = run(REF_SCENE, "photons", 7, 4000, 1500, 64, 384, 3, harness=HPM)
    save("photons_a_indirect_caustics", **sa, params=np.array([7, 4000, 1500, 64, 2 | 4], np.int64), radius=np.float32(3), **r)
    r = run(REF_SCENE, "photons", 11, 3000, 0, 64, 256, 3, harness=HPD)
    save("photons_a_direct", **sa, params=np.array([11, 3000, 0, 64, 1], np.int64), radius=np.float32(3), **r)
    save("kat_photon_a", **sa, seed=np.int64(5), **run(REF_SCENE, "kat_photon", 512, 5, harness=HPM))
    # images; params = seed, min spp, max spp, width, height, max_bounces, n_global, n_caustic, max_segments, flags
    for name, H, flags, args in (("whitted_pm_a_64x36_b5", HPM, 2 | 4, (7, 4, 4, 0.01, 64, 36, 5, 4000, 1500, 64)),
                                 ("whitted_pm_a_64x36_b0", HPM, 2 | 4, (5, 2, 2, 0.01, 64, 36, 0, 2500, 800, 64)),
                                 ("whitted_pd_a_64x36_b5", HPD, 1, (9, 2, 2, 0.01, 64, 36, 5, 3000, 0, 64))):
        img = run(REF_SCENE, "render_whitted", *args, harness=H)
        seed, mn, mx, delta, w, h, bounces, ng, nc, seg = args
        save(name, **sa, params=np.array([seed, mn, mx, w, h, bounces, ng, nc, seg, flags], np.int64), delta=np.float32(delta), **img)


def textures():
    """A scene with checkerboard and image textures on a material, the background and the environment: the reference's
    scene dump with its texture maps and known-answer lookups, a path-traced image (HEAD) and a Whitted-style one (snapshot)."""
    with tempfile.TemporaryDirectory() as tmp:
        img = scenes.texture_image()
        scenes.write_ppm(os.path.join(tmp, "tex.ppm"), img)
        xml = scenes.textured_scene_xml("tex.ppm", 96, 54)
        p = os.path.join(tmp, "textured.xml")
        open(p, "w").write(xml)
        common = dict(xml=np.frombuffer(xml.encode(), np.uint8), image=img)
        st = scene_bundle(p, cwd=tmp)
        out = run(p, "render", 7, 8, 8, 0.01, 96, 54, cwd=tmp)
        save("image_textured_96x54_s8", **common, **st, params=np.array([7, 8, 8, 96, 54], np.int64), delta=np.float32(0.01), **out)
        H = HARNESS_P12
        st12 = scene_bundle(p, cwd=tmp, harness=H)
        out = run(p, "render_whitted", 7, 4, 4, 0.01, 96, 54, 5, cwd=tmp, harness=H)
        save("whitted_textured_96x54_b5", **common, **st12, params=np.array([7, 4, 4, 96, 54, 5], np.int64), delta=np.float32(0.01), **out)


def config_c():
    """The 1M-triangle configuration at its full size (1000 x 500 quads, BVH depth 24, 1920x1080): a 16-row band of
    the frame through every object of the scene, and rays aimed at the mesh.  The mesh arrays are far too large
    for a fixture: the fixture holds SHA-256 digests of the reference's arrays (vertices, computed normals, faces,
    BVH nodes and elements), which the library's own OBJ reader / normal computation / BVH builder must reproduce
    from the generated mesh."""
    import hashlib
    with tempfile.TemporaryDirectory() as tmp:
        mesh = scenes.heightfield(1000, 500)
        scenes.write_obj(os.path.join(tmp, "mesh_1m.obj"), mesh)
        xml = scenes.mesh_scene_xml("mesh_1m.obj", 1920, 1080)
        p = os.path.join(tmp, "scene.xml")
        open(p, "w").write(xml)
        sc = scene_bundle(p, cwd=tmp)
        digests = {}
        small = {}
        for k, v in sc.items():
            if k.startswith("scene_mesh0_"):
                if k.endswith("bvh_nodes"):
                    v = v[1:]  # cyBVH leaves node 0 unused (and unwritten)
                digests[k.replace("scene_", "sha256_")] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(v).tobytes()).digest(), np.uint8)
                small["count_" + k[len("scene_"):]] = np.int64(len(v))
            else:
                small[k] = v
        common = dict(xml=np.frombuffer(xml.encode(), np.uint8), mesh_nx=np.int32(1000), mesh_ny=np.int32(500), **small, **digests)
        y0, y1 = 416, 432
        band = run(p, "render", 7, 4, 4, 0.01, 1920, 1080, y0, y1, cwd=tmp)
        kat = run(p, "kat_trace", 8192, 31, 40, 1, cwd=tmp)
        save("config_c_1m", **common, params=np.array([7, 4, 4, 1920, 1080, y0, y1], np.int64), delta=np.float32(0.01), **band, **kat)


def config_e():
    """The 10M-triangle configuration at its full size (5000 x 1000 quads, BVH depth 29, 3840x2160, 16..256 samples with
    the variance stop): digests of the reference's mesh arrays and BVH, a 16-row band of the frame through every object
    and rays aimed at the mesh, as config_c()."""
    import hashlib
    with tempfile.TemporaryDirectory() as tmp:
        mesh = scenes.heightfield(5000, 1000)
        scenes.write_obj(os.path.join(tmp, "mesh_10m.obj"), mesh)
        xml = scenes.mesh_scene_xml("mesh_10m.obj", 3840, 2160)
        p = os.path.join(tmp, "scene.xml")
        open(p, "w").write(xml)
        sc = scene_bundle(p, cwd=tmp)
        digests, small = {}, {}
        for k, v in sc.items():
            if k.startswith("scene_mesh0_"):
                if k.endswith("bvh_nodes"):
                    v = v[1:]
                digests[k.replace("scene_", "sha256_")] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(v).tobytes()).digest(), np.uint8)
                small["count_" + k[len("scene_"):]] = np.int64(len(v))
            else:
                small[k] = v
        del sc
        common = dict(xml=np.frombuffer(xml.encode(), np.uint8), mesh_nx=np.int32(5000), mesh_ny=np.int32(1000), **small, **digests)
        y0, y1 = 832, 848
        band = run(p, "render", 7, 16, 256, 0.01, 3840, 2160, y0, y1, cwd=tmp)
        kat = run(p, "kat_trace", 4096, 37, 40, 1, cwd=tmp)
        save("config_e_10m", **common, params=np.array([7, 16, 256, 3840, 2160, y0, y1], np.int64), delta=np.float32(0.01), **band, **kat)


def photons_large():
    """A photon map tall enough (>= 32768 photons: 16 levels) that the kd-tree walk of a lookup leaves the levels the
    GPU keeps in fast memory; otherwise as photons()."""
    HPM = os.path.join(HERE, "_ref", "ref_harness_p12pm")
    sa = scene_bundle(REF_SCENE, harness=HPM)
    r = run(REF_SCENE, "photons", 13, 36000, 17000, 64, 512, 3, harness=HPM)
    save("photons_a_large", **sa, params=np.array([13, 36000, 17000, 64, 2 | 4], np.int64), radius=np.float32(3), **r)


def png():
    """PNG texture files of every colour type and bit depth, plain and interlaced, decoded by the reference's own
    decoder as its texture loader calls it: the file bytes and the 8-bit RGB pixels it delivers."""
    cases = scenes.png_cases()
    with tempfile.TemporaryDirectory() as tmp:
        names = sorted(cases)
        for n in names:
            open(os.path.join(tmp, n + ".png"), "wb").write(cases[n])
        r = run("-", "png", *[os.path.join(tmp, n + ".png") for n in names])
        arrays = {}
        for i, n in enumerate(names):
            assert r[f"png{i}"].size, f"the reference's decoder refuses {n}"
            arrays["file_" + n] = np.frombuffer(cases[n], np.uint8)
            arrays["rgb_" + n] = r[f"png{i}"]
        save("png_cases", **arrays)


def multi_material():
    """An OBJ object that brings its own materials (usemtl groups, a material library; no material attribute on the
    node): the scene as the reference loads it - the members of the MultiMtl, the regrouped faces and their
    sub-material indices - and an image of the snapshot's Whitted-style shader, which shades through MultiMtl::Shade.
    (The path tracer at HEAD casts the node's material to a Blinn material unchecked, src/main.cpp:212: undefined for
    a MultiMtl, so there is no path-traced reference image.)"""
    with tempfile.TemporaryDirectory() as tmp:
        obj, mtl = scenes.multi_material_mesh()
        open(os.path.join(tmp, "multi.obj"), "w").write(obj)
        open(os.path.join(tmp, "multi.mtl"), "w").write(mtl)
        xml = scenes.multi_material_scene_xml("multi.obj", 96, 54)
        p = os.path.join(tmp, "scene.xml")
        open(p, "w").write(xml)
        H = HARNESS_P12
        common = dict(xml=np.frombuffer(xml.encode(), np.uint8), obj=np.frombuffer(obj.encode(), np.uint8), mtl=np.frombuffer(mtl.encode(), np.uint8))
        sc = scene_bundle(p, cwd=tmp, harness=H)
        out = run(p, "render_whitted", 7, 4, 4, 0.01, 96, 54, 5, cwd=tmp, harness=H)
        save("whitted_multimtl_96x54_b5", **common, **sc, params=np.array([7, 4, 4, 96, 54, 5], np.int64), delta=np.float32(0.01), **out)


def main():
    os.makedirs(GOLD, exist_ok=True)
    if "--config-e" in sys.argv:
        return config_e()
    if "--png" in sys.argv:
        return png()
    if "--multimtl" in sys.argv:
        return multi_material()
    if "--config-c" in sys.argv:
        return config_c()
    if "--photons-large" in sys.argv:
        return photons_large()
    if "--textures" in sys.argv:
        return textures()
    if "--lights" in sys.argv:
        def image(name, scene, bundle, seed, mn, mx, delta, w, h, bounces, cwd=None, **extra):
            img = run(scene, "render_whitted", seed, mn, mx, delta, w, h, bounces, cwd=cwd, harness=HARNESS_P12)
            save(name, **bundle, **extra, params=np.array([seed, mn, mx, w, h, bounces], np.int64), delta=np.float32(delta), **img)
        with tempfile.TemporaryDirectory() as tmp:
            return lights(tmp, image)
    if "--whitted" in sys.argv:  # only the snapshot's fixtures
        return whitted()
    if "--photons" in sys.argv:
        return photons()
    if not os.path.exists(HARNESS):
        sys.exit("build the harness first: make -C oracle ref")

    save("rng", **run(REF_SCENE, "kat_rng"))

    # --- scene A: the reference's shipped scene (flattened form only) -------------
    sa = scene_bundle(REF_SCENE)
    save("scene_a", **sa)
    save("kat_trace_a", **sa, **run(REF_SCENE, "kat_trace", 4096, 11, 40))
    save("kat_samplers_a", **sa, **run(REF_SCENE, "kat_samplers", 1024, 5))
    save("kat_paths_a", **sa, **run(REF_SCENE, "kat_paths", 4096, 3, 40))
    img = run(REF_SCENE, "render", 7, 16, 16, 0.01, 160, 90)
    save("image_a_160x90_s16", **sa, params=np.array([7, 16, 16, 160, 90], np.int64), delta=np.float32(0.01), **img)
    img = run(REF_SCENE, "render", 9, 4, 32, 0.05, 96, 54)
    save("image_a_96x54_adaptive", **sa, params=np.array([9, 4, 32, 96, 54], np.int64), delta=np.float32(0.05), **img)

    with tempfile.TemporaryDirectory() as tmp:
        # --- nested transforms + depth of field: our own scene text -----------------
        xml = scenes.nested_scene_xml(width=96, height=54)
        p = os.path.join(tmp, "nested.xml")
        open(p, "w").write(xml)
        sn = scene_bundle(p)
        save("kat_trace_nested", xml=np.frombuffer(xml.encode(), np.uint8), **sn, **run(p, "kat_trace", 2048, 21, 40))
        img = run(p, "render", 5, 8, 8, 0.01, 96, 54)
        save("image_nested_96x54_s8", xml=np.frombuffer(xml.encode(), np.uint8), **sn,
             params=np.array([5, 8, 8, 96, 54], np.int64), delta=np.float32(0.01), **img)

        # --- sphere grid (the sphere/plane configuration, reduced) ------------------
        xml = scenes.spheres_scene_xml(seed=1001, grid=4, width=96, height=54)
        p = os.path.join(tmp, "spheres.xml")
        open(p, "w").write(xml)
        sb = scene_bundle(p)
        img = run(p, "render", 7, 8, 8, 0.01, 96, 54)
        save("image_spheres_96x54_s8", xml=np.frombuffer(xml.encode(), np.uint8), **sb,
             params=np.array([7, 8, 8, 96, 54], np.int64), delta=np.float32(0.01), **img)

        # --- triangle mesh + BVH (the 1M-triangle configuration, reduced to 1024) ---
        mesh = scenes.heightfield(nx=32, ny=16)
        scenes.write_obj(os.path.join(tmp, "mesh_small.obj"), mesh)
        xml = scenes.mesh_scene_xml("mesh_small.obj", width=96, height=54)
        p = os.path.join(tmp, "mesh.xml")
        open(p, "w").write(xml)
        sm = scene_bundle(p, cwd=tmp)
        save("scene_mesh", xml=np.frombuffer(xml.encode(), np.uint8), mesh_nx=np.int32(32), mesh_ny=np.int32(16), **sm)
        small = {k: v for k, v in sm.items()}
        save("kat_trace_mesh", **small, **run(p, "kat_trace", 4096, 31, 40, cwd=tmp))
        img = run(p, "render", 7, 8, 8, 0.01, 96, 54, cwd=tmp)
        save("image_mesh_96x54_s8", **small, params=np.array([7, 8, 8, 96, 54], np.int64), delta=np.float32(0.01), **img)
    whitted()
    photons()
    textures()
    config_c()
    photons_large()
    config_e()
    png()
    multi_material()


if __name__ == "__main__":
    main()
