"""Development check of the GPU-built BVH: the 1M-triangle scene with RT_GPU_BVH=1 against the reference's band and rays."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oraclelib as O
from raytracer_amd import abi, lib, scenes

def bits(a): return np.ascontiguousarray(a, np.float32).view(np.uint32)

g = O.load_golden("kat_trace_mesh")
os.environ["RT_GPU_BVH"] = "1"
c = lib.Context(0)
c.upload(O.scene_from_golden(g).desc)
print("small mesh build ms", c.counters().bvh_build_ms)
hits = c.trace(g["kat_rays"])
print("small: node equal", np.array_equal(hits["node"], g["kat_id"]), "z equal", np.array_equal(bits(hits["z"]), bits(g["kat_hit"][:, 0])),
      "mismatch", int((hits["node"] != g["kat_id"]).sum()), int((bits(hits["z"]) != bits(g["kat_hit"][:, 0])).sum()))
print("shadow equal", np.array_equal(c.trace_shadow(g["kat_rays"], g["kat_tmax"]), g["kat_flags"][:, 2]))
b = c.trace_brief(g["kat_rays"])
print("brief equal", np.array_equal(b["node"], g["kat_id"]), np.array_equal(bits(b["t"]), bits(g["kat_hit"][:, 0])))

g = O.load_golden("config_c_1m")
mesh = scenes.heightfield(int(g["mesh_nx"]), int(g["mesh_ny"]))
hs = lib.HostScene.from_string(bytes(g["xml"]).decode())
hs.set_mesh("mesh_1m.obj", mesh["v"], mesh["vt"], mesh["f"])
t0 = time.time(); c.upload(hs.desc); print("1M upload s", time.time() - t0, "build ms", c.counters().bvh_build_ms)
hits = c.trace(g["kat_rays"])
print("1M rays: node mismatches", int((hits["node"] != g["kat_id"]).sum()), "z mismatches", int((bits(hits["z"]) != bits(g["kat_hit"][:, 0])).sum()))
print("1M shadow mismatches", int((c.trace_shadow(g["kat_rays"], g["kat_tmax"]) != g["kat_flags"][:, 2]).sum()))
seed, mn, mx, w, h, y0, y1 = [int(x) for x in g["params"]]
c.set_camera(hs.camera, w, h)
fr = c.render(abi.default_params(seed=seed, min_spp=mn, max_spp=mx, delta_max=float(g["delta"])))
print("band id mismatches", int((fr.object_id[y0:y1] != g["object_id"]).sum()), "z mismatches", int((bits(fr.z[y0:y1]) != bits(g["z"])).sum()), "psnr", O.psnr(fr.rgb[y0:y1], g["rgb"]))
for spp in (64,):
    prm = abi.default_params(seed=7, min_spp=spp, max_spp=spp)
    c.render(prm); t0 = time.time(); c.render(prm); print("frame ms with the GPU-built BVH", (time.time() - t0) * 1e3)
c.close()
os.environ["RT_GPU_BVH"] = "0"
c = lib.Context(0)
t0 = time.time(); c.upload(hs.desc); print("1M upload s (host BVH supplied)", time.time() - t0)
c.set_camera(hs.camera, w, h)
prm = abi.default_params(seed=7, min_spp=64, max_spp=64)
c.render(prm); t0 = time.time(); c.render(prm); print("frame ms with the reference's BVH", (time.time() - t0) * 1e3)
