#!/usr/bin/env python3
"""tests/bench_many_objects.py: scenes of many objects with and without the hierarchy over the objects' boxes
(RT_TLAS_MIN=0 walks the flat node list, as the reference does): the sphere grid at 10^4 spheres through the
Whitted-style shader and through the path tracer, and a 1M-triangle mesh among 1024 spheres through the path tracer
(wavefront pipeline: the ray-list kernel's scene pass).  Prints one line per case."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(tlas_min):
    os.environ["RT_TLAS_MIN"] = tlas_min
    from raytracer_amd import abi, lib, scenes
    ctx = lib.Context(0)
    w, h = 960, 540
    cases = []
    big = lib.HostScene.from_string(scenes.spheres_scene_xml(grid=100, width=w, height=h))
    cases.append(("10^4 spheres, Blinn-Phong + soft shadows", big, abi.default_params(mode=abi.RT_MODE_WHITTED, min_spp=4, max_spp=4, max_bounces=0, seed=7)))
    cases.append(("10^4 spheres, path tracer (50 bounces)", big, abi.default_params(min_spp=4, max_spp=4, seed=7)))
    many = lib.HostScene.from_string(scenes.mesh_and_spheres_scene_xml(grid=32, width=w, height=h))
    many.set_mesh("mesh_a.obj", **{k: v for k, v in scenes.heightfield(1000, 500).items() if k in ("v", "vt", "f")})
    many.set_mesh("mesh_b.obj", **{k: v for k, v in scenes.heightfield(64, 32).items() if k in ("v", "vt", "f")})
    cases.append(("1M-triangle mesh + 1024 spheres, path tracer", many, abi.default_params(min_spp=4, max_spp=4, seed=7)))
    for name, hs, prm in cases:
        ctx.upload(hs.desc)
        ctx.set_camera(hs.camera, w, h)
        ctx.render_device(prm)
        t0 = time.perf_counter()
        n = 2
        for _ in range(n):
            ctx.render_device(prm)
        dt = (time.perf_counter() - t0) / n
        c = ctx.counters()
        rays = c.closest_rays + c.shadow_rays
        print(f"RT_TLAS_MIN={tlas_min:2s} {name:48s} {hs.desc.n_nodes:6d} nodes  {dt * 1e3:9.2f} ms/frame  {rays / dt / 1e6:9.1f} Mrays/s", flush=True)
    ctx.close()


if __name__ == "__main__":
    run(sys.argv[1] if len(sys.argv) > 1 else "16")
