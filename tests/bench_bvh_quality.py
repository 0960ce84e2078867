#!/usr/bin/env python3
"""tests/bench_bvh_quality.py: box / triangle tests per frame and frame time through the reference's BVH (host build)
and through the BVH built on the GPU, on the small mesh scene and on the 1M-triangle scene."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from raytracer_amd import abi, lib, scenes  # noqa: E402


def main():
    ctx = lib.Context(0)
    for label, nx, ny, w, h, spp in (("1k triangles", 32, 16, 320, 180, 8), ("1M triangles", 1000, 500, 1920, 1080, 8), ("10M triangles", 5000, 1000, 1920, 1080, 4)):
        mesh = scenes.heightfield(nx, ny)
        res = {}
        for gpu in (False, True):
            hs = lib.HostScene.from_string(scenes.mesh_scene_xml("mesh.obj", w, h))
            hs.set_gpu_bvh(gpu)
            t0 = time.perf_counter()
            hs.set_mesh("mesh.obj", mesh["v"], mesh["vt"], mesh["f"])
            host_s = time.perf_counter() - t0
            ctx.upload(hs.desc)
            build_ms = ctx.counters().bvh_build_ms
            ctx.set_camera(hs.camera, w, h)
            prm = abi.default_params(min_spp=spp, max_spp=spp, seed=7)
            ctx.set_counting(True)
            ctx.render_device(prm)
            c = ctx.counters()
            ctx.set_counting(False)
            ctx.render_device(prm)
            t0 = time.perf_counter()
            for _ in range(3):
                ctx.render_device(prm)
            ms = (time.perf_counter() - t0) / 3 * 1e3
            res[gpu] = (c.box_tests, c.tri_tests, ms)
            print(f"{label:14s} {'GPU-built BVH' if gpu else 'reference BVH':14s} box tests {c.box_tests:13d} triangle tests {c.tri_tests:12d} frame {ms:8.2f} ms"
                  f"  (mesh + BVH on the host {host_s:5.2f} s, BVH on the GPU {build_ms:6.1f} ms)", flush=True)
        print(f"{label:14s} GPU-built / reference: box tests x {res[True][0] / res[False][0]:.3f}, triangle tests x {res[True][1] / res[False][1]:.3f}, frame x {res[True][2] / res[False][2]:.3f}", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
