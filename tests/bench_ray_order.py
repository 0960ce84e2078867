#!/usr/bin/env python3
"""What the order of a list of incoherent rays is worth to the ray-list kernel's four-wide walk.

Second-bounce rays of the headline scene (origins: where the camera rays of a 1920x1080 frame hit, a few samples per
pixel; directions: cosine-distributed about the surface normal, as a diffuse bounce sends them) go through
rt_trace_brief's wide walk (RT_QUERY_WIDE=1) in four orders: as the pipeline's lists hold them (pixel after pixel, a
pixel's samples together), sorted by direction octant and then by the Morton code of the origin, sorted by the Morton
code of the origin alone, and shuffled.  Kernel time only (the sort itself is not counted): the most an ordering pass
between bounces could win.  Run on the GPU box: python tests/bench_ray_order.py [samples per pixel]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RT_QUERY_WIDE"] = "1"
from raytracer_amd import abi, lib, scenes  # noqa: E402


def morton(q):  # q: [n, 3] integers below 1024 -> 30-bit codes
    def spread(x):
        x = x.astype(np.uint64)
        x = (x | (x << 16)) & 0x030000FF
        x = (x | (x << 8)) & 0x0300F00F
        x = (x | (x << 4)) & 0x030C30C3
        x = (x | (x << 2)) & 0x09249249
        return x
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)


def main():
    spp = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    w, h = 1920, 1080
    mesh = scenes.heightfield(1000, 500)
    hs = lib.HostScene.from_string(scenes.mesh_scene_xml("mesh.obj", w, h))
    hs.set_mesh("mesh.obj", mesh["v"], mesh["vt"], mesh["f"])
    ctx = lib.Context(0)
    ctx.upload(hs.desc)
    cam = abi.Camera.from_buffer_copy(hs.camera)
    # camera rays through pixel centres (pinhole), then their hits
    pos, dirn, up = np.array(cam.pos), np.array(cam.dir), np.array(cam.up)
    dirn = dirn / np.linalg.norm(dirn)
    right = np.cross(dirn, up); right /= np.linalg.norm(right)
    upv = np.cross(right, dirn)
    half_h = np.tan(np.radians(cam.fov) / 2.0); half_w = half_h * w / h
    ys, xs = np.mgrid[0:h, 0:w]
    sx = (-half_w + 2 * half_w * (xs + 0.5) / w).ravel(); sy = (half_h - 2 * half_h * (ys + 0.5) / h).ravel()
    d = dirn[None, :] + sx[:, None] * right[None, :] + sy[:, None] * upv[None, :]
    rays = np.concatenate([np.broadcast_to(pos, d.shape), d], axis=1).astype(np.float32)
    hits = ctx.trace(rays)
    ok = hits["node"] >= 0
    p, n = hits["p"][ok], hits["n"][ok]
    n = n / np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-20)
    n = np.where((np.einsum("ij,ij->i", n, d[ok]) > 0)[:, None], -n, n)  # towards the viewer's side
    rng = np.random.default_rng(1)
    p = np.repeat(p, spp, axis=0); n = np.repeat(n, spp, axis=0)  # a pixel's samples next to each other, as in the lists
    u1, u2 = rng.random(len(p)), rng.random(len(p))
    r, phi = np.sqrt(u1), 2 * np.pi * u2
    a = np.where(np.abs(n[:, :1]) < 0.9, np.array([[1.0, 0, 0]]), np.array([[0, 1.0, 0]]))
    t = np.cross(n, a); t /= np.linalg.norm(t, axis=1, keepdims=True)
    b = np.cross(n, t)
    out = t * (r * np.cos(phi))[:, None] + b * (r * np.sin(phi))[:, None] + n * np.sqrt(1 - u1)[:, None]
    second = np.concatenate([p + 1e-3 * n, out], axis=1).astype(np.float32)
    lo, hi = second[:, :3].min(0), second[:, :3].max(0)
    q = np.minimum(((second[:, :3] - lo) / np.maximum(hi - lo, 1e-20) * 1024).astype(np.int64), 1023)
    code = morton(q)
    octant = ((out[:, 0] < 0).astype(np.uint64) | ((out[:, 1] < 0).astype(np.uint64) << 1) | ((out[:, 2] < 0).astype(np.uint64) << 2))
    orders = {
        "as the lists hold them (pixel order)": np.arange(len(second)),
        "sorted by direction octant, then Morton code of the origin": np.argsort((octant << 30) | code, kind="stable"),
        "sorted by Morton code of the origin": np.argsort(code, kind="stable"),
        "shuffled": rng.permutation(len(second)),
    }
    print(f"{len(second) / 1e6:.1f} M second-bounce rays of the 1M-triangle scene ({spp} per pixel)")
    for name, idx in orders.items():
        rr = np.ascontiguousarray(second[idx])
        ctx.trace_brief(rr)  # warm
        best = 1e9
        for _ in range(3):
            ctx.trace_brief(rr)
            best = min(best, ctx.counters().trace_kernel_ms)
        print(f"{name:62s} {best:8.2f} ms  {len(rr) / best / 1e6:7.2f} Grays/s", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
