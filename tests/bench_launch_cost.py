#!/usr/bin/env python3
"""How the ray-list kernel's time grows with the length of its list: camera rays of every eighth 16x16 tile of the
1M-triangle 1080p frame (what one rank of eight traces), 1..32 rays per pixel, through rt_trace_brief.
kernel ms = a + b * rays: a is what a launch costs whatever its size."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import lib, scenes  # noqa: E402
from raytracer_amd.dist import shard_pixels  # noqa: E402

w, h = 1920, 1080
mesh = scenes.heightfield(1000, 500)
hs = lib.HostScene.from_string(scenes.mesh_scene_xml("mesh.obj", w, h))
hs.set_mesh("mesh.obj", mesh["v"], mesh["vt"], mesh["f"])
ctx = lib.Context(0)
ctx.upload(hs.desc)
cam = hs.camera
pos, d, up = np.array(cam.pos[:], np.float32), np.array(cam.dir[:], np.float32), np.array(cam.up[:], np.float32)
d /= np.linalg.norm(d)
x = np.cross(d, up); x /= np.linalg.norm(x)
y = np.cross(x, d)
half_w = np.tan(np.radians(cam.fov) / 2)
half_h = half_w * h / w
rng = np.random.RandomState(1)
for world in (8, 1):
    px = shard_pixels(w, h, 16, 0, world)
    rows = []
    for spp in (1, 2, 4, 8, 16, 32) if world == 8 else (1, 2, 4):
        ii = np.repeat(px % w, spp); jj = np.repeat(px // w, spp)
        sx = (-half_w + 2 * half_w * (ii + rng.rand(len(ii))) / w).astype(np.float32)[:, None]
        sy = (half_h - 2 * half_h * (jj + rng.rand(len(ii))) / h).astype(np.float32)[:, None]
        dirs = (d[None] + sx * x[None] + sy * y[None]).astype(np.float32)
        rays = np.concatenate([np.broadcast_to(pos, dirs.shape), dirs], 1).astype(np.float32)
        ms = []
        for rep in range(4):
            ctx.trace_brief(rays)
            ms.append(ctx.counters().kernel_ms)
        rows.append((len(rays), min(ms[1:])))
        print(f"tiles 1/{world}: {len(rays):9d} rays  {min(ms[1:]):7.3f} ms  {len(rays) / min(ms[1:]) / 1e6:7.2f} Grays/s", flush=True)
    n = np.array([r[0] for r in rows], np.float64); t = np.array([r[1] for r in rows])
    b, a = np.polyfit(n, t, 1)
    print(f"tiles 1/{world}: ms = {a:.3f} + {b * 1e6:.4f} per million rays")
