#!/usr/bin/env python3
"""Traversal in isolation: rays/s of the query kernels on three ray populations of the
1M-triangle scene - camera rays (coherent), their shadow rays, and diffuse bounce rays
(incoherent) - checked against each other bit for bit.

    python tests/bench_trace.py [--size 960x540] [--reps 3]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import lib, scenes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="1920x1080")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--quads", default="1000x500")
    ap.add_argument("--spp", type=int, default=4, help="camera rays per pixel")
    ap.add_argument("--only", default="", help="camera | shadow | bounce: run just this population, ray-list kernel only (profiling)")
    args = ap.parse_args()
    w, h = [int(x) for x in args.size.split("x")]
    nx, ny = [int(x) for x in args.quads.split("x")]
    mesh = scenes.heightfield(nx, ny)
    hs = lib.HostScene.from_string(scenes.mesh_scene_xml("mesh.obj", w, h))
    hs.set_mesh("mesh.obj", mesh["v"], mesh["vt"], mesh["f"])
    ctx = lib.Context(0)
    ctx.upload(hs.desc)
    cam = hs.camera
    pos, d, up = np.array(cam.pos[:], np.float32), np.array(cam.dir[:], np.float32), np.array(cam.up[:], np.float32)
    d /= np.linalg.norm(d)
    x = np.cross(d, up); x /= np.linalg.norm(x)
    y = np.cross(x, d)
    half_h = np.tan(np.radians(cam.fov) / 2)
    half_w = half_h * w / h
    rng = np.random.RandomState(1)
    # pixel-major, samples of a pixel adjacent: the order the renderer hands paths to lanes
    jj, ii, _ = np.mgrid[0:h, 0:w, 0:args.spp]
    sx = (-half_w + 2 * half_w * (ii + rng.rand(h, w, args.spp)) / w).astype(np.float32).reshape(-1, 1)
    sy = (half_h - 2 * half_h * (jj + rng.rand(h, w, args.spp)) / h).astype(np.float32).reshape(-1, 1)
    dirs = (d[None] + sx * x[None] + sy * y[None]).astype(np.float32)
    primary = np.concatenate([np.broadcast_to(pos, dirs.shape), dirs], 1).astype(np.float32)

    hits = ctx.trace(primary)
    surf = hits["node"] >= 0
    p = hits["p"][surf]
    n = hits["n"][surf]
    n /= np.maximum(1e-20, np.linalg.norm(n, axis=1, keepdims=True))
    n = np.where((np.einsum("ij,ij->i", n, primary[surf, 3:]) > 0)[:, None], -n, n).astype(np.float32)
    light = np.array(hs.desc.lights[0].position[:], np.float32)
    o = (p + n * 0.002).astype(np.float32)
    to_l = light[None] - o
    dist = np.linalg.norm(to_l, axis=1).astype(np.float32)
    shadow = np.concatenate([o, to_l / dist[:, None]], 1).astype(np.float32)
    # cosine-weighted bounce directions about n
    r1, r2 = rng.rand(len(o)).astype(np.float32), rng.rand(len(o)).astype(np.float32)
    a = np.where(np.abs(n[:, :1]) > 0.9, np.array([[0, 1, 0]], np.float32), np.array([[1, 0, 0]], np.float32))
    t1 = np.cross(n, a); t1 /= np.linalg.norm(t1, axis=1, keepdims=True)
    t2 = np.cross(n, t1)
    rr, ph = np.sqrt(r1), 2 * np.pi * r2
    bd = t1 * (rr * np.cos(ph))[:, None] + t2 * (rr * np.sin(ph))[:, None] + n * np.sqrt(1 - r1)[:, None]
    bounce = np.concatenate([o, bd], 1).astype(np.float32)

    def timed(fn, *a):
        best, out = 1e30, None
        for _ in range(args.reps):
            out = fn(*a)
            best = min(best, ctx.counters().kernel_ms)
        return out, best

    print(f"scene: {len(mesh['f'])} triangles; {len(primary)} camera rays, {len(shadow)} shadow rays, {len(bounce)} bounce rays")
    for name, rays, tm in (("camera", primary, None), ("shadow", shadow, dist - 0.002), ("bounce", bounce, None)):
        if args.only:
            if name == args.only:
                got, ms = timed(ctx.trace_brief, rays, tm)
                print(f"{name:7s} ray-list {len(rays) / ms / 1e6:8.3f} Grays/s ({ms:7.2f} ms)")
                ph = list(ctx.counters().phase_cycles)
                if any(ph):
                    tot = sum(ph[0:4]) + ph[12]
                    for i, nm in enumerate(("refill", "scene", "walk", "leaf")):
                        print(f"   {nm:7s} {100 * ph[i] / tot:5.1f} % of wave time, {ph[4 + i]:10d} executions, "
                              f"{ph[8 + i] / max(1, ph[4 + i]):5.1f} lanes, {ph[i] / max(1, ph[4 + i]) * 10:7.0f} ns each")
                    print(f"   other   {100 * ph[12] / tot:5.1f} %")
            continue
        if tm is None:
            ref, ms_ref = timed(ctx.trace, rays)
            got, ms = timed(ctx.trace_brief, rays)
            same = np.array_equal(ref["z"].view(np.uint32), got["t"].view(np.uint32)) and np.array_equal(ref["node"], got["node"])
        else:
            ref, ms_ref = timed(ctx.trace_shadow, rays, tm)
            got, ms = timed(ctx.trace_brief, rays, tm)
            same = np.array_equal(ref, got["node"])
        print(f"{name:7s} lockstep {len(rays) / ms_ref / 1e6:8.3f} Grays/s ({ms_ref:7.2f} ms)   ray-list {len(rays) / ms / 1e6:8.3f} Grays/s ({ms:7.2f} ms)   identical: {same}")
        if not same:
            if tm is None:
                bad = np.flatnonzero((ref["z"].view(np.uint32) != got["t"].view(np.uint32)) | (ref["node"] != got["node"]))
                print(f"  {len(bad)} of {len(rays)} rays differ; first: ray {bad[0]} lockstep t={ref['z'][bad[0]]!r} node={ref['node'][bad[0]]} ray-list t={got['t'][bad[0]]!r} node={got['node'][bad[0]]}")
            else:
                bad = np.flatnonzero(ref != got["node"])
                print(f"  {len(bad)} of {len(rays)} shadow rays differ; first: ray {bad[0]}")
            sys.exit(1)


if __name__ == "__main__":
    main()
