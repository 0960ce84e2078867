#!/usr/bin/env python3
"""Photon-map configuration in isolation: fill (emission rounds + kd-tree build) and the Whitted-style frame
that gathers from the maps, on the reference's own scene geometry.

    python tests/bench_photon.py [--photons 10000000] [--caustic 1000000] [--size 1920x1080] [--spp 16 64]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import oraclelib as O  # noqa: E402  (only for the scene arrays of the golden fixture)
from raytracer_amd import abi, lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--photons", type=int, default=10_000_000)
    ap.add_argument("--caustic", type=int, default=1_000_000)
    ap.add_argument("--size", default="1920x1080")
    ap.add_argument("--spp", type=int, nargs=2, default=[16, 64])
    ap.add_argument("--bounces", type=int, default=5)
    ap.add_argument("--flags", type=int, default=6)
    ap.add_argument("--count", action="store_true")
    ap.add_argument("--gather-only", type=int, default=0, help="time the batch irradiance kernel on about this many queries instead of rendering")
    args = ap.parse_args()
    w, h = [int(x) for x in args.size.split("x")]
    s = O.scene_from_golden(O.load_golden("scene_a"))
    ctx = lib.Context(0)
    ctx.upload(s.desc)
    ctx.set_camera(s.camera, w, h)
    pp = abi.PhotonParams(flags=args.flags, n_global=args.photons, n_caustic=args.caustic, max_segments=64, seed=7)
    t0 = time.perf_counter()
    ctx.photons_fill(pp)
    t1 = time.perf_counter()
    c = ctx.counters()
    print(f"fill: {t1 - t0:.3f} s wall, emission kernels {c.kernel_ms:.1f} ms, {c.closest_rays / 1e6:.1f} M photon rays "
          f"({c.closest_rays / c.kernel_ms / 1e3:.1f} Mrays/s), {c.camera_paths / 1e6:.1f} M paths", flush=True)
    if args.gather_only:
        import numpy as np
        cam = s.camera
        n = args.gather_only
        rng = np.random.RandomState(1)
        # coherent queries: primary hits of camera rays in scan-line order
        side = int(np.sqrt(n * 16 / 9))
        ww, hh = side, max(1, side * 9 // 16)
        pos, d, up = np.array(cam.pos[:], np.float32), np.array(cam.dir[:], np.float32), np.array(cam.up[:], np.float32)
        d /= np.linalg.norm(d)
        x = np.cross(d, up); x /= np.linalg.norm(x)
        y = np.cross(x, d)
        half_h = np.tan(np.radians(cam.fov) / 2); half_w = half_h * ww / hh
        jj, ii = np.mgrid[0:hh, 0:ww]
        sx = (-half_w + 2 * half_w * (ii + 0.5) / ww).astype(np.float32).reshape(-1, 1)
        sy = (half_h - 2 * half_h * (jj + 0.5) / hh).astype(np.float32).reshape(-1, 1)
        dirs = (d[None] + sx * x[None] + sy * y[None]).astype(np.float32)
        rays = np.concatenate([np.broadcast_to(pos, dirs.shape), dirs], 1).astype(np.float32)
        hits = ctx.trace(rays)
        # only where the shader looks photons up: surfaces with a diffuse colour (MtlBlinn::IsPhotonSurface)
        diffuse = np.array([sum(s.desc.materials[s.desc.nodes[i].material].diffuse[:]) > 0 if s.desc.nodes[i].prim else False
                            for i in range(s.desc.n_nodes)])
        ok = (hits["node"] >= 0) & diffuse[np.maximum(hits["node"], 0)]
        P = hits["p"][ok]
        N = hits["n"][ok]
        N /= np.linalg.norm(N, axis=1, keepdims=True)
        for label, order in (("coherent", np.arange(len(P))), ("shuffled", rng.permutation(len(P)))):
            for which in (0, 1):
                e = ctx.photons_irradiance(which, P[order], N[order], 3.0)
                c = ctx.counters()
                print(f"gather {label} map {which}: {len(P)} queries in {c.kernel_ms:.1f} ms = {len(P) / c.kernel_ms / 1e3:.2f} M gathers/s, "
                      f"{c.photon_visits / len(P):.0f} visits/gather, {c.photon_visits * 32 / c.kernel_ms / 1e6:.1f} GB/s of photon records, mean E {e.mean():.4f}", flush=True)
        return
    prm = abi.default_params(seed=7, min_spp=args.spp[0], max_spp=args.spp[1], max_bounces=args.bounces, mode=abi.RT_MODE_WHITTED,
                             photon_flags=args.flags, gather_radius=3.0)
    ctx.set_counting(args.count)
    for rep in range(2):
        t0 = time.perf_counter()
        fr = ctx.render(prm)
        dt = time.perf_counter() - t0
        c = ctx.counters()
        rays = c.closest_rays + c.shadow_rays
        print(f"render: {dt:.3f} s, kernels {c.trace_kernel_ms:.1f} ms in {c.trace_launches} launches, {rays / 1e6:.1f} M rays "
              f"({rays / dt / 1e6:.1f} Mrays/s), {c.gathers / 1e6:.2f} M gathers ({c.gathers / dt / 1e6:.2f} M/s), "
              f"{c.photon_visits / 1e9:.2f} G node visits, mean spp {fr.sample_count.mean():.1f}", flush=True)


if __name__ == "__main__":
    main()
