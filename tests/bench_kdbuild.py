#!/usr/bin/env python3
"""Time rt_photons_build (the GPU kd-tree build) on synthetic photons already in device memory.

    python tests/bench_kdbuild.py [--photons 10000000] [--reps 3]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import oraclelib as O  # noqa: E402  (scene arrays of the golden fixture only)
from raytracer_amd import abi, lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--photons", type=int, nargs="+", default=[10_000_000])
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    ctx = lib.Context(0)
    ctx.upload(O.scene_from_golden(O.load_golden("scene_a")).desc)
    for n in args.photons:
        rng = np.random.default_rng(1)
        p = np.zeros(n, dtype=abi.PHOTON_DTYPE)
        pos = rng.uniform(-10, 10, size=(n, 3)).astype(np.float32)
        face = rng.integers(0, 6, size=n)
        pos[np.arange(n), face % 3] = np.where(face < 3, -10.0, 10.0).astype(np.float32)  # photons lie on walls
        p["pos"] = pos
        p["power"] = 1.0
        dev = torch.from_numpy(p.view(np.uint8).reshape(n, 32)).cuda()
        torch.cuda.synchronize()
        for rep in range(args.reps):
            t0 = time.perf_counter()
            ctx.photons_build(abi.RT_MAP_GLOBAL, device_ptr=dev.data_ptr(), n=n)
            dt = time.perf_counter() - t0
            print(f"build {n} photons: {dt * 1e3:.1f} ms wall, {ctx.counters().kernel_ms:.1f} ms on the stream", flush=True)


if __name__ == "__main__":
    main()
