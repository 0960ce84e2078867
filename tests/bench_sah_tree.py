#!/usr/bin/env python3
"""Is a surface-area-heuristic tree worth building?  The frames' four-wide walk over (a) the reference's mean-split tree,
(b) a full-sweep SAH tree built here in numpy (every split position on every axis tried at every node: the quality a
binned or clustering builder on the GPU could at best reach), both collapsed into four-wide records by the library
(RT_WIDE_GPU_BVH=0), and (c) the tree the library builds on the GPU.  python tests/bench_sah_tree.py [quads_x quads_y w h spp]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import abi, lib, scenes  # noqa: E402


def sah_tree(v, f, max_leaf):
    tv = v[f]                                   # [nf, 3, 3]
    lo, hi = tv.min(1), tv.max(1)
    cen = 0.5 * (lo + hi)
    nodes = [np.zeros(7, np.uint32), None]      # node 0 unused, root = 1
    elements = []
    todo = [(1, np.arange(len(f)))]

    def area(l, h):
        d = h - l
        return d[..., 0] * d[..., 1] + d[..., 1] * d[..., 2] + d[..., 2] * d[..., 0]

    def record(l, h, data):
        w = np.zeros(7, np.uint32)
        w[:3] = l.astype(np.float32).view(np.uint32); w[3:6] = h.astype(np.float32).view(np.uint32); w[6] = data
        return w

    while todo:
        node, idx = todo.pop()
        l, h = lo[idx].min(0), hi[idx].max(0)
        n = len(idx)
        if n <= max_leaf:
            nodes[node] = record(l, h, 0x80000000 | ((n - 1) << 28) | len(elements))
            elements.extend(idx.tolist())
            continue
        best = (np.inf, None, None)
        for a in range(3):
            o = idx[np.argsort(cen[idx, a], kind="stable")]
            ll, lh = np.minimum.accumulate(lo[o], 0), np.maximum.accumulate(hi[o], 0)
            rl, rh = np.minimum.accumulate(lo[o][::-1], 0)[::-1], np.maximum.accumulate(hi[o][::-1], 0)[::-1]
            k = np.arange(1, n)
            cost = area(ll[:-1], lh[:-1]) * k + area(rl[1:], rh[1:]) * (n - k)
            i = int(np.argmin(cost))
            if cost[i] < best[0]:
                best = (cost[i], o, i + 1)
        _, o, cut = best
        c = len(nodes)
        nodes.extend([None, None])
        nodes[node] = record(l, h, c)
        todo.append((c, o[:cut])); todo.append((c + 1, o[cut:]))
    return np.stack(nodes), np.array(elements, np.uint32)


def median_tree(v, f, pair):
    """Spatial-median splits along the longest axis of the centroids' bounds (what a hierarchy over sorted keys amounts
    to), over single triangles (pair False: two per leaf at most) or over pairs of consecutive faces that share an edge
    (pair True: a pair is a leaf)."""
    tv = v[f]
    lo, hi = tv.min(1), tv.max(1)
    if pair:
        n2 = len(f) // 2
        shared = np.array([len(set(f[2 * i]) & set(f[2 * i + 1])) == 2 for i in range(n2)])
        assert shared.all() and len(f) % 2 == 0, "the experiment wants a mesh of quads cut in two"
        ulo, uhi = np.minimum(lo[0::2], lo[1::2]), np.maximum(hi[0::2], hi[1::2])
        faces_of = lambda u: [2 * u, 2 * u + 1]
    else:
        ulo, uhi = lo, hi
        faces_of = lambda u: [u]
    cen = 0.5 * (ulo + uhi)
    nodes = [np.zeros(7, np.uint32), None]
    elements = []
    todo = [(1, np.arange(len(ulo)))]
    max_units = 1 if pair else 2

    def record(l, h, data):
        w = np.zeros(7, np.uint32)
        w[:3] = l.astype(np.float32).view(np.uint32); w[3:6] = h.astype(np.float32).view(np.uint32); w[6] = data
        return w

    while todo:
        node, idx = todo.pop()
        l, h = ulo[idx].min(0), uhi[idx].max(0)
        if len(idx) <= max_units:
            faces = [x for u in idx.tolist() for x in faces_of(u)]
            nodes[node] = record(l, h, 0x80000000 | ((len(faces) - 1) << 28) | len(elements))
            elements.extend(faces)
            continue
        c = cen[idx]
        a = int(np.argmax(c.max(0) - c.min(0)))
        o = idx[np.argsort(c[:, a], kind="stable")]
        cut = len(o) // 2
        k = len(nodes)
        nodes.extend([None, None])
        nodes[node] = record(l, h, k)
        todo.append((k, o[:cut])); todo.append((k + 1, o[cut:]))
    return np.stack(nodes), np.array(elements, np.uint32)


def curve_order(cen):
    """Units along the builder's space-filling curve: 21 bits per axis, every key bit from the axis whose cells are
    still the widest (device/lbvh.h: lbvh_axis_order)."""
    lo, hi = cen.min(0), cen.max(0)
    size = (hi - lo).astype(np.float64).copy()
    cells = np.minimum(((cen - lo) / np.maximum(hi - lo, 1e-30) * 2097151.0).astype(np.uint64), 2097151)
    left = [21, 21, 21]
    key = np.zeros(len(cen), np.uint64)
    for _ in range(63):
        a = max((k for k in range(3) if left[k] > 0), key=lambda k: size[k])
        size[a] *= 0.5; left[a] -= 1
        key = (key << np.uint64(1)) | ((cells[:, a] >> np.uint64(left[a])) & np.uint64(1))
    return np.argsort(key, kind="stable")


def curve_sah_tree(v, f):
    """Pairs of triangles in curve order; every node splits its stretch of the curve where the surface-area cost is
    smallest (one sweep along the order, not along the three axes): what a builder could do level by level with
    segmented scans over the sorted units."""
    tv = v[f]
    lo, hi = tv.min(1), tv.max(1)
    ulo, uhi = np.minimum(lo[0::2], lo[1::2]), np.maximum(hi[0::2], hi[1::2])
    o = curve_order(0.5 * (ulo + uhi))
    ulo, uhi = ulo[o], uhi[o]
    nodes = [np.zeros(7, np.uint32), None]
    elements = []
    todo = [(1, 0, len(o))]

    def area(l, h):
        d = h - l
        return d[..., 0] * d[..., 1] + d[..., 1] * d[..., 2] + d[..., 2] * d[..., 0]

    def record(l, h, data):
        w = np.zeros(7, np.uint32)
        w[:3] = l.astype(np.float32).view(np.uint32); w[3:6] = h.astype(np.float32).view(np.uint32); w[6] = data
        return w

    while todo:
        node, a, b = todo.pop()
        l, h = ulo[a:b].min(0), uhi[a:b].max(0)
        n = b - a
        if n == 1:
            u = int(o[a])
            nodes[node] = record(l, h, 0x80000000 | (1 << 28) | len(elements))
            elements.extend([2 * u, 2 * u + 1])
            continue
        ll, lh = np.minimum.accumulate(ulo[a:b], 0), np.maximum.accumulate(uhi[a:b], 0)
        rl, rh = np.minimum.accumulate(ulo[a:b][::-1], 0)[::-1], np.maximum.accumulate(uhi[a:b][::-1], 0)[::-1]
        k = np.arange(1, n)
        cut = a + 1 + int(np.argmin(area(ll[:-1], lh[:-1]) * k + area(rl[1:], rh[1:]) * (n - k)))
        c = len(nodes)
        nodes.extend([None, None])
        nodes[node] = record(l, h, c)
        todo.append((c, a, cut)); todo.append((c + 1, cut, b))
    return np.stack(nodes), np.array(elements, np.uint32)


def frames(hs, w, h, spp, env):
    for k, val in env.items():
        os.environ[k] = val
    ctx = lib.Context(0)
    for k in env:
        del os.environ[k]
    ctx.upload(hs.desc)
    ctx.set_camera(hs.camera, w, h)
    prm = abi.default_params(seed=1, min_spp=spp, max_spp=spp, max_bounces=50)
    ctx.render(prm)
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.render(prm)
    ms = (time.perf_counter() - t0) / 5 * 1e3
    c = ctx.counters()
    ctx.close()
    return ms, c.trace_kernel_ms


def main():
    a = [int(x) for x in sys.argv[1:]] or [250, 128, 960, 540, 8]
    qx, qy, w, h, spp = a
    mesh = scenes.heightfield(qx, qy)
    hs = lib.HostScene.from_string(scenes.mesh_scene_xml("mesh.obj", w, h))
    hs.set_mesh("mesh.obj", mesh["v"], mesh["vt"], mesh["f"])
    print(f"{len(mesh['f'])} triangles, {w}x{h} x {spp} spp")
    print("the library's own tree (built on the GPU)            %8.2f ms/frame, ray launches %8.2f ms" % frames(hs, w, h, spp, {}))
    print("the reference's mean-split tree, collapsed            %8.2f ms/frame, ray launches %8.2f ms" % frames(hs, w, h, spp, {"RT_WIDE_GPU_BVH": "0"}))
    m = hs.desc.meshes[0]
    for pair in (False, True):
        nodes, elements = median_tree(np.asarray(mesh["v"], np.float64), np.asarray(mesh["f"], np.int64), pair)
        nodes = np.ascontiguousarray(nodes, np.uint32); elements = np.ascontiguousarray(elements, np.uint32)
        m.bvh_nodes = abi.C.cast(nodes.ctypes.data, abi.C.POINTER(abi.BvhNode)); m.n_bvh_nodes = len(nodes)
        m.bvh_elements = elements.ctypes.data_as(abi.u32p); m.bvh_depth = abi.bvh_depth(nodes)
        print(f"median splits over {'pairs of triangles that share an edge' if pair else 'triangles, two per leaf'}, collapsed"
              "  %8.2f ms/frame, ray launches %8.2f ms" % frames(hs, w, h, spp, {"RT_WIDE_GPU_BVH": "0"}), flush=True)
    nodes, elements = curve_sah_tree(np.asarray(mesh["v"], np.float64), np.asarray(mesh["f"], np.int64))
    nodes = np.ascontiguousarray(nodes, np.uint32); elements = np.ascontiguousarray(elements, np.uint32)
    m.bvh_nodes = abi.C.cast(nodes.ctypes.data, abi.C.POINTER(abi.BvhNode)); m.n_bvh_nodes = len(nodes)
    m.bvh_elements = elements.ctypes.data_as(abi.u32p); m.bvh_depth = abi.bvh_depth(nodes)
    print(f"pairs in curve order, every node split where the cost along the curve is smallest (depth {m.bvh_depth}), collapsed"
          "  %8.2f ms/frame, ray launches %8.2f ms" % frames(hs, w, h, spp, {"RT_WIDE_GPU_BVH": "0"}), flush=True)
    for max_leaf in (4, 2):
        t0 = time.time()
        nodes, elements = sah_tree(np.asarray(mesh["v"], np.float64), np.asarray(mesh["f"], np.int64), max_leaf)
        nodes = np.ascontiguousarray(nodes, np.uint32); elements = np.ascontiguousarray(elements, np.uint32)
        m.bvh_nodes = abi.C.cast(nodes.ctypes.data, abi.C.POINTER(abi.BvhNode)); m.n_bvh_nodes = len(nodes)
        m.bvh_elements = elements.ctypes.data_as(abi.u32p); m.bvh_depth = abi.bvh_depth(nodes)
        print(f"full-sweep SAH tree, <= {max_leaf} triangles per leaf (built in {time.time() - t0:.0f} s, depth {m.bvh_depth}), collapsed"
              "  %8.2f ms/frame, ray launches %8.2f ms" % frames(hs, w, h, spp, {"RT_WIDE_GPU_BVH": "0"}), flush=True)


if __name__ == "__main__":
    main()
