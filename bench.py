#!/usr/bin/env python3
"""Headline benchmark: Mrays/s of the GPU path tracer on the 1M-triangle scene.

    python bench.py --gpus N --steps K --warmup W

One step = one frame of the workload: every rank renders its tile shard
(camera rays, BVH traversal, shading, resolve), the framebuffer is gathered
(RCCL all_gather, N > 1) and rank 0 copies the finished frame to the host
(the copy of frame i runs while frame i + 1 renders into a second set of
planes; the timed region ends when the last copy has).
A ray is one TraceRay or TraceShadowRay call equivalent of the reference
(src/main.cpp:115-125); rays are counted by the kernels, not estimated.

For N > 1 there is one process per GPU: started by torch.distributed.run (RANK / WORLD_SIZE in the
environment), or, when the script is started bare (`python bench.py --gpus 8`), by the script itself, which
then starts its ranks through torch.distributed.run as a child process before anything has touched a GPU and
relays rank 0's line (the reference starts its own workers too: src/main.cpp:599-606).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from raytracer_amd import abi, lib, scenes  # noqa: E402
from raytracer_amd.dist import FrameBuffers, FrameGather, PhotonFill  # noqa: E402

# algorithmic flop constants, hand-counted from the reference code (+,-,*,/,min,max,
# compare-select, sqrt = 1): ray into node space 33, hit back to the parent 48 per level,
# sphere closest 38 / shadow 33, plane 15, light sphere 44, slab test 31,
# Moeller-Trumbore 60, shading of one path vertex ~300
F_NODE, F_BACK, F_SPH, F_SPH_SH, F_PLANE, F_LIGHT, F_BOX, F_TRI, F_SHADE = 33, 48, 38, 33, 15, 44, 31, 60, 300
PEAK_FP32_TFLOPS = 157.3  # MI355X vector FP32 peak (counts FMA as 2; this path runs un-fused for bit-exactness)
PEAK_HBM_GBPS = 8000.0    # MI355X HBM3E
# algorithmic bytes of the reference's data structures: a BVH node is 28 B per box test; a triangle test reads
# 4 B element index + 12 B face + 36 B vertices; a shaded mesh hit reads 24 B of fn/ft and 72 B of vn/vt;
# a finished pixel writes 3 + 4 + 4 + 4 B
B_BOX, B_TRI, B_HIT, B_PIXEL = 28, 52, 96, 15

WORKLOADS = {
    # path tracer on a triangle mesh: (mesh quads x, y, width, height, spp)
    "mesh_1m_1080p_64spp": (1000, 500, 1920, 1080, 64),   # the headline configuration
    "mesh_64k_540p_8spp": (250, 128, 960, 540, 8),        # development size
    "mesh_10m_4k_adaptive": (5000, 1000, 3840, 2160, (16, 256)),  # 10M triangles, 4K, 16..256 spp with the variance stop
    # the reference's shipped scene exactly as it renders it: 1280x720, 128 spp, 50 bounces (ms/frame is the number to read)
    "scene_a_720p_128spp": ("cornell", None, 1280, 720, 128),
    # Whitted-style shader (project-12 snapshot): Blinn-Phong + adaptive soft shadows, no recursion
    "spheres_1080p_16spp_whitted": ("spheres", 8, 1920, 1080, 16),
    "spheres_1080p_16spp_path": ("spheres_path", 8, 1920, 1080, 16),  # the same scene through the path tracer
    # photon maps: 10M + 1M photons traced, balanced and gathered from, 16..64 spp adaptive, 5 bounces
    "photon_box_1080p": ("photons", (10_000_000, 1_000_000), 1920, 1080, (16, 64)),
    "photon_box_540p_small": ("photons", (1_000_000, 100_000), 960, 540, (16, 64)),  # development size
}
# what `value` is, per workload: the headline line carries BASELINE.json's metric string
METRICS = {
    "mesh_1m_1080p_64spp": "Mrays/sec (whole node) at 1/2/4/8 MI355X, 1M-tri 1080p",
    "mesh_64k_540p_8spp": "Mrays/sec, 64k-tri 540p 8 spp path tracing (development size)",
    "mesh_10m_4k_adaptive": "Mrays/sec, 10M-tri 4K adaptive 16..256 spp path tracing",
    "scene_a_720p_128spp": "Mrays/sec, scene.xml as shipped (1280x720, 128 spp); ms_per_step is ms/frame scene.xml",
    "spheres_1080p_16spp_whitted": "Mrays/sec, sphere/plane scene 1080p 16 spp, Blinn-Phong + shadows",
    "spheres_1080p_16spp_path": "Mrays/sec, sphere/plane scene 1080p 16 spp, path tracing (50 bounces)",
    "photon_box_1080p": "Mrays/sec, photon-mapped caustics scene 1080p, 10M + 1M photons traced and gathered",
    "photon_box_540p_small": "Mrays/sec, photon-mapped caustics scene 540p, 1M + 0.1M photons (development size)",
}
MESHES = {}  # generated meshes by workload, kept for the reference baseline (it reads the OBJ text)
B_PHOTON = 32  # a kd-tree node visit reads one 32-byte photon record; a stored photon writes one


def build_scene(name):
    """-> host scene, width, height, (min spp, max spp), triangles, mode, max_bounces, photon params or None"""
    kind, arg, w, h, spp = WORKLOADS[name]
    spp = spp if isinstance(spp, tuple) else (spp, spp)
    if kind == "spheres":
        return lib.HostScene.from_string(scenes.spheres_scene_xml(grid=arg, width=w, height=h)), w, h, spp, 0, abi.RT_MODE_WHITTED, 0, None
    if kind == "spheres_path":
        return lib.HostScene.from_string(scenes.spheres_scene_xml(grid=arg, width=w, height=h)), w, h, spp, 0, abi.RT_MODE_PATH, 50, None
    if kind == "cornell":
        return lib.HostScene.from_string(scenes.cornell_scene_xml(w, h)), w, h, spp, 0, abi.RT_MODE_PATH, 50, None
    if kind == "photons":
        # the reference's own scene: its Cornell box with mirror, glass and glossy spheres
        hs = lib.HostScene.from_string(scenes.cornell_scene_xml(w, h))
        pp = abi.PhotonParams(flags=abi.RT_PM_INDIRECT | abi.RT_PM_CAUSTICS, n_global=arg[0], n_caustic=arg[1], max_segments=64, seed=7)
        return hs, w, h, spp, 0, abi.RT_MODE_WHITTED, 5, pp
    mesh = scenes.heightfield(kind, arg)
    MESHES[name] = mesh
    hs = lib.HostScene.from_string(scenes.mesh_scene_xml("mesh.obj", w, h))
    hs.set_mesh("mesh.obj", mesh["v"], mesh["vt"], mesh["f"])
    return hs, w, h, spp, len(mesh["f"]), abi.RT_MODE_PATH, 50, None


def algorithmic_flops(desc, c, mode=abi.RT_MODE_PATH, object_tests=None):
    """Flops one frame needs by the reference's own algorithm, from the kernels' ray / visit counters.
    object_tests: price the scene walk on the object tests the kernels made (a diagnostic build counts them)
    instead of the reference's "every ray against every node"."""
    prims = [desc.nodes[i].prim for i in range(desc.n_nodes)]
    n_nodes, n_sph, n_pl = desc.n_nodes, prims.count(abi.RT_PRIM_SPHERE), prims.count(abi.RT_PRIM_PLANE)
    per_closest = F_NODE * n_nodes + F_SPH * n_sph + F_PLANE * n_pl + F_LIGHT * desc.n_lights
    per_shadow = F_NODE * n_nodes + F_SPH_SH * n_sph + F_PLANE * n_pl
    walk = c.closest_rays * per_closest + c.shadow_rays * per_shadow
    if object_tests is not None:  # an object test = the ray into the object's space + the average primitive test
        per_object = F_NODE + (F_SPH * n_sph + F_PLANE * n_pl) / max(1, n_sph + n_pl)
        walk = object_tests * per_object + c.closest_rays * F_LIGHT * desc.n_lights
    # path tracer: every shaded surface vertex sends exactly one shadow ray; Whitted-style shader: a vertex is a
    # closest hit on a surface (about as many as closest rays) and its shading is ~120 flops + ~60 per fan ray
    if mode == abi.RT_MODE_WHITTED:
        return (walk + c.closest_rays * (2 * F_BACK + 120) + c.shadow_rays * 60
                + F_BOX * c.box_tests + F_TRI * c.tri_tests + 20 * c.photon_visits)
    vertices = c.shadow_rays
    return walk + F_BOX * c.box_tests + F_TRI * c.tri_tests + vertices * (2 * F_BACK + F_SHADE)


def algorithmic_bytes(c, pixels):
    """Bytes one frame needs by the reference's own data layout, from the same counters."""
    return B_BOX * c.box_tests + B_TRI * c.tri_tests + B_HIT * c.shadow_rays + B_PIXEL * pixels


def measured_counters(workload, kernel):
    """The newest committed counter summary of the dominant kernel on this workload (rocprofv3 --pmc passes of this
    same command, see tests/prof_round.sh and tests/prof_sq.sh) -> (memory-side bytes per launch, issue figures,
    file name); (None, None, None) if there is none."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_{kernel}.json")), reverse=True):
        d = json.load(open(f))
        if d.get("workload") != workload:
            continue
        m = d["memory_side"]
        sq = d.get("sq", {})
        issue = None
        if sq.get("SQ_ACTIVE_INST_VALU") and sq.get("SQ_WAVE_CYCLES"):
            issue = {"lanes_per_valu_instruction": sq["SQ_THREAD_CYCLES_VALU"] / sq["SQ_ACTIVE_INST_VALU"],
                     "waves_waiting": sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"],
                     "valu_instructions_per_launch": sq["SQ_INSTS_VALU"] / d.get("launches_profiled", 1),
                     "salu_instructions_per_launch": sq["SQ_INSTS_SALU"] / d.get("launches_profiled", 1)}
        return (m["read_bytes"] + m["write_bytes"]) / d.get("launches_profiled", 1), issue, os.path.basename(f)
    return None, None, None


def cpu_baseline(hs, w, h, prm_gpu, pp, budget_s=12.0):
    """The CPU restatement of the reference (test oracle), timed on this host's cores on a bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C
    import oraclelib as O
    cores = int(os.environ.get("OMP_NUM_THREADS", 0)) or min(16, os.cpu_count() or 1)
    L = O.lib(fast=True)
    cam = abi.Camera.from_buffer_copy(hs.camera)
    prm = abi.RenderParams.from_buffer_copy(prm_gpu)
    prm.tile_rank, prm.tile_world = 0, 1
    note = ""
    if pp:
        # the serial emission loop and the kd-tree build of 10M photons take minutes on the CPU: the baseline frame
        # uses maps of a tenth of the size (the gather cost per lookup is nearly independent of the map size)
        small = abi.PhotonParams.from_buffer_copy(pp)
        small.n_global, small.n_caustic = max(1, pp.n_global // 10), max(1, pp.n_caustic // 10)
        t0 = time.perf_counter()
        O.fill_photon_maps(hs, small)
        note = f"; photon maps of {small.n_global}+{small.n_caustic} photons filled and balanced single-threaded in {time.perf_counter() - t0:.1f} s (not in the rate)"
    fr = O.Frame(w, h)
    band = 16
    order = np.random.RandomState(0).permutation(h // band)
    rays, t0, bands = 0, time.perf_counter(), 0
    for b in order:
        st = O.Stats()
        L.orc_render(C.byref(hs.desc), C.byref(cam), C.byref(prm), C.byref(fr.out), cores, int(b) * band,
                     int(b) * band + band, C.byref(st))
        rays += st.closest + st.shadow
        bands += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": f"{bands} random {band}-row bands of the same {w}x{h} frame at {prm.min_spp}..{prm.max_spp} spp "
                      f"({rays} rays in {dt:.1f} s){note}"}


def cpu_baseline_reference(ctx, mesh, cores, width=960, height=540):
    """The reference renderer itself (oracle/_ref/ref_harness_time: the reference's sources compiled -O3 in the build
    container, its own thread pool, its compiled-in 128 spp and 50 bounces) on the same scene at a reduced frame
    size.  The reference does not count rays; the ray count is the GPU's for the same frame (same scene, camera,
    spp; path tracing statistics, so the counts agree to well under 1 %).  None if the binary is not there."""
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_harness_time")
    if not os.path.exists(exe):
        return None
    with tempfile.TemporaryDirectory() as tmp:
        scenes.write_obj(os.path.join(tmp, "mesh.obj"), mesh)
        xml = scenes.mesh_scene_xml("mesh.obj", width, height)
        with open(os.path.join(tmp, "scene.xml"), "w") as fp:
            fp.write(xml)
        try:
            r = subprocess.run([exe, "scene.xml", tmp, "time", str(cores)], cwd=tmp, capture_output=True, text=True, timeout=600)
            ms = json.loads(r.stdout.strip().splitlines()[-1])["ms"]
        except Exception as e:  # an illegal instruction on an older host CPU, a timeout: fall back to the port
            print(f"reference baseline not available: {e}", file=sys.stderr)
            return None
    hs = lib.HostScene.from_string(xml)
    hs.set_mesh("mesh.obj", mesh["v"], mesh["vt"], mesh["f"])
    ctx.set_camera(hs.camera)
    ctx.render_device(abi.default_params(min_spp=128, max_spp=128, seed=7))
    c = ctx.counters()
    rays = c.closest_rays + c.shadow_rays
    return {"value": rays / ms / 1e3, "unit": "Mrays/s", "cores": cores, "kind": "reference",
            "sample": f"the reference's own renderer (its thread pool with {cores} threads, 128 spp, 50 bounces) on the same scene at "
                      f"{width}x{height}: {ms / 1e3:.1f} s for {rays} rays (ray count taken from the GPU's frame of the same size)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="mesh_1m_1080p_64spp", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--max-bounces", type=int, default=None, help="development only: the headline number uses 50")
    ap.add_argument("--tile-size", type=int, default=None, help="development only: side of the tiles dealt to the ranks")
    ap.add_argument("--shard", default=None, metavar="R/W",
                    help="development only: one GPU renders the tiles rank R of W would get (no gather): per-rank time of a W-GPU run")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started bare: this process (which has not touched a GPU: importing torch does not) starts one rank per GPU
        # as a child and hands on its exit code; the ranks inherit stdout, so rank 0's JSON line appears as usual
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        sys.exit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the two must agree")
    # RT_BENCH_REHEARSE=1: every rank on GPU 0, records exchanged over gloo through host memory - the N > 1 code path
    # (launch, sharding, pack, all-gather, unpack, reductions) on a box with one GPU.  Its line says so; not a measurement.
    rehearsal = os.environ.get("RT_BENCH_REHEARSE") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # RT_BENCH_GROUP=1 takes the N > 1 code path (process group, frame exchange, reductions) with a single rank
    grouped = world > 1 or os.environ.get("RT_BENCH_GROUP") == "1"
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    hs, w, h, (spp_min, spp_max), ntri, mode, bounces, pp = build_scene(args.workload)
    if args.max_bounces is not None:
        bounces = args.max_bounces
    headline = args.workload == "mesh_1m_1080p_64spp" and bounces == 50
    ctx = lib.Context(local)
    ctx.upload(hs.desc)
    ctx.set_camera(hs.camera)
    tile_rank, tile_world = (int(x) for x in args.shard.split("/")) if args.shard else (rank, world)
    headline = headline and not args.shard
    prm = abi.default_params(mode=mode, min_spp=spp_min, max_spp=spp_max, seed=7, tile_rank=tile_rank, tile_world=tile_world,
                             max_bounces=bounces, photon_flags=pp.flags if pp else 0, gather_radius=3.0)
    if args.tile_size:
        prm.tile_size = args.tile_size
    # two sets of planes and host images, used in turn: the copy of a finished frame to the host (queued on torch's stream)
    # runs while the library, on its own stream, renders the next frame into the other set
    fbs = [FrameBuffers(w, h, dev), FrameBuffers(w, h, dev)]
    fb = fbs[0]
    gather = FrameGather(w, h, prm.tile_size, rank, world, dev, ctx=ctx, exchange=grouped, via_host=rehearsal)
    filler = PhotonFill(ctx, rank, world, dev, via_host=rehearsal) if pp else None
    host_rgbs = [torch.empty((w * h, 3), dtype=torch.uint8).pin_memory() for _ in fbs]
    copied = [torch.cuda.Event() for _ in fbs]
    photon_stats = {}
    frames = [0]

    def step():
        k = frames[0] % 2
        frames[0] += 1
        f = fbs[k]
        copied[k].synchronize()  # the copy out of this set of planes, two frames ago, before they are written again
        if filler:  # the maps are part of the frame: emitted by all ranks, gathered, balanced on every rank
            torch.cuda.current_stream().synchronize()
            photon_stats.update(filler(pp))
        ctx.render_device(prm, rgb=f.rgb.data_ptr(), z=f.z.data_ptr(), sample_count=f.sample_count.data_ptr(),
                          object_id=f.object_id.data_ptr())
        gather(f)
        if rank == 0:
            host_rgbs[k].copy_(f.rgb, non_blocking=True)
            copied[k].record()

    def fence():
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    # one counting frame (separately compiled kernel variant) for the visit counts the roofline needs
    ctx.set_counting(True)
    step()
    counted = ctx.counters()
    ctx.set_counting(False)
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    kernel_ms, launches = 0.0, 0
    for _ in range(args.steps):
        step()
        c = ctx.counters()
        kernel_ms += c.trace_kernel_ms
        launches += c.trace_launches
    fence()
    dt = time.perf_counter() - t0

    exchange_ms = None
    if args.shard and tile_world > 1:
        # what the frame exchange costs a rank beside the all-gather itself: packing its records and unpacking everybody's
        g = FrameGather(w, h, prm.tile_size, tile_rank, tile_world, dev, ctx=ctx)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(20):
            g.pack(fb)
            g.unpack(fb)
        exchange_ms = (time.perf_counter() - t1) / 20 * 1e3
    c = ctx.counters()
    photon_rays = photon_stats.get("rays", 0)
    stats = torch.tensor([c.closest_rays, c.shadow_rays, c.camera_paths, counted.box_tests, counted.tri_tests,
                          c.gathers, counted.photon_visits, photon_rays], dtype=torch.float64, device="cpu" if rehearsal else dev)
    tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else dev)
    if grouped:
        dist.all_reduce(stats)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    rays_per_frame = float(stats[0] + stats[1] + stats[7])

    if rank == 0:
        n_launch = max(1, counted.trace_launches)
        kernel = "k_rays" if counted.trace_pipeline == 1 else {abi.RT_MODE_PATH: "k_paths", abi.RT_MODE_WHITTED: "k_whitted"}[mode]
        flops_launch = algorithmic_flops(hs.desc, counted, mode) / n_launch
        bytes_launch = (algorithmic_bytes(counted, w * h // world) + B_PHOTON * counted.photon_visits) / n_launch
        avg_launch_s = kernel_ms / max(1, launches) / 1e3
        tflops = flops_launch / avg_launch_s / 1e12
        gbps = bytes_launch / avg_launch_s / 1e9
        traffic, issue, traffic_src = measured_counters(args.workload, kernel) if world == 1 else (None, None, None)
        valu = {"achieved": tflops, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": tflops / PEAK_FP32_TFLOPS,
                "algorithmic_gflop_per_launch": flops_launch / 1e9}
        hbm = {"achieved": gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": gbps / PEAK_HBM_GBPS,
               "algorithmic_gbyte_per_launch": bytes_launch / 1e9}
        # The contract's figure is algorithmic bytes (or flops) over time; `bound` says what the kernel is in fact held by.
        if ntri > 0:
            first, second, second_key, bound = hbm, valu, "valu_fp32", "valu_issue"
            note = ("k_rays, rank 0's launches. `achieved` is the contract's figure: the bytes of the REFERENCE's walk (28 B per box "
                    "test and 52 B per triangle test of its binary BVH, counted by the counting variant of the kernel, which walks "
                    "that tree) over the launch time of the timed kernel - which walks a four-wide tree built on the GPU and needs "
                    "fewer bytes for the same rays, so the figure can pass the HBM peak and is no bound on this kernel. `traffic` is "
                    "what the memory side saw per launch (counter passes of this command); the BVH and triangle pools stay in the "
                    "256 MiB Infinity Cache, so even that is fabric traffic, not DRAM reads. The kernel is held by instruction issue: "
                    "waves run partly filled (`issue.lanes_per_valu_instruction` of 64) and wait for dependent node fetches "
                    "(`issue.waves_waiting`).")
        elif pp is not None:
            first, second, second_key, bound = hbm, valu, "valu_fp32", "latency"
            note = ("k_whitted with photon lookups, rank 0's launches. Algorithmic bytes: 32 B per kd-tree node a lookup visits (the "
                    "reference's visits: same tree, same order) plus the frame's ray bytes. A lookup is a chain of dependent 32-byte "
                    "fetches whose order the reference's heap fixes; lanes of a wave are at unrelated nodes of unrelated lookups, so "
                    "the kernel is held by that chain's latency at a few active lanes per instruction, far from HBM's rate.")
        elif mode == abi.RT_MODE_WHITTED:
            first, second, second_key, bound = valu, hbm, "hbm", "valu_fp32"
            note = ("k_whitted, one launch per frame. Algorithmic flops price every ray against every object of the scene, as the "
                    "reference does it (hand counts per ray-node transform and primitive test); the kernel itself walks a hierarchy "
                    "over the objects' boxes and tests between one and two of the 66 objects per ray (measured with a diagnostic build "
                    "that counts them: with such a library `executed` holds the figure priced on the object tests made; the "
                    "hierarchy's own box tests are not priced). The scene's records come through the scalar "
                    "cache; HBM sees the frame planes only.")
        else:
            first, second, second_key, bound = valu, hbm, "hbm", "valu_fp32"
            note = ("k_paths, one launch per frame: the one-kernel form of the path tracer for scenes without meshes. Algorithmic "
                    "flops price every ray against every object, as the reference does it (scenes of sixteen objects and more are walked "
                    "through a hierarchy over the objects' boxes: `executed`, from a diagnostic build, prices the object tests made); un-fused FP32 (the code is built without "
                    "contraction so that its results are the reference's bit for bit) can reach a quarter of the packed-FMA peak "
                    "the fraction is taken against. The scene's records come through the scalar cache; HBM sees the sample buffer.")
        roof = dict(first, bound=bound, traffic=traffic, traffic_source=traffic_src,
                    kernel=kernel, avg_launch_ms=avg_launch_s * 1e3, launches_per_frame=n_launch,
                    kernel_ms_per_frame=kernel_ms / args.steps,
                    box_tests_per_frame=int(stats[3].item()), tri_tests_per_frame=int(stats[4].item()),
                    note=note, **{second_key: second})
        if traffic:
            roof["traffic_gbps"] = traffic / avg_launch_s / 1e9
        if issue:
            roof["issue"] = issue
        if c.phase_cycles[10] and ntri == 0:  # a build that counts the object tests its scene walks make (the last timed frame's)
            ex = algorithmic_flops(hs.desc, c, mode, object_tests=c.phase_cycles[10]) / n_launch
            roof["executed"] = {"object_tests_per_ray": c.phase_cycles[10] / max(1, c.closest_rays + c.shadow_rays),
                                "gflop_per_launch": ex / 1e9, "achieved": ex / avg_launch_s / 1e12, "unit": "TFLOP/s",
                                "frac": ex / avg_launch_s / 1e12 / PEAK_FP32_TFLOPS}
        out = {
            "metric": METRICS[args.workload],
            "value": rays_per_frame * args.steps / dt / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" if not rehearsal else "synthetic; REHEARSAL: all ranks share GPU 0 and exchange over gloo - not a measurement",
            "config": {"workload": args.workload, "triangles": ntri, "width": w, "height": h, "spp": [spp_min, spp_max],
                       "max_bounces": prm.max_bounces, "mode": "path" if mode == abi.RT_MODE_PATH else "whitted",
                       "parallelism": f"tiles/{world}", "rays_per_frame": int(rays_per_frame),
                       "camera_paths_per_frame": int(stats[2].item())},
            "roofline": roof,
        }
        if pp:
            out["photons"] = {"global": pp.n_global, "caustics": pp.n_caustic, "paths": photon_stats.get("paths"),
                              "photon_rays_per_frame": int(stats[7].item()), "emit_ms": photon_stats.get("emit_ms"),
                              "gather_ms": photon_stats.get("gather_ms"), "build_ms": photon_stats.get("build_ms"),
                              "gathers_per_frame": int(stats[5].item()), "node_visits_per_frame": int(stats[6].item()),
                              "gathers_per_s": float(stats[5].item()) * args.steps / dt}
        if exchange_ms is not None:
            out["frame_exchange_pack_unpack_ms"] = exchange_ms
        if any(c.phase_cycles):
            out["phase_cycles"] = list(c.phase_cycles)
        if world == 1 and not args.no_cpu_baseline:
            port = cpu_baseline(hs, w, h, prm, pp)
            ref = cpu_baseline_reference(ctx, MESHES[args.workload], port["cores"]) if args.workload in MESHES and mode == abi.RT_MODE_PATH and ntri <= 2_000_000 else None
            out["cpu_baseline"] = ref or port
            if ref:
                out["cpu_baseline_port"] = port  # our CPU restatement of the same algorithm (the parity oracle), -O3, same cores
        print(json.dumps(out))
    if grouped:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
