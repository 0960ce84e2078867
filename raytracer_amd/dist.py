"""Multi-GPU frame assembly: one process per GPU, image tiles dealt round-robin.

The reference hands 16x16 tiles to threads from one atomic counter
(src/main.cpp:93,276-284).  Across GPUs the same tiles are dealt statically:
rank r renders the tiles t with t % world == r (fine interleaving evens out
the cost of glass / mesh regions).  Pixels never interact, so the only
exchange step is the final framebuffer gather: every rank packs the pixels it
owns into one byte buffer and a single all_gather (RCCL over xGMI with the
"nccl" backend, gloo on CPU) puts the whole frame on every rank.
"""
import numpy as np
import torch
import torch.distributed as dist

# bytes per pixel in the exchange buffer: one record of four 32-bit words (r | g << 8 | b << 16, z, sample count, object id)
PACK = 16


def shard_pixels(width, height, tile_size, rank, world):
    """Pixel indices (row-major) of the tiles owned by `rank`, in tile order - the
    same list, in the same order, the library renders for that shard."""
    tx = (width + tile_size - 1) // tile_size
    ty = (height + tile_size - 1) // tile_size
    out = []
    for t in range(rank, tx * ty, world):
        x0, y0 = (t % tx) * tile_size, (t // tx) * tile_size
        xs = np.arange(x0, min(x0 + tile_size, width))
        ys = np.arange(y0, min(y0 + tile_size, height))
        out.append((ys[:, None] * width + xs[None, :]).reshape(-1))
    return np.concatenate(out) if out else np.zeros(0, np.int64)


def frame_records(width, height, tile_size, world):
    """Records per rank in the exchange buffer: every tile takes tile_size^2 (rt_frame_records)."""
    tiles = -(-width // tile_size) * -(-height // tile_size)
    return -(-tiles // world) * tile_size * tile_size


def record_pixels(width, height, tile_size, rank, world):
    """(records, pixels): which record of `rank`'s exchange array holds which pixel (the arithmetic of k_frame_pack)."""
    tx = -(-width // tile_size)
    ty = -(-height // tile_size)
    per = tile_size * tile_size
    s = np.arange(frame_records(width, height, tile_size, world), dtype=np.int64)
    t = rank + (s // per) * world
    o = s % per
    x = (t % tx) * tile_size + o % tile_size
    y = (t // tx) * tile_size + o // tile_size
    ok = (t < tx * ty) & (x < width) & (y < height)
    return s[ok], (y * width + x)[ok]


def all_gather_flat(recv, send, via_host=False):
    """all_gather_into_tensor of flat views; via_host stages both sides through host memory (a gloo group whose
    ranks keep their tensors on a GPU: the rehearsal of the multi-GPU path on one GPU)."""
    if not via_host or not send.is_cuda:
        dist.all_gather_into_tensor(recv.view(-1), send.view(-1))
        return
    r = torch.empty(recv.numel(), dtype=recv.dtype)
    dist.all_gather_into_tensor(r, send.reshape(-1).cpu())
    recv.view(-1).copy_(r)


class FrameBuffers:
    """Full-frame planes in the memory of `device` (torch tensors; the library writes through data_ptr())."""

    def __init__(self, width, height, device):
        n = width * height
        self.width, self.height = width, height
        self.rgb = torch.zeros((n, 3), dtype=torch.uint8, device=device)
        self.z = torch.zeros((n,), dtype=torch.float32, device=device)
        self.sample_count = torch.zeros((n,), dtype=torch.int32, device=device)
        self.object_id = torch.zeros((n,), dtype=torch.int32, device=device)

    def pointers(self):
        return (self.rgb.data_ptr(), self.z.data_ptr(), self.sample_count.data_ptr(), self.object_id.data_ptr())


class FrameGather:
    """Packs this rank's pixels into 16-byte records, all-gathers, unpacks every other rank's records into the full frame.

    With a library context (GPU) the two steps are one kernel each (rt_frame_pack / rt_frame_unpack: the place of a
    pixel in the exchange array is arithmetic, no index lists); without one (the CPU tests of the N > 1 path) the same
    record layout is produced with torch indexing."""

    def __init__(self, width, height, tile_size, rank, world, device, ctx=None, exchange=None, via_host=False):
        self.world, self.rank, self.ctx, self.tile, self.via_host = world, rank, ctx, tile_size, via_host
        self.n = frame_records(width, height, tile_size, world)
        # a single rank has nothing to exchange; exchange=True makes it go through pack / all_gather / unpack all the same
        self.exchange = world > 1 if exchange is None else bool(exchange)
        if not self.exchange:
            return
        self.send = torch.zeros((self.n, 4), dtype=torch.int32, device=device)
        self.recv = torch.zeros((world, self.n, 4), dtype=torch.int32, device=device)
        if ctx is None:
            s, px = record_pixels(width, height, tile_size, rank, world)
            self.my_s, self.my_px = torch.from_numpy(s).to(device), torch.from_numpy(px).to(device)
            rows, dest = [], []
            for r in range(world):
                if r != rank:
                    s, px = record_pixels(width, height, tile_size, r, world)
                    rows.append(r * self.n + s)
                    dest.append(px)
            self.rows = torch.from_numpy(np.concatenate(rows)).to(device)
            self.dest = torch.from_numpy(np.concatenate(dest)).to(device)

    def pack(self, fb):
        if self.ctx is not None:
            self.ctx.frame_pack(fb.pointers(), self.tile, self.rank, self.world, self.send.data_ptr())
            return
        s, px = self.my_s, self.my_px
        rgb = fb.rgb[px].to(torch.int32)
        self.send.zero_()
        self.send[s, 0] = rgb[:, 0] | (rgb[:, 1] << 8) | (rgb[:, 2] << 16)
        self.send[s, 1] = fb.z[px].view(torch.int32)
        self.send[s, 2] = fb.sample_count[px]
        self.send[s, 3] = fb.object_id[px]

    def unpack(self, fb):
        if self.ctx is not None:
            self.ctx.frame_unpack(self.recv.data_ptr(), self.tile, self.rank, self.world, fb.pointers())
            return
        blk = self.recv.view(-1, 4)[self.rows]
        fb.rgb[self.dest] = torch.stack([blk[:, 0] & 255, (blk[:, 0] >> 8) & 255, (blk[:, 0] >> 16) & 255], 1).to(torch.uint8)
        fb.z[self.dest] = blk[:, 1].contiguous().view(torch.float32)
        fb.sample_count[self.dest] = blk[:, 2]
        fb.object_id[self.dest] = blk[:, 3]

    def __call__(self, fb):
        if not self.exchange:
            return fb
        self.pack(fb)  # on the library's stream, finished when it returns
        all_gather_flat(self.recv, self.send, self.via_host)
        if self.ctx is not None:
            torch.cuda.current_stream().synchronize()  # the unpack kernel runs on the library's stream
        self.unpack(fb)
        return fb


class PhotonFill:
    """Fills the photon maps with every rank tracing a share of the photon paths.

    The reference fills a map with one serial loop: photon paths run one after the other until the map is
    full (project12.zip src/main.cpp:292-470).  With one random stream per path the photons a path stores
    do not depend on who traces it, so a round of paths is cut into `world` contiguous blocks, rank r
    traces block r, and the blocks' photons - concatenated in rank order, which is path order - are what
    the serial loop would have stored.  Per round: one all_gather of the ranks' photon counts and one of
    their 32-byte records (padded to the largest count); no other exchange.  Every rank ends up with the
    whole list, cuts it at the map's capacity and balances its own copy of the kd-tree.

    `emit(pp, which, path_begin, path_count, capacity) -> (uint8 tensor [m, 32] on `device`, stored)` and
    `build(which, uint8 tensor [n, 32])` default to the GPU library; tests pass CPU stand-ins.
    """

    FIRST_ROUND = 1 << 16

    def __init__(self, ctx, rank, world, device, emit=None, build=None, via_host=False):
        self.ctx, self.rank, self.world, self.device, self.via_host = ctx, rank, world, device, via_host
        self.emit = emit or self._emit_gpu
        self.build = build or self._build_gpu
        self._scratch = None

    def _emit_gpu(self, pp, which, path_begin, path_count, capacity):
        if self._scratch is None or self._scratch.shape[0] < capacity:
            self._scratch = torch.empty((max(1, capacity), 32), dtype=torch.uint8, device=self.device)
        # the library writes the scratch on its own stream: copies out of it that torch queued earlier must be through
        torch.cuda.current_stream().synchronize()
        _, stored = self.ctx.photons_emit(pp, which, path_begin, path_count, capacity, device_ptr=self._scratch.data_ptr())
        c = self.ctx.counters()
        self.stats["rays"] += int(c.closest_rays)
        self.stats["emit_ms"] += float(c.kernel_ms)
        return self._scratch[:min(stored, capacity)], stored

    def _build_gpu(self, which, records):
        torch.cuda.synchronize(self.device)
        self.ctx.photons_build(which, device_ptr=records.data_ptr(), n=records.shape[0])

    def __call__(self, pp):
        import time
        self.stats = {"rays": 0, "paths": 0, "emit_ms": 0.0, "gather_ms": 0.0, "build_ms": 0.0}
        world, rank = self.world, self.rank
        for which in (0, 1):
            cap = pp.n_caustic if which == 1 else pp.n_global
            enabled = (pp.flags & 4) if which == 1 else (pp.flags & (1 | 2 | 8))
            if not enabled or cap == 0:
                self.build(which, torch.empty((0, 32), dtype=torch.uint8, device=self.device))
                continue
            buf = torch.empty((cap, 32), dtype=torch.uint8, device=self.device)
            stored, path, rnd, barren = 0, 0, self.FIRST_ROUND, 0
            while stored < cap:
                block = -(-rnd // world)
                mine, n_mine = self.emit(pp, which, path + rank * block, block, cap - stored)
                n_mine = min(int(n_mine), cap - stored)
                t0 = time.perf_counter()
                if world > 1:
                    counts = torch.zeros(world, dtype=torch.int64, device=self.device)
                    me = torch.tensor([n_mine], dtype=torch.int64, device=self.device)
                    all_gather_flat(counts, me, self.via_host)
                    counts = counts.tolist()
                    most = max(1, max(counts))
                    send = torch.zeros((most, 32), dtype=torch.uint8, device=self.device)
                    send[:n_mine] = mine[:n_mine]
                    recv = torch.empty((world, most, 32), dtype=torch.uint8, device=self.device)
                    all_gather_flat(recv, send, self.via_host)
                    parts = [recv[r, :counts[r]] for r in range(world)]
                else:
                    counts, parts = [n_mine], [mine[:n_mine]]
                for part in parts:  # rank order = path order
                    take = min(part.shape[0], cap - stored)
                    buf[stored:stored + take] = part[:take]
                    stored += take
                self.stats["gather_ms"] += (time.perf_counter() - t0) * 1e3
                path += block * world
                got = sum(counts)
                # the next round asks for what is still missing at the yield seen so far, plus a little
                yield_ = max(stored, 1) / path
                rnd = int(min(1 << 24, max(self.FIRST_ROUND, 1.05 * (cap - stored) / yield_ + 1024)))
                barren = barren + 1 if got == 0 else 0
                if barren >= 8 and stored == 0:
                    raise RuntimeError(f"no photon stored after {path} paths: the scene has no surface that takes photons for this map")
            self.stats["paths"] += path
            t0 = time.perf_counter()
            self.build(which, buf)
            self.stats["build_ms"] += (time.perf_counter() - t0) * 1e3
        return self.stats
