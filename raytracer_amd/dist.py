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

# bytes per pixel in the packed exchange buffer: rgb u8*3, z f32, sample_count i32, object_id i32
PACK = 15


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


class FrameBuffers:
    """Full-frame planes in the memory of `device` (torch tensors; the library writes through data_ptr())."""

    def __init__(self, width, height, device):
        n = width * height
        self.width, self.height = width, height
        self.rgb = torch.zeros((n, 3), dtype=torch.uint8, device=device)
        self.z = torch.zeros((n,), dtype=torch.float32, device=device)
        self.sample_count = torch.zeros((n,), dtype=torch.int32, device=device)
        self.object_id = torch.zeros((n,), dtype=torch.int32, device=device)


class FrameGather:
    """Packs this rank's pixels, all-gathers, unpacks every rank's pixels into the full frame."""

    def __init__(self, width, height, tile_size, rank, world, device):
        self.world, self.rank = world, rank
        lists = [shard_pixels(width, height, tile_size, r, world) for r in range(world)]
        self.n_max = max(len(x) for x in lists)
        self.idx = [torch.from_numpy(x.astype(np.int64)).to(device) for x in lists]
        self.send = torch.zeros((self.n_max, PACK), dtype=torch.uint8, device=device)
        self.recv = torch.zeros((world, self.n_max, PACK), dtype=torch.uint8, device=device)

    def __call__(self, fb):
        if self.world == 1:
            return fb
        mine = self.idx[self.rank]
        n = len(mine)
        s = self.send
        s[:n, 0:3] = fb.rgb[mine]
        s[:n, 3:7] = fb.z[mine].view(torch.uint8).reshape(n, 4)
        s[:n, 7:11] = fb.sample_count[mine].view(torch.uint8).reshape(n, 4)
        s[:n, 11:15] = fb.object_id[mine].view(torch.uint8).reshape(n, 4)
        dist.all_gather_into_tensor(self.recv.view(-1), s.view(-1))
        for r in range(self.world):
            if r == self.rank:
                continue
            idx = self.idx[r]
            m = len(idx)
            blk = self.recv[r, :m]
            fb.rgb[idx] = blk[:, 0:3]
            fb.z[idx] = blk[:, 3:7].contiguous().view(torch.float32).reshape(m)
            fb.sample_count[idx] = blk[:, 7:11].contiguous().view(torch.int32).reshape(m)
            fb.object_id[idx] = blk[:, 11:15].contiguous().view(torch.int32).reshape(m)
        return fb
