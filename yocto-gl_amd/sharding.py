"""Row sharding of trace_state across ranks and the one exchange
step of the path: the framebuffer gather.

Every pixel owns its PCG stream and its accumulators (yocto_trace.cpp:1466-1491),
so image rows — the reference's own unit of parallel work (yocto_trace.cpp:66-69)
— shard with no data-path collective: rank g renders rows [r0, r1) with a full
replica of scene + BVH + lights and the matching slice of the serially seeded
rngs.  Only the finished rows travel: one all-gather per batch (RCCL over xGMI
on GPUs — torch.distributed backend "nccl" — or gloo on CPU for the tests).
"""
import numpy as np


def shard_rows(height, world, rank):
    """Contiguous, near-equal row blocks: the first height % world ranks get one
    extra row."""
    base, rem = divmod(height, world)
    r0 = rank * base + min(rank, rem)
    return r0, r0 + base + (1 if rank < rem else 0)


def shard_rngs(rngs, width, rows):
    """Slice of make_trace_state's per-pixel seeds (computed once, serially, for
    the whole frame: yocto_trace.cpp:1512-1515) for rows [r0, r1)."""
    r0, r1 = rows
    return np.ascontiguousarray(rngs[r0 * width:r1 * width])


class FrameGather:
    """All-gather of per-rank row blocks into the full frame on every rank.

    `channels` floats per pixel (4 for trace_state.image).  Ranks may hold
    different row counts (height % world != 0): blocks are padded to the largest
    count for the collective and trimmed when unpacked, so one
    all_gather_into_tensor call moves everything."""

    def __init__(self, dist, width, height, channels, device, dtype=None):
        import torch
        self.dist, self.torch = dist, torch
        self.world = dist.get_world_size() if dist is not None else 1
        self.rank = dist.get_rank() if dist is not None else 0
        self.width, self.height, self.channels = width, height, channels
        self.rows = [shard_rows(height, self.world, r) for r in range(self.world)]
        self.max_rows = max(r1 - r0 for r0, r1 in self.rows)
        dtype = dtype or torch.float32
        n = self.max_rows * width
        self.even = all(r1 - r0 == self.max_rows for r0, r1 in self.rows)
        self.staging = None if self.even else torch.zeros(n, channels, device=device, dtype=dtype)
        self.packed = torch.empty(self.world * n, channels, device=device, dtype=dtype)

    def gather(self, local):
        """local: [rows_of_this_rank * width, channels] tensor.  Returns the
        padded gather buffer (use frame() to unpack)."""
        if self.world == 1:
            return local
        src = local
        if not self.even:
            self.staging[:local.shape[0]].copy_(local)
            src = self.staging
        self.dist.all_gather_into_tensor(self.packed, src)
        return self.packed

    def frame(self, local):
        """Gather + unpack into a [height * width, channels] tensor."""
        if self.world == 1:
            return local
        packed = self.gather(local).view(self.world, self.max_rows * self.width, self.channels)
        if self.even:
            return packed.reshape(self.height * self.width, self.channels)
        parts = [packed[r, :(r1 - r0) * self.width] for r, (r0, r1) in enumerate(self.rows)]
        return self.torch.cat(parts, 0)
