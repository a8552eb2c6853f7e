"""Sharding of trace_state across ranks and the one exchange step
of the path: the framebuffer gather.

Every pixel owns its PCG stream and its accumulators (yocto_trace.cpp:1466-1491),
so pixels shard with no data-path collective: a rank renders its slice with a
full replica of scene + BVH + lights and the matching slice of the serially
seeded rngs.  Only the finished pixels travel: one all-gather per batch (RCCL
over xGMI on GPUs — torch.distributed backend "nccl" — or gloo on CPU for the
tests) followed by a device-side un-permute into the frame's row-major order.

Two slicings:

  "rows"     contiguous, near-equal row blocks — the reference's own unit of
             parallel work (yocto_trace.cpp:66-69).  Cheap to unpack, but the
             cost of a row is not uniform: on BASELINE configs[1] the top 40 % of
             the frame is sky (one miss per sample) and the bottom is the
             1M-triangle plane (eight bounces), so the ranks owning sky idle.
  "columns"  (default) 16-pixel tile columns dealt round-robin: rank r of G owns
             tile columns r, r + G, r + 2G, ... over the full height
             (ythip_state_create_striped).  Every rank sees the same mix of sky
             and ground; 1280 / 16 = 80 columns divide evenly by 2, 4 and 8.
"""
import numpy as np

TILE = 16  # yt_kernels.h YT_TILE


def shard_rows(height, world, rank):
    """Contiguous, near-equal row blocks: the first height % world ranks get one
    extra row."""
    base, rem = divmod(height, world)
    r0 = rank * base + min(rank, rem)
    return r0, r0 + base + (1 if rank < rem else 0)


def slice_columns(width, col_first=0, col_stride=1):
    """Global x of every pixel column of a column-striped slice, in local order."""
    cols = [np.arange(c * TILE, min(width, (c + 1) * TILE))
            for c in range(col_first, (width + TILE - 1) // TILE, col_stride)]
    return np.concatenate(cols) if cols else np.zeros(0, np.int64)


class Shard:
    """The slice of a width x height frame one rank renders: rows [r0, r1) x the
    tile columns cols = (first, stride).  `pixels` lists the frame's row-major
    pixel index of every local pixel, in the slice's own row-major order."""

    def __init__(self, width, height, rows, cols=(0, 1)):
        self.width, self.height = width, height
        self.rows, self.cols = tuple(rows), tuple(cols)
        self.xs = slice_columns(width, *self.cols)
        self.local_width = len(self.xs)
        r0, r1 = self.rows
        self.pixels = (np.arange(r0, r1, dtype=np.int64)[:, None] * width
                       + self.xs[None, :]).reshape(-1)

    @property
    def npixels(self):
        return len(self.pixels)

    def take(self, frame_array):
        """Slice of a per-pixel frame array [height * width, ...] (e.g. the rngs of
        make_trace_state, computed once, serially, for the whole frame:
        yocto_trace.cpp:1512-1515)."""
        return np.ascontiguousarray(np.asarray(frame_array)[self.pixels])


def shard_frame(width, height, world, rank, mode="columns"):
    if mode == "rows":
        return Shard(width, height, shard_rows(height, world, rank))
    if mode == "columns":
        ncols = (width + TILE - 1) // TILE
        if world > ncols:  # more ranks than tile columns: fall back to row blocks
            return Shard(width, height, shard_rows(height, world, rank))
        return Shard(width, height, (0, height), (rank, world))
    raise ValueError(f"unknown sharding mode {mode!r}")


def shard_rngs(rngs, width, rows):
    """Row-block slice of the per-pixel seeds (kept for the row mode's callers)."""
    r0, r1 = rows
    return np.ascontiguousarray(rngs[r0 * width:r1 * width])


class FrameGather:
    """All-gather of the per-rank slices into the full frame on every rank.

    `channels` values per pixel (4 for trace_state.image).  Ranks may hold
    different pixel counts: slices are padded to the largest for the collective,
    so ONE all_gather_into_tensor call moves everything, and one index_select
    with a precomputed permutation puts the pixels in the frame's row-major order
    (a 14.7 MB copy at 720p)."""

    def __init__(self, dist, width, height, channels, device, dtype=None, mode="columns",
                 shards=None, always=False):
        import torch
        self.dist, self.torch = dist, torch
        self.world = dist.get_world_size() if dist is not None else 1
        self.rank = dist.get_rank() if dist is not None else 0
        self.width, self.height, self.channels = width, height, channels
        self.shards = shards or [shard_frame(width, height, self.world, r, mode)
                                 for r in range(self.world)]
        self.shard = self.shards[self.rank]
        self.always = always  # run the collective + un-permute even for one rank (rehearsal)
        self.max_n = max(s.npixels for s in self.shards)
        dtype = dtype or torch.float32
        self.even = all(s.npixels == self.max_n for s in self.shards)
        self.staging = None if self.even else torch.zeros(self.max_n, channels, device=device, dtype=dtype)
        self.packed = torch.empty(self.world * self.max_n, channels, device=device, dtype=dtype)
        # perm[g] = position of frame pixel g in the packed gather buffer
        perm = np.full(width * height, -1, np.int64)
        for r, s in enumerate(self.shards):
            perm[s.pixels] = r * self.max_n + np.arange(s.npixels)
        if (perm < 0).any():
            raise ValueError("shards do not cover the frame")
        self.identity = bool((perm == np.arange(width * height)).all())
        self.perm = torch.from_numpy(perm).to(device)
        self.out = torch.empty(width * height, channels, device=device, dtype=dtype)
        # Column stripes of equal width over the full height (what bench.py uses): the
        # un-permute is a transposition of whole 16-pixel runs — packed [rank, row, col, 16]
        # → frame [row, col, rank, 16] — i.e. one strided copy, no index array to read.
        s0 = self.shards[0]
        self.blocked = None
        if ((self.world > 1 or always) and self.even and width % (TILE * self.world) == 0 and
                all(s.rows == (0, height) and s.cols == (r, self.world) for r, s in enumerate(self.shards))):
            self.blocked = (height, s0.local_width // TILE)

    def gather(self, local):
        """local: [npixels of this rank, channels] tensor.  Returns the padded
        gather buffer (rank-major)."""
        if self.world == 1 and not self.always:
            return local
        src = local
        if not self.even:
            self.staging[:local.shape[0]].copy_(local)
            src = self.staging
        self.dist.all_gather_into_tensor(self.packed, src)
        return self.packed

    def frame(self, local):
        """Gather + un-permute into a [height * width, channels] tensor."""
        if self.world == 1 and self.identity and not self.always:
            return local
        packed = self.gather(local)
        if self.identity and not self.always:  # row blocks: only the tail padding to drop
            return packed[:self.width * self.height]
        if self.blocked is not None:
            h, c = self.blocked
            self.out.view(h, c, self.world, TILE, self.channels).copy_(
                packed.view(self.world, h, c, TILE, self.channels).permute(1, 2, 0, 3, 4))
            return self.out
        self.torch.index_select(packed, 0, self.perm, out=self.out)
        return self.out
