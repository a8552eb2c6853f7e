"""N > 1 path on CPU: world_size-2 (and 3, uneven slices) `gloo` runs of the
sharding (row blocks and round-robin tile columns) + framebuffer gather that
bench.py uses with RCCL on GPUs."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "yocto-gl_amd"))

import sharding  # noqa: E402


def test_shard_rows_partition_every_row_once():
    for h in [1, 2, 7, 90, 720, 1080]:
        for w in [1, 2, 3, 4, 8]:
            rows = [sharding.shard_rows(h, w, r) for r in range(w)]
            assert rows[0][0] == 0 and rows[-1][1] == h
            assert all(a[1] == b[0] for a, b in zip(rows, rows[1:]))
            sizes = [b - a for a, b in rows]
            assert max(sizes) - min(sizes) <= 1


def test_shard_frame_covers_every_pixel_once():
    for mode in ["rows", "columns"]:
        for (w, h) in [(1, 1), (16, 9), (33, 7), (100, 40), (1280, 720), (1920, 1080)]:
            for world in [1, 2, 3, 4, 8]:
                shards = [sharding.shard_frame(w, h, world, r, mode) for r in range(world)]
                allpix = np.concatenate([s.pixels for s in shards])
                assert len(allpix) == w * h and len(np.unique(allpix)) == w * h
                assert all(s.npixels == s.local_width * (s.rows[1] - s.rows[0]) for s in shards)


def test_column_shards_balance_the_baseline_frame():
    # 1280 / 16 = 80 tile columns: 2, 4 and 8 ranks get identical pixel counts
    for world in [2, 4, 8]:
        n = [sharding.shard_frame(1280, 720, world, r).npixels for r in range(world)]
        assert len(set(n)) == 1 and sum(n) == 1280 * 720
    # and each rank's slice spans the full height (sky and ground alike)
    s = sharding.shard_frame(1280, 720, 8, 3)
    assert s.rows == (0, 720) and s.cols == (3, 8) and s.local_width == 160
    assert s.xs[:17].tolist() == list(range(48, 64)) + [48 + 128]


def test_weak_scaling_frame_keeps_per_rank_work_fixed():
    # bench.py's primary N > 1 line ("scaling": "weak"): the configs[1] camera at N x the
    # pixels; every rank owns the same number of tile columns and (to 2 %) the
    # pixel count of the 1280x720 single-GPU frame
    sys.path.insert(0, os.path.join(HERE, ".."))
    import bench
    assert [bench.weak_resolution(1280, n) for n in (1, 2, 4, 8)] == [1280, 1792, 2560, 3584]
    for world in [1, 2, 3, 4, 8]:
        w = bench.weak_resolution(1280, world)
        h = int(round(w * 9 / 16))  # image-size rule for the 16:9 camera
        assert w % (16 * world) == 0
        n = [sharding.shard_frame(w, h, world, r).npixels for r in range(world)]
        assert len(set(n)) == 1 and sum(n) == w * h
        assert abs(n[0] / (1280 * 720) - 1) < 0.03


def test_local_width_matches_library():
    sys.path.insert(0, os.path.join(HERE, "..", "yocto-gl_amd"))
    import ythip as yt
    lib = yt.load_library()
    for w in [1, 15, 16, 17, 100, 1280, 1283]:
        for stride in [1, 2, 3, 8]:
            for first in range(stride):
                assert lib.ythip_state_local_width(w, first, stride) == \
                    len(sharding.slice_columns(w, first, stride))
    assert lib.ythip_state_local_width(0, 0, 1) == -1
    assert lib.ythip_state_local_width(16, 0, 0) == -1


def _worker(rank, world, port, height, width, mode, out):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # the "render": a golden full frame; each rank only ever sees its rows
        g = np.load(os.path.join(HERE, "golden", "render_cornellbox_path.npz"))
        full = np.tile(g["image"].reshape(-1, 4), (8, 1))[:height * width].copy()
        full += np.arange(height * width, dtype="f4")[:, None]  # make every pixel unique
        rngs = np.arange(2 * height * width, dtype="u8").reshape(-1, 2)
        shard = sharding.shard_frame(width, height, world, rank, mode)
        assert shard.take(rngs)[0, 0] == 2 * shard.pixels[0]
        local = torch.from_numpy(shard.take(full))
        fg = sharding.FrameGather(dist, width, height, 4, "cpu", mode=mode)
        frame = fg.frame(local)
        ok = torch.equal(frame, torch.from_numpy(full))
        if mode == "columns" and width % (16 * world) == 0:
            ok = ok and fg.blocked is not None  # even stripes take the strided-copy un-permute
            fg.blocked = None                   # ... which must agree with the index_select one
            ok = ok and torch.equal(fg.frame(local), torch.from_numpy(full))
        # the timing contract of bench.py: barrier, then MAX over ranks
        dist.barrier()
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ok = ok and t.item() == float(world)
        if rank == 0:
            with open(out, "w") as f:
                f.write("ok" if ok else "mismatch")
        else:
            assert ok
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,height,width,mode", [
    (2, 32, 32, "rows"), (2, 9, 16, "rows"), (3, 10, 8, "rows"),
    (2, 12, 64, "columns"), (2, 9, 40, "columns"), (3, 10, 100, "columns"),
    (2, 8, 96, "columns"), (4, 6, 128, "columns"), (3, 7, 144, "columns"),  # even stripes: the blocked un-permute
    (3, 10, 8, "columns")])  # fewer tile columns than ranks: falls back to rows
def test_sharded_gather_gloo(tmp_path, world, height, width, mode):
    import torch.multiprocessing as mp
    out = str(tmp_path / "result.txt")
    port = 29500 + (os.getpid() + world * 7 + height + width) % 2000
    mp.spawn(_worker, args=(world, port, height, width, mode, out), nprocs=world, join=True)
    assert open(out).read() == "ok"
