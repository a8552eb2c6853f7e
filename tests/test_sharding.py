"""N > 1 path on CPU: world_size-2 (and 3, uneven rows) `gloo` runs of the row
sharding + framebuffer gather that bench.py uses with RCCL on GPUs."""
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


def _worker(rank, world, port, height, width, out):
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
        r0, r1 = sharding.shard_rows(height, world, rank)
        assert sharding.shard_rngs(rngs, width, (r0, r1))[0, 0] == 2 * r0 * width
        local = torch.from_numpy(full[r0 * width:r1 * width].copy())
        fg = sharding.FrameGather(dist, width, height, 4, "cpu")
        frame = fg.frame(local)
        ok = torch.equal(frame, torch.from_numpy(full))
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


@pytest.mark.parametrize("world,height,width", [(2, 32, 32), (2, 9, 16), (3, 10, 8)])
def test_row_sharded_gather_gloo(tmp_path, world, height, width):
    import torch.multiprocessing as mp
    out = str(tmp_path / "result.txt")
    port = 29500 + (os.getpid() + world * 7 + height) % 2000
    mp.spawn(_worker, args=(world, port, height, width, out), nprocs=world, join=True)
    assert open(out).read() == "ok"
