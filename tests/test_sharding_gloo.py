"""CPU tests of the multi-rank host logic: band partition + gather over torch.distributed/gloo, world size 2 and 3."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gpuart_amd import sharding


def test_balanced_bands_properties():
    rng = np.random.RandomState(0)
    for world in (1, 2, 3, 4, 8):
        for height in (64, 1080, 2160, 4320):
            cost = rng.uniform(1, 10, height) * (np.arange(height) > height // 3)
            bands = sharding.balanced_bands(world, height, cost)
            assert len(bands) == world and bands[0][0] == 0 and sum(h for _, h in bands) == height
            for (y0, h), (y1, _) in zip(bands, bands[1:] + [(height, 0)]):
                assert h >= 8 and y0 + h == y1 and y0 % 8 == 0
            if world > 1 and height >= 1080:
                loads = [cost[y0:y0 + h].sum() for y0, h in bands]
                assert max(loads) < 1.35 * (sum(loads) / world)


def test_interleaved_rows_partition_the_frame():
    for world in (1, 2, 3, 8):
        for height in (8, 60, 1080, 2160, 4321):
            seen = np.zeros(height, int)
            for rank in range(world):
                y0, n, band, stride, rows = sharding.interleaved_rows(rank, world, height)
                assert n == len(rows) and band == 8 and stride == 8 * world and y0 == 8 * rank
                ly = np.arange(n)
                np.testing.assert_array_equal(rows, y0 + (ly // band) * stride + ly % band)  # the device's frame_y()
                seen[rows] += 1
            assert (seen == 1).all()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _pixel_value(H, W):
    y, x = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    return np.stack([x * 0.5 + y, y * 0.25 - x, x * y * 1e-3, np.ones_like(x)], -1)


def _worker(rank, world, port, H, W, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cost = np.where(np.arange(H) < H // 2, 1.0, 9.0) * W
    bands = sharding.balanced_bands(world, H, cost)
    y0, rows = bands[rank]
    band = torch.from_numpy(_pixel_value(H, W)[y0:y0 + rows].copy())     # "this rank's rendered tile"
    full = torch.zeros((H, W, 4)) if rank == 0 else None
    dist.barrier()
    out = sharding.gather_bands(dist, band, bands, rank, full)
    dist.barrier()
    # the interleaved form used by bench.py: 8-row bands dealt round-robin
    _, n, _, _, rows = sharding.interleaved_rows(rank, world, H)
    local = torch.from_numpy(_pixel_value(H, W)[rows].copy())
    full2 = torch.zeros((H, W, 4)) if rank == 0 else None
    out2 = sharding.gather_interleaved(dist, local, rank, world, H, full2)
    dist.barrier()
    if rank == 0:
        ok = bool((out.numpy() == _pixel_value(H, W)).all()) and bool((out2.numpy() == _pixel_value(H, W)).all())
        q.put((bands, ok))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gather_bands_over_gloo(world):
    H, W = 96, 40
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, H, W, q)) for r in range(world)]
    for p in procs:
        p.start()
    bands, ok = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok and len(bands) == world
    assert bands[0][1] > bands[-1][1]  # cheap rows get taller bands
