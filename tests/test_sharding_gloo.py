"""CPU tests of the multi-rank host logic: the share layout + the host form of the gather over torch.distributed/gloo, world size 2 and 3."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gpuart_amd import sharding


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


def test_c_share_layout_matches_the_numpy_restatement():
    """gpuart_hip_share_of_rank / gpuart_hip_frame_row (pure host functions of libgpuart_hip.so) against interleaved_rows."""
    from gpuart_amd import binding as B
    for world in (1, 2, 3, 8):
        for W, H in ((40, 8), (37, 61), (1920, 1080), (3840, 2160), (7680, 4320)):
            seen = np.zeros(H, int)
            for rank in range(world):
                g = B.share_of_rank(W, H, rank, world)
                y0, n, band, stride, rows = sharding.interleaved_rows(rank, world, H)
                assert (g.W, g.H, g.x0, g.tw, g.y0, g.th, g.band_rows, g.band_stride) == (W, H, 0, W, y0, n, band, stride)
                np.testing.assert_array_equal(g.rows(), rows)
                seen[rows] += 1
            assert (seen == 1).all()
    with pytest.raises(ValueError):
        B.share_of_rank(64, 64, 2, 2)


def test_share_table_validation_of_the_gather():
    """What every rank of gpuart_hip_gather checks before any transfer is posted (csrc/hip/gpuart_hip.hip check_shares, through
    gpuart_hip_test_share_table): the library's own round-robin shares pass for every rank count — also with more ranks than bands —,
    and a table that leaves rows to nobody, deals a row twice, mixes frame sizes, holds a partial-width share or a rank that
    reported a failure is refused with the same verdict on every rank."""
    import ctypes as C
    from gpuart_amd import binding as B
    L = B.hip_lib()

    def verdict(shares, status=None):
        arr = (B.TileGeom * len(shares))(*shares)
        st = None if status is None else (C.c_uint32 * len(shares))(*status)
        rc = L.gpuart_hip_test_share_table(arr, st, len(shares), 1, 0)
        return rc, L.gpuart_hip_last_error().decode()

    for W, H in ((1920, 1080), (37, 61), (64, 8), (5, 3)):
        for n in (1, 2, 3, 8, 11):
            for band in (1, 3, 8, 16):
                shares = [B.share_of_rank(W, H, r, n, band) for r in range(n)]
                assert verdict(shares)[0] == 0, (W, H, n, band, verdict(shares)[1])
    W, H = 64, 48
    good = [B.share_of_rank(W, H, r, 3) for r in range(3)]
    assert any(g.th == 0 for g in [B.share_of_rank(64, 8, r, 4) for r in range(4)])  # (empty shares were part of the loop above)
    rc, why = verdict(good[:2] + [good[1]]); assert rc != 0 and "overlap" in why
    rc, why = verdict(good[:2]); assert rc != 0 and "belongs to no rank" in why
    rc, why = verdict([good[0], good[1], B.share_of_rank(W, H + 8, 2, 3)]); assert rc != 0 and "inconsistent" in why
    narrow = B.share_of_rank(W, H, 2, 3); narrow.tw = W - 8
    rc, why = verdict([good[0], good[1], narrow]); assert rc != 0 and "full-width" in why
    rc, why = verdict(good, status=[0, 1, 0]); assert rc != 0 and "rank 1 could not prepare" in why
    assert verdict(good, status=[0, 0, 0])[0] == 0
    # a bogus frame in rank 0's entry (caller / peer data sizes the coverage map): an argument error, not a 4 GB allocation or an
    # exception across the C ABI
    for bad_h in (0xffffffff, 70000, 0):
        bogus = B.share_of_rank(W, H, 0, 1); bogus.H = bad_h
        rc, why = verdict([bogus]); assert rc == -1 and "inconsistent shares" in why, (bad_h, rc, why)


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
    # the interleaved shares bench.py and gpuart_cli --gpus use: the library's own layout + host scatter (the host half of
    # gpuart_hip_gather), incl. a ragged last band and a frame with fewer bands than ranks (an empty share)
    ok = True
    for (h3, w3) in ((H, W), (61, 37), (8, 5)):
        from gpuart_amd import binding as B
        g = B.share_of_rank(w3, h3, rank, world)
        mine = torch.from_numpy(_pixel_value(h3, w3)[g.rows()].copy()) if g.th else torch.zeros((0, w3, 4))
        full3 = torch.full((h3, w3, 4), -1.0) if rank == 0 else None
        out3 = sharding.gather_shares_host(dist, mine, rank, world, w3, h3, full3)
        dist.barrier()
        if rank == 0:
            ok = ok and bool((out3.numpy() == _pixel_value(h3, w3)).all())
    if rank == 0:
        q.put(ok)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gather_shares_over_gloo(world):
    H, W = 96, 40
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, H, W, q)) for r in range(world)]
    for p in procs:
        p.start()
    ok = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok
