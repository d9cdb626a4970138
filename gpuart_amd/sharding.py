"""Screen-space sharding of one frame across ranks — host logic only.

Pixels are independent, so ranks share nothing per pass; the only exchange is the final gather of the radiance rows to
rank 0: gpuart_hip_gather (RCCL) in the library; here the numpy restatement of the share layout and the transport-agnostic
host form of the gather that the gloo tests and bench.py's announced fallback use."""
import numpy as np


def interleaved_rows(rank, world, height, band=8):
    """Frame rows owned by `rank` when row bands of `band` rows are dealt round-robin to the ranks
    (band k goes to rank k % world). Returns (y0, local_rows, band_rows, band_stride, rows) where `rows` is the
    array of frame rows in local order. The layout itself is defined in C (gpuart_hip_share_of_rank, include/gpuart_hip.h);
    this is its numpy restatement, kept for the tests that compare the two."""
    nbands = (height + band - 1) // band
    mine = np.arange(rank, nbands, world)
    rows = np.concatenate([np.arange(b * band, min(height, (b + 1) * band)) for b in mine]) if len(mine) else np.zeros(0, int)
    return rank * band, int(len(rows)), band, band * world, rows


def gather_shares_host(dist, tile, rank, world, W, H, full=None, root=0, band=8):
    """Transport-agnostic form of gpuart_hip_gather for tensors in HOST memory (gloo): every rank sends the rows of its
    share (tile: (th, W, 4) float32 tensor, share = gpuart_hip_share_of_rank), the root places them with the library's own
    host scatter (gpuart_hip_scatter_rows_host). Used by the CPU tests (world size 2, 3 over gloo) and by bench.py only as
    the announced fallback when the RCCL gather of the library cannot run. Returns `full` (H, W, 4) on root."""
    import torch
    from gpuart_amd import binding as B
    shares = [B.share_of_rank(W, H, r, world, band) for r in range(world)]
    assert tuple(tile.shape) == (shares[rank].th, W, 4), (tuple(tile.shape), shares[rank].th)
    if rank != root:
        if shares[rank].th:
            dist.send(tile.contiguous(), dst=root)
        return None
    out = full.numpy() if isinstance(full, torch.Tensor) else full
    for src in range(world):
        g = shares[src]
        if not g.th:
            continue
        if src == root:
            part = tile
        else:
            part = torch.empty((g.th, W, 4), dtype=torch.float32)
            dist.recv(part, src=src)
        B.scatter_rows_host(g, part.numpy(), out)
    return full
