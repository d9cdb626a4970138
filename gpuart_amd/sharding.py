"""Screen-space sharding of one frame across ranks — host logic only.

Pixels are independent, so ranks share nothing per pass; the only exchange is the final gather of the radiance
bands to rank 0 (RCCL on GPUs; the same code runs over gloo on CPU tensors in the tests)."""
import numpy as np


def balanced_bands(world, height, cost_rows, align=8):
    """Splits rows [0, height) into `world` contiguous bands [(y0, rows), ...] of roughly equal cost.

    cost_rows: per-row cost estimate (e.g. 1 per sky pixel, 10 per pixel that hits geometry, from a cheap
    direct-lighting probe that every rank renders identically). Cuts are multiples of `align` rows (8x8 pixel tiles)
    and every band is at least `align` rows high."""
    assert world >= 1 and height >= world * align, "frame too small for that many bands"
    c = np.cumsum(np.asarray(cost_rows, np.float64))
    total = c[-1] if c[-1] > 0 else 1.0
    cuts = [0]
    for r in range(1, world):
        y = int(np.searchsorted(c, total * r / world))
        y = (y + align // 2) // align * align
        y = max(cuts[-1] + align, min(height - align * (world - r), y))
        cuts.append(y)
    cuts.append(height)
    return [(cuts[i], cuts[i + 1] - cuts[i]) for i in range(world)]


def cost_rows_from_probe(probe_rgb):
    """Row costs from a direct-lighting probe frame (H, W, 3): sky pixels (equal to the top row's sky gradient
    within 1e-3) cost 1, anything else 10."""
    sky = np.abs(probe_rgb - probe_rgb[-1:, :, :]).sum(-1) < 1e-3
    return np.where(sky, 1.0, 10.0).sum(1)


def gather_bands(dist, band, bands, rank, full=None, root=0):
    """Gathers row bands (tensors of shape (rows_r, W, C), possibly different heights) into `full` on `root`.

    Point-to-point sends (every peer owns a direct xGMI link to the root on an 8-GPU MI355X node), no collective that
    would need equal sizes. Returns `full` on root, None elsewhere."""
    if rank == root:
        y0, rows = bands[root]
        full[y0:y0 + rows].copy_(band)
        reqs = [dist.irecv(full[b0:b0 + bh], src=src) for src, (b0, bh) in enumerate(bands) if src != root]
        for q in reqs:
            q.wait()
        return full
    dist.isend(band.contiguous(), dst=root).wait()
    return None


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
