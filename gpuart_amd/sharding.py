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
    array of frame rows in local order — the arguments of gpuart_hip_set_tile_interleaved plus the scatter map."""
    nbands = (height + band - 1) // band
    mine = np.arange(rank, nbands, world)
    rows = np.concatenate([np.arange(b * band, min(height, (b + 1) * band)) for b in mine]) if len(mine) else np.zeros(0, int)
    return rank * band, int(len(rows)), band, band * world, rows


def gather_interleaved(dist, local, rank, world, height, full=None, root=0, band=8):
    """Gathers interleaved row bands (local: (local_rows, W, C) tensor) into `full` (height, W, C) on root.
    Point-to-point: the root posts all receives at once (every peer has its own xGMI link to the root), peers send."""
    import torch
    if rank == root:
        bufs, ops = {}, []
        for src in range(world):
            _, n, _, _, rows = interleaved_rows(src, world, height, band)
            if n == 0:
                continue
            if src == root:
                bufs[src] = (local, rows)
            else:
                buf = torch.empty((n,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
                bufs[src] = (buf, rows)
                ops.append(dist.P2POp(dist.irecv, buf, src))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        for src, (buf, rows) in bufs.items():
            full[torch.as_tensor(rows, device=full.device)] = buf
        return full
    if local.shape[0]:
        for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, local.contiguous(), root)]):
            req.wait()
    return None
