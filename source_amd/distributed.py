"""
Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU
tests). Rays shard with no data-path collective; the only exchange is the final spectral framebuffer:

* sample sharding — every rank renders the full frame with its own Philox sample counters; ``merge_sample_sharded`` exchanges
  the (mean, variance, samples) frames and folds them in rank order with the combine_samples law (statsarray.pyx:780-859), so the
  result is deterministic and identical on every rank;
* tile sharding — ``gather_tiles`` all-gathers column tiles of the x-major frame (contiguous, no arithmetic).
"""
import numpy as np


def rank_sample_offset(step, rank, world_size, spp):
    """First Philox sample counter of `rank` in pass `step`: ranks and passes never reuse a (pixel, sample) counter."""
    return (step * world_size + rank) * spp


def tile_rect(rank, world_size, nx, ny):
    """Column tile [x0, x1) x [0, ny) of rank (frame is x-major, so a tile is one contiguous block of the frame arrays)."""
    x0 = (nx * rank) // world_size
    x1 = (nx * (rank + 1)) // world_size
    return (x0, 0, x1, ny)


def combine_arrays(ma, va, na, mb, vb, nb):
    """Vectorised numpy restatement of _combine_samples (core/math/statsarray.pyx:780-859): returns combine(a, b)."""
    ma, va, mb, vb = (np.asarray(x, dtype=np.float64) for x in (ma, va, mb, vb))
    na, nb = np.asarray(na, dtype=np.int64), np.asarray(nb, dtype=np.int64)
    swap = na < nb
    mx, my = np.where(swap, mb, ma), np.where(swap, ma, mb)
    vx, vy = np.where(swap, vb, va), np.where(swap, va, vb)
    nx, ny = np.where(swap, nb, na), np.where(swap, na, nb)
    with np.errstate(divide="ignore", invalid="ignore"):
        nt = nx + ny
        mt = (nx * mx + ny * my) / nt
        bx = (nx - 1) * vx / nx
        by = (ny - 1) * vy / ny
        vt = (nx * (mx * mx + bx) + ny * (my * my + by)) / nt - mt * mt
        vt = nt * vt / (nt - 1)
        # special cases
        two = (nx == 1) & (ny == 1)
        m2 = 0.5 * (mx + my)
        v2 = 2 * (mx - m2) * (mx - m2)
        add1 = (nx > 1) & (ny == 1)                       # _add_sample(my) onto set x
        n1 = nx + 1
        m1 = mx + (my - mx) / n1
        v1 = (vx * (nx - 1) + (my - mx) * (my - m1)) / (n1 - 1)
    general = (nx > 1) & (ny > 1)
    keep = (ny == 0)                                       # nothing to add: (nx==0,ny==0) -> zeros, (1,0) -> (mx,0,1), (>1,0) -> x
    out_m = np.where(general, mt, np.where(two, m2, np.where(add1, m1, np.where(keep & (nx > 0), mx, 0.0))))
    out_v = np.where(general, vt, np.where(two, v2, np.where(add1, v1, np.where(keep & (nx > 1), vx, 0.0))))
    out_n = np.where(general, nt, np.where(two, 2, np.where(add1, n1, nx)))
    return out_m, out_v, out_n.astype(np.int32)


def merge_sample_sharded(mean, variance, samples, dist, combine=None, mode="scatter"):
    """
    mean/variance/samples: this rank's frame as torch tensors (CUDA for nccl, CPU for gloo). Returns the merged frame
    (same on every rank). `combine(m, v, n, mb, vb, nb)` folds b into a in place; default = numpy restatement (CPU tensors).

    Every element of the result is the fold combine(...combine(combine(rank 0, rank 1), rank 2)..., rank W-1): rank order, so the
    merge is deterministic and independent of how the exchange is routed. Two routings:

    * ``"scatter"`` (default) — the reduce-scatter / all-gather shape of a ring all-reduce with combine_samples as the operator:
      one all_to_all hands rank j the j-th 1/W segment of every rank's frame, rank j folds its W pieces, one all_gather
      redistributes the merged segments. Each rank receives 2 (W-1)/W frames instead of W-1 (xGMI links are point to point, so
      bytes per link is what bounds the exchange);
    * ``"gather"`` — all_gather of whole frames, every rank folds everything (simple; fine for W = 2).
    """
    import torch
    world = dist.get_world_size()

    def fold(parts_m, parts_v, parts_n):
        m, v, n = parts_m[0].clone(), parts_v[0].clone(), parts_n[0].clone()
        for r in range(1, world):
            mb, vb, nb = parts_m[r], parts_v[r], parts_n[r]
            if combine is not None:
                combine(m, v, n, mb, vb, nb)
            else:
                om, ov, on = combine_arrays(m.numpy(), v.numpy(), n.numpy(), mb.numpy(), np.maximum(vb.numpy(), 0.0), nb.numpy())
                m, v, n = torch.from_numpy(om), torch.from_numpy(ov), torch.from_numpy(on)
        return m, v, n

    if mode == "gather" or world == 1:
        gathered = []
        for t in (mean, variance, samples):
            parts = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(parts, t.contiguous())
            gathered.append(parts)
        return fold(*gathered)

    shape, length = mean.shape, mean.numel()
    seg = -(-length // world)                                # segment length, frame padded to world * seg elements
    pieces = []
    for t in (mean, variance, samples):
        flat = t.contiguous().reshape(-1)
        if seg * world != length:
            flat = torch.cat([flat, torch.zeros(seg * world - length, dtype=t.dtype, device=t.device)])   # n = 0: neutral element
        recv = torch.empty_like(flat)
        dist.all_to_all_single(recv, flat)                   # recv[r*seg:(r+1)*seg] = rank r's copy of my segment
        pieces.append([recv[r * seg:(r + 1) * seg] for r in range(world)])
    m, v, n = fold(*pieces)
    out = []
    for t, like in ((m, mean), (v, variance), (n, samples)):
        full = torch.empty(seg * world, dtype=like.dtype, device=like.device)
        dist.all_gather_into_tensor(full, t.contiguous())
        out.append(full[:length].reshape(shape))
    return tuple(out)


def gather_tiles(tile, dist):
    """All-gather equally sized column tiles [x0:x1, :, :] of an x-major frame tensor into the full frame."""
    import torch
    world = dist.get_world_size()
    parts = [torch.empty_like(tile) for _ in range(world)]
    dist.all_gather(parts, tile.contiguous())
    return torch.cat(parts, dim=0)
