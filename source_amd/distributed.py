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


def tile_rect(rank, world_size, nx, ny, bounds=None):
    """Column tile [x0, x1) x [0, ny) of rank (frame is x-major, so a tile is one contiguous block of the frame arrays).
    `bounds` = the world_size + 1 column boundaries of a cost-balanced split (balanced_bounds); default: equal widths."""
    if bounds is not None:
        if len(bounds) != world_size + 1 or bounds[0] != 0 or bounds[-1] != nx or any(b > a for a, b in zip(bounds[1:], bounds[:-1])):
            raise ValueError("tile bounds must be %d non-decreasing columns from 0 to %d" % (world_size + 1, nx))
        return (int(bounds[rank]), 0, int(bounds[rank + 1]), ny)
    x0 = (nx * rank) // world_size
    x1 = (nx * (rank + 1)) // world_size
    return (x0, 0, x1, ny)


def balanced_bounds(block_cost, block_width, nx, world_size):
    """Column boundaries of `world_size` contiguous tiles whose slowest tile is as fast as a split at block boundaries allows.
    block_cost[i] = measured cost of columns [i * block_width, (i + 1) * block_width). A scene's rays do not cost the same
    everywhere (configs[2]: the columns through the mesh instances take 1.4x the edge columns), and a step ends with its slowest
    rank; the reference balances with a task queue (workflow.py:201-251), a static frame split balances by cost instead.
    Exact minimum of the maximum tile cost (dynamic programme over the block prefix sums); every tile gets at least one block."""
    cost = [float(c) for c in block_cost]
    n = len(cost)
    if n < world_size:
        raise ValueError("need at least one cost block per rank")
    prefix = [0.0]
    for c in cost:
        prefix.append(prefix[-1] + c)
    inf = float("inf")
    # best[k][i] = the smallest possible maximum tile cost when the first i blocks are cut into k non-empty tiles
    best = [[inf] * (n + 1) for _ in range(world_size + 1)]
    cut = [[0] * (n + 1) for _ in range(world_size + 1)]
    best[0][0] = 0.0
    for k in range(1, world_size + 1):
        for i in range(k, n + 1):
            for s in range(k - 1, i):
                worst = max(best[k - 1][s], prefix[i] - prefix[s])
                if worst < best[k][i]:
                    best[k][i], cut[k][i] = worst, s
    bounds, i = [nx], n
    for k in range(world_size, 0, -1):
        i = cut[k][i]
        bounds.append(min(nx, i * block_width))
    return bounds[::-1]


def rebalance_bounds(bounds, times, nx, quantum=8):
    """One step of balancing column tiles by MEASURED tile times: every rank times its own tile of the current split, the times are
    shared, and the cuts move to the equal-cost quantiles of the piecewise-constant cost density times[r] / width[r]. Iterated a few
    times this converges on tiles of equal time — it sees what a sum of small-block costs does not: a tile's time is not additive in
    its columns (a bigger pass fills the chip better, its tail is amortised). Cuts stay multiples of `quantum` columns (the 8 x 8
    pixel tiles of the work units) and every tile keeps at least one quantum. Deterministic: every rank computes the same cuts."""
    n = len(times)
    if len(bounds) != n + 1:
        raise ValueError("one time per tile")
    widths = [bounds[r + 1] - bounds[r] for r in range(n)]
    density = [max(float(times[r]), 1e-12) / max(widths[r], 1) for r in range(n)]
    total = sum(density[r] * widths[r] for r in range(n))
    new, r, acc = [0], 0, 0.0
    for k in range(1, n):
        target = total * k / n
        while r < n - 1 and acc + density[r] * widths[r] < target:
            acc += density[r] * widths[r]
            r += 1
        x = bounds[r] + (target - acc) / density[r]
        x = int(round(x / quantum)) * quantum
        x = max(new[-1] + quantum, min(x, nx - (n - k) * quantum))
        new.append(x)
    new.append(nx)
    return new


def balance_tiles(cam, rank, world_size, allgather, synchronize, rounds=4, min_seconds=10e-3, quantum=8):
    """Column boundaries of a tile-sharded render balanced by MEASURED tile times (what the reference's task queue does dynamically,
    workflow.py:201-251, a static split has to do up front). Every rank times ordinary passes of its own tile of the current split —
    with Philox counters of their own (1 << 40 ...), into a frame that is dropped afterwards — the times are shared and the cuts
    move to the equal-cost quantiles (rebalance_bounds); `rounds` iterations bring mean / max tile time of configs[2] from 0.90 to
    0.97 at 8 ranks (tools/tile_balance.py). Every rank computes the same cuts from the same gathered times.

    cam: the observer to be sharded (HipEngine with rng='philox'; its frame sampler is left on this rank's balanced tile);
    allgather(x) -> [x of rank 0, ..., x of rank W-1] (FrameComm.allgather_scalar, a torch.distributed all_gather_object, ...);
    synchronize(): wait for the device (Context.synchronize). Returns the world_size + 1 boundaries."""
    import time
    from .optical.observer import RectFrameSampler2D
    nx, ny = cam.pixels
    bounds = [(nx * r) // world_size for r in range(world_size)] + [nx]
    if world_size == 1:
        return bounds
    engine = cam.render_engine
    saved = engine.sample_offset
    engine.sample_offset = 1 << 40
    for _ in range(rounds):
        cam.frame_sampler = RectFrameSampler2D(rect=tile_rect(rank, world_size, nx, ny, bounds))
        cam.observe()
        synchronize()
        reps, spent = 1, 0.0
        while True:
            t0 = time.perf_counter()
            for _ in range(reps):
                cam.observe()
            synchronize()
            spent = time.perf_counter() - t0
            if spent >= min_seconds or reps >= 64:
                break
            reps *= 4
        times = [float(t) for t in allgather(spent / reps)]
        bounds = [int(b) for b in rebalance_bounds(bounds, times, nx, quantum)]
    for pipe in cam.pipelines:                              # the timed passes are not part of the render: drop their frames
        frame = getattr(pipe, "frame", None)
        if frame is not None:
            frame.release()
            pipe.frame = None
    engine.sample_offset = saved
    cam.frame_sampler = RectFrameSampler2D(rect=tile_rect(rank, world_size, nx, ny, bounds))
    return bounds


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


def gather_tile_sharded(mean, variance, samples, rank, dist, bounds=None):
    """Tile sharding on torch tensors (bench.py --collective torch, and the gloo tests): every rank holds full-size x-major frame
    tensors of which it rendered the column tile tile_rect(rank); returns the assembled (mean, variance, samples). Tiles of unequal
    width (nx not divisible by the world size) are padded to the widest one for the all_gather and cut back afterwards."""
    import torch
    world = dist.get_world_size()
    nx, ny = mean.shape[0], mean.shape[1]
    rects = [tile_rect(r, world, nx, ny, bounds) for r in range(world)]
    widest = max(r[2] - r[0] for r in rects)
    x0, _, x1, _ = rects[rank]
    out = []
    for t in (mean, variance, samples):
        tile = torch.zeros((widest,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        tile[:x1 - x0] = t[x0:x1]
        parts = [torch.empty_like(tile) for _ in range(world)]
        dist.all_gather(parts, tile)
        out.append(torch.cat([p[:r[2] - r[0]] for p, r in zip(parts, rects)], dim=0))
    return tuple(out)


def gather_tiles(tile, dist):
    """All-gather equally sized column tiles [x0:x1, :, :] of an x-major frame tensor into the full frame."""
    import torch
    world = dist.get_world_size()
    parts = [torch.empty_like(tile) for _ in range(world)]
    dist.all_gather(parts, tile.contiguous())
    return torch.cat(parts, dim=0)


def frame_segment(n, world_size, rank):
    """(offset, length) of the frame segment rank owns in rsx_allreduce_frame's reduce-scatter (librsx's own arithmetic)."""
    import ctypes as C
    from . import _lib
    off, length = C.c_int64(0), C.c_int64(0)
    _lib.check(_lib.lib().rsx_frame_segment(int(n), int(world_size), int(rank), C.byref(off), C.byref(length)))
    return int(off.value), int(length.value)


def slice_bounds(n_slices, world_size):
    """Slice sharding: rank r renders the spectral slices [bounds[r], bounds[r + 1]) (512 slices -> 64 per GPU)."""
    return [(n_slices * r) // world_size for r in range(world_size)] + [n_slices]


def gather_slice_sharded(mean, variance, samples, rank, dist, bin_bounds):
    """torch / gloo form of rsx_allgather_bins (test aid and fallback): every rank contributes frame[..., b0:b1] of its bin range."""
    import torch
    world = dist.get_world_size()
    out = []
    for a in (mean, variance, samples):
        full = a.clone()
        for r in range(world):
            b0, b1 = int(bin_bounds[r]), int(bin_bounds[r + 1])
            if b1 <= b0:
                continue
            part = a[..., b0:b1].contiguous() if r == rank else torch.empty(a.shape[:-1] + (b1 - b0,), dtype=a.dtype, device=a.device)
            dist.broadcast(part, src=r)
            full[..., b0:b1] = part
        out.append(full)
    return out


class FrameComm:
    """rsx_comm wrapper: the framebuffer exchange of a multi-GPU render over RCCL, called straight from librsx (no PyTorch in the
    data path). ``exchange(payload)`` is any callable that returns rank 0's payload on every rank (a torch.distributed
    broadcast_object_list over gloo, an MPI bcast, a shared file ...): it carries the 128-byte RCCL unique id."""

    def __init__(self, context, rank, world_size, exchange):
        import ctypes as C
        from . import _lib
        self.context, self.rank, self.world_size = context, int(rank), int(world_size)
        L = _lib.lib()
        ident = (C.c_char * 128)()
        if self.rank == 0:
            _lib.check(L.rsx_comm_unique_id(ident))
        payload = exchange(bytes(ident.raw) if self.rank == 0 else None)
        ident = (C.c_char * 128).from_buffer_copy(payload)
        self._h = C.c_void_p()
        _lib.check(L.rsx_comm_create(context.handle, self.world_size, self.rank, ident, C.byref(self._h)))

    def barrier(self):
        from . import _lib
        _lib.check(_lib.lib().rsx_comm_barrier(self._h))

    def max(self, value):
        import ctypes as C
        from . import _lib
        v = C.c_double(float(value))
        _lib.check(_lib.lib().rsx_comm_max_f64(self._h, C.byref(v)))
        return float(v.value)

    def allgather_scalar(self, value):
        """[value of rank 0, ..., value of rank W-1] on every rank (W max-reductions of one double: control-plane sized)."""
        return [self.max(value if r == self.rank else float("-inf")) for r in range(self.world_size)]

    def allgather_tiles(self, frame, nx, ny, bounds=None):
        """frame: StatsArray3D whose column tile tile_rect(rank) this rank rendered; afterwards the whole frame on every rank."""
        from . import _lib
        begin = np.array([tile_rect(r, self.world_size, nx, ny, bounds)[0] * ny * frame.nz for r in range(self.world_size)] + [nx * ny * frame.nz], dtype=np.int64)
        fm, fv, fn = frame._device(self.context)
        _lib.check(_lib.lib().rsx_allgather_frame(self._h, fm, fv, fn, _lib.ptr(begin)))
        frame._mark_device_written()

    def allreduce_samples(self, frame):
        """frame: StatsArray3D holding this rank's samples of every pixel; afterwards the combine_samples fold over the ranks."""
        from . import _lib
        fm, fv, fn = frame._device(self.context)
        _lib.check(_lib.lib().rsx_allreduce_frame(self._h, fm, fv, fn, frame.length))
        frame._mark_device_written()

    def allgather_slices(self, frame, nx, ny, bin_bounds):
        """frame: StatsArray3D of which this rank rendered the spectral slices filling bins [bin_bounds[rank], bin_bounds[rank + 1]);
        afterwards every bin on every rank (slice sharding)."""
        from . import _lib
        begin = np.asarray(bin_bounds, dtype=np.int32)
        assert len(begin) == self.world_size + 1
        fm, fv, fn = frame._device(self.context)
        _lib.check(_lib.lib().rsx_allgather_bins(self._h, fm, fv, fn, nx * ny, frame.nz, _lib.ptr(begin)))
        frame._mark_device_written()

    def size(self):
        """Ranks RCCL says the communicator spans."""
        import ctypes as C
        from . import _lib
        n = C.c_int32(0)
        _lib.check(_lib.lib().rsx_comm_size(self._h, C.byref(n)))
        return int(n.value)

    def close(self):
        from . import _lib
        if self._h:
            _lib.lib().rsx_comm_free(self._h)
            self._h = None
