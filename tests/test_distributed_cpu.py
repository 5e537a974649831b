"""world_size-2 gloo tests of the multi-GPU merge path (CPU): sample-sharded frames merged with the combine_samples law in rank
order, and tile-sharded frames assembled by all_gather. The per-rank frames come from the oracle (no GPU here)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world_size, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world_size))
    import torch
    import torch.distributed as dist
    from oracle import oracle as orc
    from source_amd import api as ns, scenes
    from source_amd import distributed as D
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    world, mesh, box = scenes.build_c2(ns, n=24)
    nx, ny, spp, bins = 24, 16, 3, 5
    cam, pipe = scenes.c2_camera(ns, world, (nx, ny), spp=spp, bins=bins)
    flat = world.flatten()
    sl = cam._slice_spectrum()[0]
    # --- sample sharding ---------------------------------------------------------------------------------
    eng = ns.HipEngine(rng="philox", seed=5, sample_offset=D.rank_sample_offset(0, rank, world_size, spp))
    keep = []
    desc = cam.render_desc(world, None, sl, eng, keep, rect=(0, 0, nx, ny))
    m, v, rays = orc.render_pinhole(flat, desc)
    to_frame = lambda a: np.ascontiguousarray(a.reshape(ny, nx, bins).transpose(1, 0, 2))   # rect tasks are iy-outer
    fm, fv = to_frame(m), to_frame(v)
    fn = np.full((nx, ny, bins), spp, dtype=np.int32)
    M, V, N = D.merge_sample_sharded(torch.from_numpy(fm), torch.from_numpy(fv), torch.from_numpy(fn), dist)
    G = D.merge_sample_sharded(torch.from_numpy(fm), torch.from_numpy(fv), torch.from_numpy(fn), dist, mode="gather")
    assert all(torch.equal(a, b) for a, b in zip((M, V, N), G))        # both routings give the same fold, bit for bit
    # odd length (padding path): a 7-element frame
    odd = [torch.from_numpy(np.ascontiguousarray(a.reshape(-1)[:7])) for a in (fm, fv, fn)]
    assert all(torch.equal(a, b) for a, b in zip(D.merge_sample_sharded(*odd, dist), D.merge_sample_sharded(*odd, dist, mode="gather")))
    # --- tile sharding -------------------------------------------------------------------------------------
    rect = D.tile_rect(rank, world_size, nx, ny)
    eng2 = ns.HipEngine(rng="philox", seed=5)
    desc2 = cam.render_desc(world, None, sl, eng2, keep, rect=rect)
    tm, tv, _ = orc.render_pinhole(flat, desc2)
    w = rect[2] - rect[0]
    tile = np.ascontiguousarray(tm.reshape(ny, w, bins).transpose(1, 0, 2))
    full = D.gather_tiles(torch.from_numpy(tile), dist)
    # bench.py's tile path (--collective torch): full-size frames of which each rank filled its own tile; 25 columns over 2 ranks
    # would be uneven, so also exercise the padded gather on a 5-column crop
    fr = [np.zeros((nx, ny, bins)), np.zeros((nx, ny, bins)), np.zeros((nx, ny, bins), dtype=np.int32)]
    fr[0][rect[0]:rect[2]] = tile
    fr[1][rect[0]:rect[2]] = np.ascontiguousarray(tv.reshape(ny, w, bins).transpose(1, 0, 2))
    fr[2][rect[0]:rect[2]] = spp
    tm_, tv_, tn_ = D.gather_tile_sharded(*(torch.from_numpy(a) for a in fr), rank, dist)
    assert torch.equal(tm_, full) and int(tn_.min()) == int(tn_.max()) == spp
    r5 = D.tile_rect(rank, world_size, 5, ny)
    odd = np.zeros((5, ny, bins))
    odd[r5[0]:r5[2]] = rank + 1.0
    o5 = D.gather_tile_sharded(torch.from_numpy(odd), torch.from_numpy(odd.copy()), torch.from_numpy(odd.astype(np.int32)), rank, dist)[0].numpy()
    assert (o5[:2] == 1.0).all() and (o5[2:] == 2.0).all()
    # cost-balanced tiles (bench.py --tiles balanced): unequal widths from measured block costs, same assembly
    bounds = D.balanced_bounds([5.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0], 1, 7, world_size)
    assert bounds == [0, 2, 7] or bounds == [0, 1, 7]
    rb = D.tile_rect(rank, world_size, 7, ny, bounds)
    odd = np.zeros((7, ny, bins))
    odd[rb[0]:rb[2]] = rank + 1.0
    ob = D.gather_tile_sharded(torch.from_numpy(odd), torch.from_numpy(odd.copy()), torch.from_numpy(odd.astype(np.int32)), rank, dist, bounds)[0].numpy()
    assert (ob[:bounds[1]] == 1.0).all() and (ob[bounds[1]:] == 2.0).all()
    # --- slice sharding (bench.py --sharding slice, host route): each rank owns a range of bins of every pixel -------------------------
    bb = [0, 2, bins]
    own = np.zeros((nx, ny, bins))
    own[..., bb[rank]:bb[rank + 1]] = rank + 1.0
    sm_, sv_, sn_ = D.gather_slice_sharded(torch.from_numpy(own), torch.from_numpy(own.copy()), torch.from_numpy(own.astype(np.int32)), rank, dist, bb)
    assert (sm_.numpy()[..., :2] == 1.0).all() and (sm_.numpy()[..., 2:] == 2.0).all() and torch.equal(sm_, sv_) and (sn_.numpy()[..., 2:] == 2).all()
    assert D.slice_bounds(512, 8) == [64 * r for r in range(9)] and D.slice_bounds(5, 2) == [0, 2, 5]
    np.savez(out % rank, M=M.numpy(), V=V.numpy(), N=N.numpy(), fm=fm, fv=fv, full=full.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_merge(tmp_path, orc, ns):
    import torch.multiprocessing as mp
    from source_amd import scenes
    from source_amd import distributed as D
    out = str(tmp_path / "rank%d.npz")
    mp.spawn(_worker, args=(2, 29517 + os.getpid() % 500, out), nprocs=2, join=True)
    r0, r1 = np.load(out % 0), np.load(out % 1)
    for k in ("M", "V", "N", "full"):
        assert np.array_equal(r0[k], r1[k]), k                      # every rank ends with the same frame
    # merged frame == reference law applied to (rank0, rank1) in rank order (oracle's combine)
    spp = 3
    n = np.full(r0["fm"].shape, spp, dtype=np.int32)
    m, v, nn = orc.frame_combine(r0["fm"], r0["fv"], n, r1["fm"], np.maximum(r1["fv"], 0), n)
    assert np.array_equal(r0["M"], m) and np.array_equal(r0["V"], v) and np.array_equal(r0["N"], nn)
    assert (r0["N"] == 2 * spp).all()
    # ... and agrees with a single-process render of all 2*spp samples to rounding (different summation order)
    world, mesh, box = scenes.build_c2(ns, n=24)
    cam, pipe = scenes.c2_camera(ns, world, (24, 16), spp=2 * spp, bins=5)
    keep = []
    desc = cam.render_desc(world, None, cam._slice_spectrum()[0], ns.HipEngine(rng="philox", seed=5), keep, rect=(0, 0, 24, 16))
    sm, sv, _ = orc.render_pinhole(world.flatten(), desc)
    sm = sm.reshape(16, 24, 5).transpose(1, 0, 2)
    sv = sv.reshape(16, 24, 5).transpose(1, 0, 2)
    assert np.allclose(r0["M"], sm, rtol=1e-12, atol=0) and np.allclose(r0["V"], sv, rtol=1e-9, atol=1e-300)
    # tile sharding: assembling rank tiles reproduces the single-process frame bit for bit (Philox counters are per pixel)
    desc1 = cam.render_desc(world, None, cam._slice_spectrum()[0], ns.HipEngine(rng="philox", seed=5), keep, rect=(0, 0, 24, 16))
    cam.pixel_samples = spp
    desc1 = cam.render_desc(world, None, cam._slice_spectrum()[0], ns.HipEngine(rng="philox", seed=5), keep, rect=(0, 0, 24, 16))
    fm1, _, _ = orc.render_pinhole(world.flatten(), desc1)
    assert np.array_equal(r0["full"], fm1.reshape(16, 24, 5).transpose(1, 0, 2))


def test_balanced_tile_bounds():
    """Contiguous tiles of (near) equal measured cost: every tile non-empty, cuts at block boundaries, never worse than equal widths."""
    from source_amd import distributed as D
    rng = np.random.default_rng(3)
    for n_blocks, ranks, bw, nx in ((32, 8, 64, 2048), (32, 4, 64, 2048), (32, 2, 32, 1024), (8, 8, 3, 24), (33, 5, 3, 97)):
        cost = 1.0 + rng.random(n_blocks) * (np.arange(n_blocks) % 7 == 0) * 4
        b = D.balanced_bounds(cost, bw, nx, ranks)
        assert len(b) == ranks + 1 and b[0] == 0 and b[-1] == nx and all(x1 > x0 for x0, x1 in zip(b[:-1], b[1:]))
        assert all(x % bw == 0 for x in b[:-1])
        tile_cost = lambda bounds: max(sum(cost[x0 // bw:-(-x1 // bw)]) for x0, x1 in zip(bounds[:-1], bounds[1:]))
        equal = [(n_blocks * r // ranks) * bw for r in range(ranks)] + [nx]
        assert tile_cost(b) <= tile_cost(equal) + 1e-12
        for r in range(ranks):
            assert D.tile_rect(r, ranks, nx, 5, b) == (b[r], 0, b[r + 1], 5)
    with pytest.raises(ValueError):
        D.balanced_bounds([1.0, 1.0], 4, 8, 3)
    with pytest.raises(ValueError):
        D.tile_rect(0, 2, 8, 5, [0, 9, 8])


def test_rebalance_bounds_converges_on_equal_tile_times():
    """Cuts moved to the equal-cost quantiles of measured tile times: with a fixed (unknown to the function) per-column cost the
    slowest tile approaches the mean within a few rounds; cuts stay multiples of 8 columns, tiles stay non-empty."""
    from source_amd import distributed as D
    nx = 2048
    col = 1.0 + 0.6 * np.exp(-((np.arange(nx) - 900.0) / 350.0) ** 2) + 0.3 * (np.arange(nx) > 1500)
    for n in (2, 4, 8):
        b = [(nx * r) // n for r in range(n)] + [nx]
        first = None
        for it in range(5):
            times = [col[b[r]:b[r + 1]].sum() for r in range(n)]
            balance = np.mean(times) / np.max(times)
            first = balance if first is None else first
            b = D.rebalance_bounds(b, times, nx)
            assert b[0] == 0 and b[-1] == nx and all(x1 > x0 for x0, x1 in zip(b[:-1], b[1:])) and all(x % 8 == 0 for x in b)
        assert balance > 0.97 and balance >= first
    assert D.rebalance_bounds([0, 8, 16], [1.0, 100.0], 16) == [0, 8, 16]        # one quantum per tile is the floor
    with pytest.raises(ValueError):
        D.rebalance_bounds([0, 8, 16], [1.0], 16)


def test_combine_arrays_matches_reference(golden):
    from source_amd import distributed as D
    g = golden("f09_stats")
    m, v, n = D.combine_arrays(g["ma"], g["va"], g["na"], g["mb"], np.maximum(g["vb"], 0), g["nb"])
    assert np.array_equal(np.stack([m, v, n.astype(float)], axis=1), g["comb"])


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT torch.distributed.run around it (what a driver that launches N = 1 as `python bench.py --gpus 1`
    may well do for N = 2): bench.py re-executes itself under torch.distributed.run on 127.0.0.1 at a free port, the two ranks meet
    (gloo control plane) and rank 0's line comes back through the parent. No GPU here: RSX_BENCH_RENDEZVOUS_ONLY stops every rank after
    the first barrier (the GPU suite runs the same self-launch through a whole two-rank bench, test_bench_self_launch_two_ranks_one_gpu)."""
    import json
    import subprocess
    env = dict(os.environ, RSX_BENCH_RENDEZVOUS_ONLY="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--collective", "host"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["rendezvous_only"] is True
