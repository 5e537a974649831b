"""One rank of the stub-transport exchange test (tests/test_gpu_parity.py::test_multi_rank_exchange_through_transport_stub):
renders its shard of a small frame through HipEngine on the one GPU of the box and runs librsx's own exchange — FrameComm over
RSX_RCCL_LIB = tests/stub_rccl's librccl_stub.so — with the other ranks. Writes the exchanged frame to <dir>/<mode>_rank<r>.npz.

usage: worker.py <rank> <world_size> <dir> <mode>[,<mode>...]      modes: tile, tile_balanced, sample, slice, slice_chunked
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
rank, world_size, directory, modes = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4].split(",")

from source_amd import api as ns, scenes                    # noqa: E402
from source_amd import distributed as D                     # noqa: E402
from source_amd.device import get_context                   # noqa: E402

NX, NY, SPP, BINS, PASSES = 72, 40, 3, 6, 2


def exchange_id(payload):
    """rank 0's 128-byte id through a file (any channel will do: FrameComm only asks for a callable)"""
    path = os.path.join(directory, "unique_id_%s" % modes[0])
    if rank == 0:
        with open(path + ".tmp", "wb") as f:
            f.write(payload)
        os.rename(path + ".tmp", path)
        return payload
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > 120:
            raise RuntimeError("no unique id from rank 0")
        time.sleep(0.01)
    return open(path, "rb").read()


ctx = get_context()
comm = D.FrameComm(ctx, rank, world_size, exchange_id)
assert comm.size() == world_size
assert comm.allgather_scalar(10.0 + rank) == [10.0 + r for r in range(world_size)]
assert comm.max(float(rank)) == float(world_size - 1)
comm.barrier()


def build(slices=1):
    world, mesh, box = scenes.build_c2(ns, n=24)
    cam, pipe = scenes.c2_camera(ns, world, (NX, NY), spp=SPP, bins=BINS)
    cam.spectral_rays = slices
    cam.render_engine = ns.HipEngine(rng="philox", seed=77)
    cam.frame_sampler = ns.RectFrameSampler2D()
    return world, cam, pipe


for mode in modes:
    if mode in ("tile", "tile_balanced"):
        world, cam, pipe = build()
        bounds = None
        if mode == "tile_balanced":
            bounds = D.balance_tiles(cam, rank, world_size, comm.allgather_scalar, ctx.synchronize, rounds=2, min_seconds=1e-3)
            assert bounds[0] == 0 and bounds[-1] == NX and all(b > a for a, b in zip(bounds[:-1], bounds[1:]))
        cam.frame_sampler = ns.RectFrameSampler2D(rect=D.tile_rect(rank, world_size, NX, NY, bounds))
        for p in range(PASSES):
            cam.render_engine.sample_offset = p * SPP        # the counters of a one-GPU render
            cam.observe()
        comm.allgather_tiles(pipe.frame, NX, NY, bounds)
    elif mode == "sample":
        world, cam, pipe = build()
        for p in range(PASSES):
            cam.render_engine.sample_offset = D.rank_sample_offset(p, rank, world_size, SPP)
            cam.observe()
        comm.allreduce_samples(pipe.frame)
    elif mode in ("slice", "slice_chunked"):
        # (slice_chunked: a workspace of 50 000 bytes pushes rsx_allgather_bins through seven pixel chunks — the frame is 72 x 40 x 6 x 20 B)
        if mode == "slice_chunked":
            os.environ["RSX_BINS_SCRATCH_BYTES"] = "50000"
        else:
            os.environ.pop("RSX_BINS_SCRATCH_BYTES", None)
        world, cam, pipe = build(slices=BINS)
        sb = D.slice_bounds(BINS, world_size)
        sl = cam._slice_spectrum()
        bin_bounds = [sl[k].offset if k < BINS else BINS for k in sb]
        cam.render_engine.slice_range = (sb[rank], sb[rank + 1])
        for p in range(PASSES):
            cam.render_engine.sample_offset = p * SPP
            cam.observe()
        comm.allgather_slices(pipe.frame, NX, NY, bin_bounds)
    else:
        raise SystemExit("unknown mode " + mode)
    f = pipe.frame
    np.savez(os.path.join(directory, "%s_rank%d.npz" % (mode, rank)), mean=np.array(f.mean), variance=np.array(f.variance), samples=np.array(f.samples))
    comm.barrier()
comm.close()
print("rank %d of %d: %s done" % (rank, world_size, ",".join(modes)))
