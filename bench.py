#!/usr/bin/env python3
"""
bench.py — primary rays/s of the hot path on MI355X (BASELINE.json metric), with the kernel's roofline position
and the CPU baseline timed in the same run.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (default, --workload c3): BASELINE.json configs[2], the configuration the metric is quoted on — the ~1M-triangle
instanced scene (15 instances of the 69 432-triangle Stanford-bunny stand-in of source_amd/scenes.py + floor box),
PinholeCamera 2048x2048, 64 samples/pixel/pass, 15 spectral bins, primary rays only (closed-form materials).
One "step" = one observe() pass over the whole frame = 268 435 456 primary rays: ray generation (Philox jitter) ->
two-level KD traversal + watertight triangle tests -> shading -> per-pixel/bin Welford over the 64 samples, merged into the
device-resident spectral frame. Scene, camera tables and the frame are resident in HBM when the timed region starts.
--workload c2 = configs[1] (single 69 432-triangle mesh, 1024x1024, 1 spp/pass); --workload c4 = configs[3] (demos/csg.py
tree, 1024x1024, 16 spp/pass); --workload flat = one 1M-triangle mesh without instancing (geometry far larger than L2).

N>1: sample sharding — every rank renders the same frame with its own sample counters (weak scaling: per-GPU work is
fixed); the only collective is one RCCL all_gather of the (mean, variance, samples) frames after the K passes, followed
by the combine_samples merge on every rank. That collective is inside the timed region.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

PEAK_HBM_GBS = 8000.0       # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
BINS = 15
WORKLOADS = {
    "c3": dict(nx=2048, ny=2048, spp=64, counter_rows=32,
               name="configs[2]: 1 041 480-triangle instanced scene (15 instances of the 69 432-triangle Stanford-bunny stand-in + floor "
                    "box), PinholeCamera 2048x2048, 64 spp/pass, 15 spectral bins, primary rays only, 1 MI355X per rank"),
    "c2": dict(nx=1024, ny=1024, spp=1, counter_rows=8,
               name="configs[1]: 69 432-triangle displaced-sphere mesh (Stanford-bunny stand-in), PinholeCamera 1024x1024, 1 spp/pass, "
                    "15 spectral bins, primary rays only, 1 MI355X per rank"),
    "flat": dict(nx=2048, ny=2048, spp=64, counter_rows=32,
                 name="HBM stress (SURVEY.md 8d M1M-flat): ONE 1 047 552-triangle displaced-sphere mesh (no instancing), PinholeCamera "
                      "2048x2048, 64 spp/pass, 15 spectral bins, primary rays only, 1 MI355X per rank"),
    "c4": dict(nx=1024, ny=1024, spp=16, counter_rows=8,
               name="configs[3]: demos/csg.py Boolean tree (sphere/box/cylinder Union/Intersect/Subtract, 5 CSG objects), PinholeCamera "
                    "1024x1024, 16 spp/pass, 15 spectral bins, primary rays only, 1 MI355X per rank"),
}


def build_workload(key, ns, scenes):
    w = WORKLOADS[key]
    if key == "c3":
        world = scenes.build_c3(ns, n=132)[0]
        cam, pipe = scenes.c3_camera(ns, world, (w["nx"], w["ny"]), spp=w["spp"], bins=BINS)
    elif key == "c2":
        world = scenes.build_c2(ns, n=132)[0]
        cam, pipe = scenes.c2_camera(ns, world, (w["nx"], w["ny"]), spp=w["spp"], bins=BINS)
    elif key == "flat":
        world = scenes.build_flat(ns, n=512)[0]
        cam, pipe = scenes.c2_camera(ns, world, (w["nx"], w["ny"]), spp=w["spp"], bins=BINS)
    else:
        world = scenes.build_csg_demo(ns)[0]
        cam, pipe = scenes.csg_camera(ns, world, (w["nx"], w["ny"]), spp=w["spp"], bins=BINS)
    return world, cam, pipe


def ray_bytes(counters, n_rays):
    """Algorithmic bytes per primary ray of the traversal kernel:
    56 (ray) + 16/KD node + 4/leaf item + 48/triangle test + 216/world-leaf primitive test + 24 (sample record)."""
    per = {k: v / n_rays for k, v in counters.items()}
    b = 56 + 16 * per["nodes"] + 4 * per["items"] + 48 * per["tris"] + 216 * per["prims"] + 24
    return b, per


def _lib_combine(ctx, m, v, n, mb, vb, nb):
    """combine_samples law applied in place on the device (rsx_frame_combine_dev) to torch tensors."""
    from source_amd import _lib
    _lib.check(_lib.lib().rsx_frame_combine_dev(ctx.handle, m.numel(), m.data_ptr(), v.data_ptr(), n.data_ptr(),
                                                mb.data_ptr(), vb.data_ptr(), nb.data_ptr()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU work the cpu_baseline sample is sized for")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c3")
    args = ap.parse_args()
    W = WORKLOADS[args.workload]
    NX, NY, SPP = W["nx"], W["ny"], W["spp"]

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    torch = None
    if args.gpus > 1 or world_size > 1 or "RANK" in os.environ:      # launched by torch.distributed.run: take the collective path even for 1 rank
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        world_size = dist.get_world_size()
    os.environ["RSX_DEVICE"] = str(local_rank)

    import __graft_entry__ as ge
    ge.build_librsx()                                      # no-op when the in-tree .so is fresh
    from source_amd import api as ns, scenes
    from source_amd.device import get_context
    from source_amd.distributed import rank_sample_offset

    world, cam, pipe = build_workload(args.workload, ns, scenes)
    cam.frame_sampler = ns.RectFrameSampler2D()
    engine = ns.HipEngine(rng="philox", seed=20250905, timing=False)
    cam.render_engine = engine
    ctx = get_context()
    scene = world.build_accelerator()                      # flatten + KD build (host) + upload: outside the timed region

    frames = None
    if dist is not None:
        # frame storage = torch tensors so RCCL can move them; librsx writes through the raw device pointers
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        frames = [torch.zeros((NX, NY, BINS), dtype=torch.float64, device="cuda"),
                  torch.zeros((NX, NY, BINS), dtype=torch.float64, device="cuda"),
                  torch.zeros((NX, NY, BINS), dtype=torch.int32, device="cuda")]

    def bind():
        if frames is not None:
            pipe.frame.bind_device(ctx, *(f.data_ptr() for f in frames))

    step_counter = [0]

    def step():
        engine.sample_offset = rank_sample_offset(step_counter[0], rank, world_size, SPP)
        step_counter[0] += 1
        cam.observe()
        bind()

    def sync():
        if torch is not None:
            torch.cuda.synchronize()
        ctx.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()

    # first observe() creates the pipeline frame; bind external storage before anything is rendered into it
    pipe.initialise((NX, NY), SPP, cam.min_wavelength, cam.max_wavelength, BINS, cam._slice_spectrum(), True)
    bind()
    # Pre-warm: a freshly loaded MI355X takes one ~75 ms hit some tens of ms after sustained work starts (clock / power-state
    # transition; measured as a single stalled kernel in otherwise 0.6 ms passes). Run ~0.4 s of untimed passes so that it lands
    # here and not inside the W warmup or K timed steps. These passes are ordinary passes into the same accumulating frame.
    prewarm = 0
    t_pre = time.perf_counter()
    burst = 16 if NX * NY * SPP < (1 << 24) else 1
    while time.perf_counter() - t_pre < 0.4:
        for _ in range(burst):
            step()
        sync()
        prewarm += burst
    for _ in range(args.warmup):
        step()
    if dist is not None:
        # untimed: the first all_to_all / all_gather of a process group sets up its RCCL channels and peer connections (tens of
        # ms); run the frame merge once on small dummy tensors so that this one-time cost is not charged to the timed steps
        from source_amd import distributed as D0
        tiny = [torch.zeros(4096, dtype=torch.float64, device="cuda"), torch.zeros(4096, dtype=torch.float64, device="cuda"),
                torch.ones(4096, dtype=torch.int32, device="cuda")]
        D0.merge_sample_sharded(tiny[0], tiny[1], tiny[2], dist,
                                lambda m, v, n, mb, vb, nb: _lib_combine(ctx, m, v, n, mb, vb, nb))
    sync()
    barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    collective_ms = 0.0
    if dist is not None:
        sync()
        tc = time.perf_counter()
        from source_amd import distributed as D
        merged = D.merge_sample_sharded(frames[0], frames[1], frames[2], dist,
                                        lambda m, v, n, mb, vb, nb: _lib_combine(ctx, m, v, n, mb, vb, nb))
        sync()
        collective_ms = (time.perf_counter() - tc) * 1e3
    sync()
    barrier()
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    rays_per_step = NX * NY * SPP
    total_rays = rays_per_step * args.steps * world_size
    value = total_rays / elapsed

    # per-launch kernel durations of the timed steps (HIP events recorded by librsx on its launch stream)
    n_hist = min(args.steps, 512)
    trace_ms, accum_ms = ctx.render_history(n_hist)
    trace_avg, accum_avg = float(np.mean(trace_ms)), float(np.mean(accum_ms))

    out = None
    if rank == 0:
        # sanity of the rendered frame (rank 0's own frame)
        samples = pipe.frame.samples
        mean = pipe.frame.mean
        assert int(samples.min()) == int(samples.max()) == (prewarm + args.warmup + args.steps) * SPP, "frame sample count mismatch"
        assert np.isfinite(mean).all() and mean.max() > 0

        from oracle import oracle as orc
        flat = scene.flat
        # mean per-ray traversal counters on every k-th row of the same camera (pixel-centre rays), instrumented oracle
        rows = np.arange(0, NY, W["counter_rows"])
        tasks = np.array([(ix, iy) for iy in rows for ix in range(NX)], dtype=np.int32)
        from source_amd import _lib
        d2 = _lib.RenderDesc()
        d2.camera = cam.device_camera()
        u = np.full(2 * len(tasks), 0.5)
        d2.tasks, d2.n_tasks, d2.spp, d2.uniforms = _lib.ptr(tasks), len(tasks), 1, _lib.ptr(u)
        rays = orc.pinhole_rays(d2)
        nthreads = orc.max_threads()
        cnt = orc.hit_batch(flat, rays[:, 0:3], rays[:, 3:6], None, threads=nthreads, counters=True)["counters"]
        b_ray, per_ray = ray_bytes(cnt, len(tasks))
        achieved = b_ray * rays_per_step / (trace_avg * 1e-3) / 1e9
        traffic = None
        import glob
        pmcs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_%s.json" % args.workload)))   # latest committed rocprofv3 --pmc summary
        if pmcs:
            traffic = json.load(open(pmcs[-1])).get("hbm_bytes_per_launch")
        roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(achieved / PEAK_HBM_GBS, 5), "traffic": traffic, "kernel": "k_render_trace",
                    "kernel_ms": round(trace_avg, 4), "accumulate_kernel_ms": round(accum_avg, 4),
                    "bytes_per_ray": round(b_ray, 1), "per_ray": {k: round(v, 3) for k, v in per_ray.items()}}

        cpu = None
        if not args.no_cpu_baseline and world_size == 1:       # the CPU baseline is timed at N = 1 only (the other ranks would idle at the barrier)
            # bounded sample of the same workload on the host cores, same Philox samples: calibrate on a 16-row band, then size
            # the sample for ~args.cpu_seconds of CPU work — whole passes when one fits, otherwise a centred band of rows
            keep = []
            sl = cam._slice_spectrum()[0]
            engine.sample_offset = 0
            band = (0, NY // 2 - 8, NX, NY // 2 + 8)
            desc = cam.render_desc(world, None, sl, engine, keep, rect=band)
            orc.render_pinhole(flat, desc, threads=nthreads)        # thread-pool warm-up
            tcal = time.perf_counter()
            m, v, nr = orc.render_pinhole(flat, desc, threads=nthreads)
            rate = nr / (time.perf_counter() - tcal)
            target = rate * args.cpu_seconds
            saved = cam.pixel_samples
            if target >= rays_per_step:
                passes = int(min(2048, max(1, round(target / rays_per_step))))
                cam.pixel_samples = SPP * passes
                rect = (0, 0, NX, NY)
                what = "%d full %dx%d passes of %d spp" % (passes, NX, NY, SPP)
            else:
                nrows = int(max(16, min(NY, target // (NX * SPP))))
                rect = (0, NY // 2 - nrows // 2, NX, NY // 2 - nrows // 2 + nrows)
                what = "centred band of %d of %d rows x %d px x %d spp" % (nrows, NY, NX, SPP)
            desc = cam.render_desc(world, None, sl, engine, keep, rect=rect)
            cam.pixel_samples = saved
            tcpu = time.perf_counter()
            m, v, nr = orc.render_pinhole(flat, desc, threads=nthreads)
            tcpu = time.perf_counter() - tcpu
            assert np.isfinite(m).all()
            cpu = {"value": round(nr / tcpu, 1), "unit": "primary rays/s", "cores": nthreads, "kind": "port",
                   "sample": "%s of the same workload (%d rays), oracle/rsx_oracle.c (C restatement of the reference algorithm) "
                             "with OpenMP on %d host threads, %.1f s" % (what, nr, nthreads, tcpu)}

        out = {
            "metric": "primary rays/sec", "value": round(value, 1), "unit": "rays/s", "n_gpus": world_size,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": W["name"], "rays_per_step_per_gpu": rays_per_step, "rng": "philox4x32-10", "sharding": "sample" if world_size > 1 else "none",
                       "collective_ms": round(collective_ms, 3)},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
