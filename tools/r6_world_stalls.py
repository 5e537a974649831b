#!/usr/bin/env python3
"""Reproducer of round 5's open fault (DESIGN 8.7 there): "a process that builds a second world meets two or three ~80 ms device stalls in the
next few hundred milliseconds". Per-call wall time of a render loop over a FIRST, a SECOND and a THIRD world built in one process, the device
synchronised after every call; the trigger between the phases can be narrowed (third argument) — a new world, only a new device scene over the
same world, only a new camera / frame / engine, only device allocations and uploads, nothing.

What it found (profiles/r06_world_stalls.txt, the rocprofv3 --hip-trace timeline in profiles/r06_world_stalls_timeline.txt): the trigger is the
device SCENE (rsx_scene_create after FlatScene's KD builds), the lost time sits in whatever host call waits next (hipStreamSynchronize,
hipMemcpyAsync) with the device idle behind a finished copy kernel, it comes in the 100 ms rhythm of the kernel's CPU bandwidth control — and
it disappears with KMP_BLOCKTIME=0, OMP_WAIT_POLICY=passive or OMP_NUM_THREADS=8. Cause: the container sees 256 hardware threads but owns a
16-core cgroup quota (cpu.max 1600000 100000); rsx_kd_build's `#pragma omp parallel` started 256 threads and LLVM's OpenMP runtime keeps
finished workers spinning for 200 ms: they burnt each following period's quota in milliseconds and the kernel froze the whole process —
the thread that feeds the GPU included — for the rest of the period. Fixed in csrc/rsx_host.cpp (host_team_size: team <= affinity and
quota, kmp_set_blocktime(0)). `RSX_HOST_THREADS=256 RSX_HOST_SPIN=1` brings the fault back.
usage: python tools/r6_world_stalls.py [calls per world] [scene: cornell|c2] [trigger: world|scene|frame|engine|pipe|release|upload3|alloc|gc|none]
env WORLD_KEEP=1 keeps the earlier worlds alive"""
import glob
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from source_amd import api as ns, scenes  # noqa: E402
from source_amd.device import get_context  # noqa: E402

CALLS = int(sys.argv[1]) if len(sys.argv) > 1 else 60
SCENE = sys.argv[2] if len(sys.argv) > 2 else "cornell"
KEEP = os.environ.get("WORLD_KEEP", "0") == "1"
TRIGGER = sys.argv[3] if len(sys.argv) > 3 else "world"      # world: a new world per phase | frame: a new camera + frame over the first world | alloc: a device allocation + upload only | none


def cpu_throttled():
    """(periods in which the cgroup was throttled, microseconds it spent throttled) from cpu.stat — the evidence for the cause"""
    out = {}
    for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            for line in open(path):
                k, v = line.split()
                out[k] = int(v)
            break
        except Exception:
            continue
    return out.get("nr_throttled", 0), out.get("throttled_usec", out.get("throttled_time", 0) // 1000)


kept = []
t_origin = time.perf_counter()
print("cgroup cpu.max:", (open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "?"), "| hardware threads visible:", os.cpu_count())
world = cam = pipe = None
for w in range(3):
    t0 = time.perf_counter()
    if w > 0 and TRIGGER in ("engine", "pipe", "release", "gc", "upload3", "scene"):
        ctx = get_context()
        if TRIGGER == "engine":
            cam.render_engine = ns.HipEngine(rng="philox", seed=5, auto_batch=False)
        elif TRIGGER == "pipe":                       # a new pipeline (hence a new frame: three device arrays + three host arrays) on the same camera
            pipe = ns.SpectralRadiancePipeline2D()
            cam.pipelines = [pipe]
        elif TRIGGER == "release":                    # the frame's device copy is dropped and made again (rsx_dev_free x 3, rsx_dev_alloc x 3, uploads)
            pipe.frame.release()
        elif TRIGGER == "scene":                      # the same world, the same camera: only a new device scene (rsx_scene_create; the old one is freed)
            world.build_accelerator(force=True)
        elif TRIGGER == "gc":
            import gc
            gc.collect()
        elif TRIGGER == "upload3":                    # three fresh host arrays of the frame's size uploaded into fresh device blocks, the old ones freed
            import numpy as np
            for p in kept:
                ctx.free(p)
            del kept[:]
            for dt in (np.float64, np.float64, np.int32):
                a = np.zeros((256, 256, 15), dtype=dt)
                p = ctx.alloc(a.nbytes)
                ctx.upload(p, a)
                kept.append(p)
    elif w > 0 and TRIGGER in ("alloc", "none", "frame"):
        if TRIGGER == "alloc":
            import numpy as np
            ctx = get_context()
            buf = np.zeros(8 << 20)
            p = ctx.alloc(buf.nbytes)
            ctx.upload(p, buf)
            kept.append(p)
        elif TRIGGER == "frame":
            if SCENE == "cornell":
                cam, pipe = scenes.cornell_camera(ns, world, (256, 256), spp=4, bins=15)
            else:
                cam, pipe = scenes.c2_camera(ns, world, (512, 512), spp=4, bins=15)
            cam.frame_sampler = ns.RectFrameSampler2D()
            cam.render_engine = ns.HipEngine(rng="philox", seed=5, auto_batch=False)
    elif SCENE == "cornell":
        world = scenes.build_cornell(ns)[0]
        cam, pipe = scenes.cornell_camera(ns, world, (256, 256), spp=4, bins=15)
    else:
        world = scenes.build_c2(ns, n=48)[0]
        cam, pipe = scenes.c2_camera(ns, world, (512, 512), spp=4, bins=15)
    if w == 0 or TRIGGER == "world":
        cam.frame_sampler = ns.RectFrameSampler2D()
        cam.render_engine = ns.HipEngine(rng="philox", seed=5, auto_batch=False)
    world.build_accelerator()
    get_context().synchronize()
    print("world %d built in %.1f ms" % (w + 1, (time.perf_counter() - t0) * 1e3), flush=True)
    times, stalls = [], []
    for k in range(CALLS):
        e0 = cpu_throttled()
        t0 = time.perf_counter()
        cam.observe()
        get_context().synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        times.append(dt)
        e1 = cpu_throttled()
        if dt > 20.0:
            stalls.append("call %d at +%.0f ms: %.1f ms wall; the cgroup was throttled in %d period(s) for %.1f ms meanwhile" % (k, (t0 - t_origin) * 1e3, dt, e1[0] - e0[0], (e1[1] - e0[1]) / 1e3))
    s = sorted(times)
    print("world %d: median %.2f ms, max %.1f ms per call; %d calls over 20 ms" % (w + 1, s[len(s) // 2], s[-1], len(stalls)))
    for line in stalls:
        print("    " + line)
    if KEEP:
        kept.append((world, cam, pipe))
print("cgroup throttling over the whole run: %d periods, %.1f ms" % (cpu_throttled()[0], cpu_throttled()[1] / 1e3))
