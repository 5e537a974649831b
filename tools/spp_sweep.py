#!/usr/bin/env python3
"""Rays/s of one scene at several samples per pixel (GPU box): tools/spp_sweep.py [c2|c3|c4] — run it under RSX_PACKET_MIN_SPP=64 / 2 to
compare the per-lane and the packet walk pass by pass."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from source_amd import api as ns, scenes  # noqa: E402
from source_amd.device import get_context  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "c4"
for spp in (1, 2, 4, 8, 16, 32):
    if which == "c3":
        world = scenes.build_c3(ns, n=132)[0]
        cam, pipe = scenes.c3_camera(ns, world, (1024, 1024), spp=spp, bins=15)
    elif which == "c4":
        world = scenes.build_csg_demo(ns)[0]
        cam, pipe = scenes.csg_camera(ns, world, (1024, 1024), spp=spp, bins=15)
    else:
        world = scenes.build_c2(ns, n=132)[0]
        cam, pipe = scenes.c2_camera(ns, world, (1024, 1024), spp=spp, bins=15)
    cam.frame_sampler = ns.RectFrameSampler2D()
    cam.render_engine = ns.HipEngine(rng="philox", seed=1)
    ctx = get_context()
    t_warm = time.perf_counter()                            # (0.6 s of passes first: a freshly loaded GPU stalls once for ~75 ms while its clocks ramp)
    while time.perf_counter() - t_warm < 0.6:
        for _ in range(20):
            cam.observe()
        ctx.synchronize()
    t = time.perf_counter()
    n = 60
    for _ in range(n):
        cam.observe()
    ctx.synchronize()
    dt = (time.perf_counter() - t) / n
    print(which, "spp %2d" % spp, "RSX_PACKET_MIN_SPP", os.environ.get("RSX_PACKET_MIN_SPP"), "%.3f ms  %.3e rays/s" % (dt * 1e3, 1024 * 1024 * spp / dt))
