#!/usr/bin/env python3
"""Wall time of every call of a loop of small path-traced calls (Cornell box 256 x 256, 4 spp per pass, K passes per call), device
synchronised after each: which calls are slow? usage: python tools/r5_path_batches_diag.py [K,K,...] [calls] [timing 0|1]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from source_amd import api as ns, scenes  # noqa: E402
from source_amd.device import get_context  # noqa: E402

KS = [int(k) for k in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 4]
CALLS = int(sys.argv[2]) if len(sys.argv) > 2 else 24
TIMING = len(sys.argv) > 3 and sys.argv[3] == "1"
import gc  # noqa: E402
MODE = os.environ.get("DIAG_GC", "")
if MODE == "off":
    gc.disable()
for K in KS:
    if MODE == "collect":                                   # everything of the previous phase goes now, timed
        world = cam = pipe = None
        t0 = time.perf_counter()
        n = gc.collect()
        get_context().synchronize()
        print("gc.collect: %d objects, %.1f ms" % (n, (time.perf_counter() - t0) * 1e3))
    world, _ = scenes.build_cornell(ns)
    cam, pipe = scenes.cornell_camera(ns, world, (256, 256), spp=4, bins=15)
    cam.frame_sampler = ns.RectFrameSampler2D()
    cam.render_engine = ns.HipEngine(rng="philox", seed=5, passes_per_call=K, auto_batch=False, timing=TIMING)
    world.build_accelerator()
    out = []
    for _ in range(CALLS):
        t0 = time.perf_counter()
        cam.observe()
        get_context().synchronize()
        out.append("%.1f" % ((time.perf_counter() - t0) * 1e3))
    print("K=%d timing=%d ms per call: " % (K, TIMING) + " ".join(out), flush=True)
