#!/usr/bin/env python3
"""Where the host spends an observe() of configs[4]'s shape (prism, 1024^2, 512 one-bin slices x 1 spp): cProfile of one pass after a
warm-up pass (GPU box). tools/c5_hostprof.py [slices] [lanes]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
slices = int(sys.argv[1]) if len(sys.argv) > 1 else 512
from source_amd import api as ns, scenes  # noqa: E402
from source_amd.device import get_context  # noqa: E402

world = scenes.build_prism(ns)[0]
cam, pipe = scenes.prism_camera(ns, world, (1024, 1024), 1, slices, slices)
cam.frame_sampler = ns.RectFrameSampler2D()
cam.render_engine = ns.HipEngine(rng="philox", seed=20250905)
ctx = get_context()
world.build_accelerator()
cam.observe()
ctx.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
cam.observe()
t1 = time.perf_counter()
ctx.synchronize()
pr.disable()
t2 = time.perf_counter()
print("observe() returned after %.3f s, device idle after %.3f s" % (t1 - t0, t2 - t0))
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
