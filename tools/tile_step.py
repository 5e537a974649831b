#!/usr/bin/env python3
"""Step time of one rank's tile of configs[2] on one GPU (GPU box): tools/tile_step.py N rank [steps] — what a rank of an N-way
tile-sharded run does between two frame exchanges."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from source_amd import api as ns, scenes
from source_amd import distributed as D
from source_amd.device import get_context
N, r = int(sys.argv[1]), int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
world = scenes.build_c3(ns, n=132)[0]
cam, pipe = scenes.c3_camera(ns, world, (2048, 2048), spp=64, bins=15)
cam.render_engine = ns.HipEngine(rng="philox", seed=1)
ctx = get_context()
world.build_accelerator()
cam.frame_sampler = ns.RectFrameSampler2D(rect=D.tile_rect(r, N, 2048, 2048))
for k in range(5): cam.observe()
ctx.synchronize()
t0 = time.perf_counter()
for k in range(steps): cam.observe()
ctx.synchronize()
print("N=%d rank %d: %.3f ms per step" % (N, r, (time.perf_counter() - t0) / steps * 1e3))
