#!/usr/bin/env python3
"""Renders only a silhouette strip of configs[1] (the slowest tiles) a few times: PMC target for tail analysis."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from source_amd import api as ns, scenes
from source_amd.device import get_context
world = scenes.build_c2(ns, n=132)[0]
cam, pipe = scenes.c2_camera(ns, world, (1024, 1024), spp=1, bins=15)
rect = tuple(int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (416, 232, 536, 256)))
cam.frame_sampler = ns.RectFrameSampler2D(rect)
cam.render_engine = ns.HipEngine(rng="philox", seed=20250905, timing=False)
ctx = get_context()
world.build_accelerator()
for _ in range(6): cam.observe()
tr, ac = ctx.render_history(4)
print(json.dumps({"rect": rect, "tiles": (rect[2]-rect[0])*(rect[3]-rect[1])//64, "trace_ms": [round(float(x), 4) for x in tr]}))
