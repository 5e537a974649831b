#!/usr/bin/env python3
"""The cpu_baseline leg of bench.py by itself (no GPU needed): host cores (affinity, cgroup quota, OpenMP default), the thread sweep and
the bounded sample, for one or more workloads. usage: tools/cpu_scaling.py [--seconds S] c3 c2 ...   (JSON lines on stdout)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                               # noqa: E402
from oracle import oracle as orc                           # noqa: E402
from source_amd import api as ns, scenes                   # noqa: E402

args = sys.argv[1:]
seconds = 6.0
if args and args[0] == "--seconds":
    seconds = float(args[1]); args = args[2:]
for wl in args or ["c3"]:
    W = bench.WORKLOADS[wl]
    world, cam, pipe = bench.build_workload(wl, ns, scenes)
    engine = ns.HipEngine(rng="philox", seed=20250905)
    cam.render_engine = engine
    flat = world.flatten()
    NX, NY = cam.pixels
    slices = len(cam._slice_spectrum())
    out = bench.cpu_baseline_port(orc, flat, world, cam, engine, NX, NY, cam.pixel_samples, slices, seconds, orc.max_threads())
    print(json.dumps({"workload": wl, "cpu_baseline": out}))
