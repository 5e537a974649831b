#!/usr/bin/env python3
"""rsx_hit_batch throughput on scattered rays (GPU box): random origins inside the scene bounds, random directions."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from source_amd import api as ns, scenes  # noqa: E402

for name, build in (("cornell", scenes.build_cornell), ("lambert", scenes.build_lambert), ("c3", lambda n: scenes.build_c3(n, n=132))):
    world = build(ns)[0]
    scene = world.build_accelerator()
    kd = scene.flat.world_kd
    rng = np.random.RandomState(1)
    lo, hi = np.array(kd.lower), np.array(kd.upper)
    n = 1 << 22
    o = lo + (hi - lo) * rng.uniform(0.2, 0.8, (n, 3))
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1)[:, None]
    scene.hit_batch(o[:1000], d[:1000])
    t0 = time.perf_counter()
    r = scene.hit_batch(o, d)
    dt = time.perf_counter() - t0
    from source_amd.device import get_context
    print("%-8s %d rays: %.1f ms end to end (host buffers), kernel %.2f ms, hits %d, checksum %.6f" % (
        name, n, dt * 1e3, get_context().last_kernel_ms(), (r["prim"] >= 0).sum(), float(np.nansum(r["t"][r["prim"] >= 0]))))
