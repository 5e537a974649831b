#!/usr/bin/env python3
"""rsx_hit_batch on the CSG worlds (GPU box): demos/csg.py's world, the mixed world (F07) and the prism scene — 2^22 rays from a pinhole
grid and 2^22 scattered ones; kernel time of the call (fast pass + stream-merge pass) and a checksum.  Under
`rocprofv3 --kernel-trace --stats` the split between k_hit_batch<true, 1> and k_hit_batch<true, 2> shows."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np  # noqa: E402

import raysets  # noqa: E402
from source_amd import api as ns, scenes  # noqa: E402
from source_amd.device import get_context  # noqa: E402

for name, build, origin, extent in (("csg_demo", scenes.build_csg_demo, (0.0, 0.0, -4.0), 4.5), ("mixed", scenes.build_mixed, (0.0, 0.0, -5.0), 2.2),
                                    ("prism", scenes.build_prism, (0.0, 0.5, -3.0), 2.0)):
    world = build(ns)[0]
    scene = world.build_accelerator()
    for kind in ("grid", "scattered"):
        if kind == "grid": o, d, m = raysets.pinhole_grid(2048, origin, 60.0)
        else: o, d, m = raysets.scene_rays(1 << 22, 7, 2.0 * extent, extent)
        scene.hit_batch(o[:4096], d[:4096])
        best = 1e9
        for rep in range(3):
            r = scene.hit_batch(o, d)
            best = min(best, get_context().last_kernel_ms())
        hit = r["prim"] >= 0
        print("%-9s %-9s %8d rays: kernels %.3f ms (%.3g rays/s), hits %d, sum t %.9g" % (name, kind, len(o), best, len(o) / best * 1e3, int(hit.sum()), float(np.sum(r["t"][hit]))))
