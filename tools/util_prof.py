#!/usr/bin/env python3
"""Lane-utilisation per loop level (GPU box; needs a librsx built with -DRSX_UTIL_PROF=1, pass it as $RSX_LIB):
tools/util_prof.py [c2|c3] — renders one pass with the per-unit counter buffer attached and prints active / total lane-slots of the
world loop, the mesh loop, the node steps and the leaf batches."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from source_amd import api as ns, scenes, _lib  # noqa: E402
from source_amd.device import get_context  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
if cfg == "c3":
    world = scenes.build_c3(ns, n=132)[0]
    cam, pipe = scenes.c3_camera(ns, world, (2048, 2048), spp=int(os.environ.get("SPP", "4")), bins=15)
else:
    world = scenes.build_c2(ns, n=132)[0]
    cam, pipe = scenes.c2_camera(ns, world, (1024, 1024), spp=1, bins=15)
cam.frame_sampler = ns.RectFrameSampler2D()
cam.render_engine = ns.HipEngine(rng="philox", seed=20250905, timing=False)
ctx = get_context()
world.build_accelerator()
n_units = cam.pixels[0] * cam.pixels[1] * cam.pixel_samples // 64
buf = ctx.alloc(n_units * 12 * 8)
ctx.memset(buf, 0, n_units * 12 * 8)
_lib.check(_lib.lib().rsx_debug_unit_times(ctx.handle, buf))
cam.observe()
ctx.synchronize()
_lib.check(_lib.lib().rsx_debug_unit_times(ctx.handle, None))
host = np.zeros((n_units, 12), dtype=np.uint64)
ctx.download(host, buf)
c = host[:, 3:11].astype(np.float64).sum(axis=0)
if os.environ.get("PHASES") == "2":  # -DRSX_PHASE_PROF=2: coarse s_memtime phases of a unit
    tot = c[6]
    print("units %d, mesh visits per unit %.2f" % (n_units, c[5] / n_units))
    print("ray generation      %.3f of the unit" % (c[0] / tot))
    print("world_trace_wave    %.3f" % (c[1] / tot))
    print("  mesh visits       %.3f   (setup %.3f, traversal loop %.3f)" % (c[2] / tot, c[3] / tot, (c[2] - c[3]) / tot))
    print("  world tree + gates %.3f   (world descents %.3f, wide-primitive answers %.3f, leaf items and the rest %.3f)" % (
        (c[1] - c[2]) / tot, c[4] / tot, c[7] / tot, (c[1] - c[2] - c[4] - c[7]) / tot))
    print("cost / shade        %.3f (before the record store)" % ((c[6] - c[0] - c[1]) / tot))
    sys.exit(0)
if os.environ.get("PHASES"):       # library built with -DRSX_PHASE_PROF=1: s_memtime cycles per phase of mesh_trace_wave
    total = (host[:, 1] - host[:, 0]).astype(np.float64).sum()       # 100 MHz ticks per unit, summed
    names = ["loop head", "descend", "small leaves", "big leaves (coop)", "pop"]
    tot = c[:5].sum()
    for nme, v in zip(names, c[:5]):
        print("%-18s %.3f of the mesh loop" % (nme, v / tot))
    print("mesh-loop iterations %.4g, lanes active per iteration %.1f, big-leaf lanes per iteration %.3f" % (c[5], c[7] / c[5], c[6] / c[5]))
    print("mesh loop = %.3f of the units' wall time (both in s_memtime / s_memrealtime ticks)" % (tot / total))
    sys.exit(0)
names = (("world loop", 0), ("mesh loop", 2), ("node steps", 4), ("leaf batches", 6))
if os.environ.get("UTIL") == "2":    # -DRSX_UTIL_PROF=2: the world level
    names = (("world loop (leaf visits)", 0), ("world leaf items (gates)", 2), ("world node steps", 4), ("analytic primitive tests", 6))
    print("per 64-ray unit: %.2f world leaf visits, %.2f item rounds, %.2f node steps, %.2f analytic tests" % tuple(c[k + 1] / 64 / n_units for k in (0, 2, 4, 6)))
for name, k in names:
    print("%-13s active lane-slots %.4g of %.4g  -> utilisation %.3f" % (name, c[k], c[k + 1], c[k] / max(c[k + 1], 1)))
