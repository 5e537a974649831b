#!/usr/bin/env python3
"""Coarse phase profile of the path kernel (GPU box; needs a librsx built with -DRSX_PHASE_PROF=3, pass it as $RSX_LIB):
tools/path_prof.py [cornell|lambert|lambert_plain|glass|prism] — s_memtime per phase of every segment round, summed over the waves."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from source_amd import api as ns, scenes, _lib  # noqa: E402
from source_amd.device import get_context  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cornell"
if cfg == "cornell":
    world = scenes.build_cornell(ns)[0]
    cam, pipe = scenes.cornell_camera(ns, world, (1024, 1024), 16, 15)
elif cfg == "prism":
    world = scenes.build_prism(ns)[0]
    cam, pipe = scenes.prism_camera(ns, world, (1024, 1024), int(sys.argv[2]) if len(sys.argv) > 2 else 4, 1, 1)
elif cfg == "glass":
    world = scenes.build_glass(ns)[0]
    cam, pipe = scenes.glass_camera(ns, world, (1024, 1024), 16, 5, 1, (0.01, 3, 500))
else:
    world = scenes.build_lambert(ns, with_volume=cfg == "lambert", csg=cfg == "lambert")[0]
    cam, pipe = scenes.lambert_camera(ns, world, (1024, 1024), 16, 15, (0.01, 3, 500))
cam.frame_sampler = ns.RectFrameSampler2D()
cam.render_engine = ns.HipEngine(rng="philox", seed=20250905)
ctx = get_context()
world.build_accelerator()
cam.observe()                                            # warm-up (arena sizing)
buf = ctx.alloc(64 * 8)
ctx.memset(buf, 0, 64 * 8)
_lib.check(_lib.lib().rsx_debug_unit_times(ctx.handle, buf))
cam.observe()
ctx.synchronize()
_lib.check(_lib.lib().rsx_debug_unit_times(ctx.handle, None))
c = np.zeros(64, dtype=np.uint64)
ctx.download(c, buf)
c = c.astype(np.float64)
tot = c[:5].sum()
for name, v in zip(("refill / ray generation", "world_trace_wave", "hit geometry", "volume pass (world.contains)", "material + bookkeeping"), c[:5]):
    print("%-30s %.3f" % (name, v / tot))
print("segment rounds %.4g, live lanes per round %.1f" % (c[7], c[6] / max(c[7], 1)))
print("inside world_trace_wave: descents %.3f of it; %.1f leaf-visit rounds per segment round" % (c[5] / max(c[1], 1), c[8] / max(c[7], 1)))
print("  before the walk: gates and wave-uniform answers %.3f of it, per-lane box rounds %.3f of it (%.2f rounds per segment round, %.1f lanes per round)" % (
    c[9] / max(c[1], 1), c[10] / max(c[1], 1), c[11] / max(c[7], 1), c[12] / max(c[11], 1)))
if c[36] + c[44] > 0:                                    # the level-by-level form (dev_wavefront.hpp) rendered the pass
    for name, o in (("level 0", 40), ("levels >= 1", 32)):
        t = c[o:o + 4].sum()
        if t > 0:
            print("%-12s iterations %.4g, live lanes %.1f, lanes that go on %.1f; clocks per iteration %.0f: entry + path record %.3f, arm %.3f, walk %.3f, filing %.3f" % (
                name, c[o + 4], c[o + 5] / max(c[o + 4], 1), c[o + 6] / max(c[o + 4], 1), t / max(c[o + 4], 1), c[o] / t, c[o + 1] / t, c[o + 2] / t, c[o + 3] / t))
# (the accumulators are lane 0's: inside a divergent arm they count the rounds in which lane 0 took the arm — the time is scaled by that share)
la, di = c[14] / max(c[7], 1), c[16] / max(c[7], 1)
print("  inside material + bookkeeping: lane 0 is in the Lambert arm in %.2f of the rounds, in the Dielectric arm in %.2f" % (la, di))
print("    Lambert arm ~%.2f of it, Dielectric arm ~%.2f, roulette and term ~%.2f, before the arms (scattering draw, first use of the material record) %.2f" % (
    c[13] / max(c[4], 1) / max(la, 1e-9), c[15] / max(c[4], 1) / max(di, 1e-9), c[17] / max(c[4], 1) / max(la + di, 1e-9), c[18] / max(c[4], 1)))
if c[20] > 0 and c[8] == 0:                              # packed CSG prefill (no leaf visits: the scene is answered before the walk)
    print("CSG prefill, packed: %.2f questions per round for %.1f live lanes, %.2f turns of 64" % (c[21] / c[20], c[23] / c[20], c[22] / c[20]))
elif c[20] > 0:                                          # CSG prefill round: how many lanes ask about a solid when the wave evaluates it
    print("CSG prefill: %.2f solids per segment round offered, %.2f evaluated; asking lanes per evaluation %.1f of %.1f live" % (
        c[20] / max(c[7], 1), c[22] / max(c[7], 1), c[21] / max(c[22], 1), c[23] / max(c[20], 1)))
