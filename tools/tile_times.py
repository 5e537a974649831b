#!/usr/bin/env python3
"""Per-tile timing of one configs[1] pass (GPU box): which 8x8 tiles are slow, how long the tail is."""
import ctypes as C, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from source_amd import api as ns, scenes, _lib
from source_amd.device import get_context
world = scenes.build_c2(ns, n=132)[0]
cam, pipe = scenes.c2_camera(ns, world, (1024, 1024), spp=1, bins=15)
cam.frame_sampler = ns.RectFrameSampler2D()
cam.render_engine = ns.HipEngine(rng="philox", seed=20250905, timing=False)
ctx = get_context()
world.build_accelerator()
for _ in range(8): cam.observe()
n_units = 128 * 128
buf = ctx.alloc(n_units * 96)
ctx.memset(buf, 0, n_units * 96)
_lib.check(_lib.lib().rsx_debug_unit_times(ctx.handle, buf))
cam.observe()
ctx.synchronize()
_lib.check(_lib.lib().rsx_debug_unit_times(ctx.handle, None))
t = np.zeros((n_units, 12), dtype=np.uint64)
ctx.download(t, buf)
start, end = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64)
t0 = start.min()
dur = (end - start) / 100.0          # us (100 MHz)
print("kernel span %.1f us; sum of unit durations %.1f us; mean %.2f us; median %.2f; p99 %.1f; max %.1f" %
      ((end.max() - t0) / 100.0, dur.sum(), dur.mean(), np.median(dur), np.percentile(dur, 99), dur.max()))
order = np.argsort(-dur)[:12]
for u in order:
    print("  tile (%3d,%3d): %.1f us, starts at %.1f us, wave %d" % (u % 128, u // 128, dur[u], (start[u] - t0) / 100.0, t[u, 2]))
waves = t[:, 2]
uw, cnt = np.unique(waves, return_counts=True)
busy = np.array([dur[waves == w].sum() for w in uw])
lastend = np.array([(end[waves == w].max() - t0) / 100.0 for w in uw])
print("waves %d: tiles/wave mean %.1f; busy us mean %.1f max %.1f; last-end us: p50 %.1f p90 %.1f max %.1f" %
      (len(uw), cnt.mean(), busy.mean(), busy.max(), np.median(lastend), np.percentile(lastend, 90), lastend.max()))
hist, edges = np.histogram((end - t0) / 100.0, bins=10)
print("unit end-time histogram (us):", [int(e) for e in edges], hist.tolist())

ph = t[:, 3:11].astype(np.float64)
if ph.sum() > 0:
    names = ["loop-top", "descend", "small-leaf", "coop/staged", "pop", "iterations", "big-leaf lanes", "active lanes"]
    tot = ph[:, :5].sum()
    print("phase share of mesh_trace_wave cycles (all tiles):", {n: round(float(ph[:, i].sum() / tot), 3) for i, n in enumerate(names[:5])})
    for u in order[:6]:
        c = ph[u]
        print("  tile (%3d,%3d): cycles descend %.0fk small-leaf %.0fk coop %.0fk pop %.0fk | iterations %d, big-leaf lane-visits %d, mean active lanes %.1f" %
              (u % 128, u // 128, c[1] / 1e3, c[2] / 1e3, c[3] / 1e3, c[4] / 1e3, c[5], c[6], c[7] / max(c[5], 1)))
