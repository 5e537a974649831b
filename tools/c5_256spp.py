#!/usr/bin/env python3
"""BASELINE configs[4] as written, once: demos/prism.py's dispersive-prism scene, 1024 x 1024, 512 spectral bins rendered as 512 one-bin
slices, 256 samples per pixel — accumulated as PASSES passes of SPP samples (observer.pyx:265-340 called PASSES times into an
accumulating pipeline, power.pyx:399-437) into the 10.7 GB device-resident frame. Afterwards three slice strips are rendered by the
oracle with the SAME Philox counters, pass by pass, merged with the combine_samples law, and compared: sample counts exact, mean
within 1e-12 relative, variance within 16 eps (mean^2 + var) per merged pass, and the z-score of the difference of the means
against the frame's own standard error (SURVEY 8d asks z <= 4; the counters being the same, it is ~0).
usage (GPU box): python tools/c5_256spp.py [passes] [spp] > gpurun_out/c5_256spp.log"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc                            # noqa: E402  (checker, after the timed passes)
from source_amd import api as ns, scenes                    # noqa: E402
from source_amd import distributed as D                     # noqa: E402
from source_amd.device import get_context                   # noqa: E402

PASSES = int(sys.argv[1]) if len(sys.argv) > 1 else 16
SPP = int(sys.argv[2]) if len(sys.argv) > 2 else 16
NX = NY = int(os.environ.get("C5_PIXELS", "1024"))
BINS = int(os.environ.get("C5_BINS", "512"))
world, prims = scenes.build_prism(ns)
cam, pipe = scenes.prism_camera(ns, world, (NX, NY), SPP, BINS, BINS)
cam.frame_sampler = ns.RectFrameSampler2D()
eng = ns.HipEngine(rng="philox", seed=29)
cam.render_engine = eng
ctx = get_context()
times = []
t_all = time.perf_counter()
for p in range(PASSES):
    t0 = time.perf_counter()
    cam.observe()
    ctx.synchronize()
    times.append(time.perf_counter() - t0)
    print("pass %3d: %.3f s  (%d rays so far this pass)" % (p, times[-1], cam.stats["rays"]), flush=True)
elapsed = time.perf_counter() - t_all
f = pipe.frame
n = f.samples
assert f.shape == (NX, NY, BINS) and (n == PASSES * SPP).all(), "sample counts"
mean, var = f.mean, f.variance
assert np.isfinite(mean).all() and np.isfinite(var).all() and (var >= 0).all()
paths = PASSES * SPP * NX * NY * BINS
print("configs[4] as written: %d passes x %d spp x %d slices x %dx%d = %.4g paths in %.1f s (%.3g paths/s); per pass: median %.3f s, first %.3f s"
      % (PASSES, SPP, BINS, NX, NY, paths, elapsed, paths / elapsed, float(np.median(times)), times[0]), flush=True)
flat = world.flatten()
slices = cam._slice_spectrum()
row0 = int(0.586 * NY)
rect = (0, row0, NX, row0 + 6)
eps = np.finfo(np.float64).eps
worst = {}
for k in (3, BINS // 2, BINS - 4):
    om = ov = on = None
    for p in range(PASSES):
        keep = []
        desc = cam.render_desc(world, None, slices[k], eng, keep, rect=rect, sample_offset=p * SPP)
        m, v, rays = orc.render_pinhole(flat, desc, threads=min(16, orc.max_threads()))
        m, v = m.reshape(6, NX).T, v.reshape(6, NX).T
        cnt = np.full(m.shape, SPP, dtype=np.int32)
        if om is None:
            om, ov, on = m, np.maximum(v, 0.0), cnt
        else:
            om, ov, on = D.combine_arrays(om, ov, on, m, np.maximum(v, 0.0), cnt)
    dm, dv = mean[:, row0:row0 + 6, k], var[:, row0:row0 + 6, k]
    assert (on == PASSES * SPP).all()
    rel = float(np.max(np.abs(dm - om) / np.maximum(np.abs(om), 1e-300)))
    vb = float(np.max(np.abs(dv - ov) / (16 * eps * (om * om + ov) * PASSES + 1e-300)))
    se = np.sqrt(np.maximum(ov, 0) / (PASSES * SPP))
    z = float(np.max(np.where(se > 0, np.abs(dm - om) / np.maximum(se, 1e-300), 0.0)))
    lit = int((om > 0).sum())
    worst[k] = dict(mean_rel=rel, variance_over_bound=vb, z_max=z, lit_pixels=lit, bit_identical_mean=bool(np.array_equal(dm, om)))
    print("slice %3d strip rows %d-%d: mean rel err %.3g, variance err / bound %.3g, max z %.3g, lit pixels %d, mean bit-identical %s"
          % (k, row0, row0 + 6, rel, vb, z, lit, np.array_equal(dm, om)), flush=True)
    assert rel <= 1e-12 and vb <= 1.0 and z <= 4.0, k
print(json.dumps({"passes": PASSES, "spp": SPP, "slices": BINS, "pixels": [NX, NY], "paths": paths, "seconds": round(elapsed, 2),
                  "paths_per_s": round(paths / elapsed, 1), "seconds_per_pass_median": round(float(np.median(times)), 4), "checks": worst}))
