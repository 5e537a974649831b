import sys, time; sys.path.insert(0, '/root/repo')
import numpy as np
from source_amd import api as ns, scenes
world = scenes.build_c3(ns, n=132)[0]
cam, pipe = scenes.c3_camera(ns, world, (8192, 8192), spp=16, bins=15)
cam.frame_sampler = ns.RectFrameSampler2D()
cam.render_engine = ns.HipEngine(rng="philox", seed=4, timing=False)
t0 = time.perf_counter(); cam.observe()
from source_amd.device import get_context
get_context().synchronize(); dt = time.perf_counter() - t0
n = pipe.frame.samples
print("8192x8192x16spp = %.2e rays in %.3f s (%.2e rays/s incl. first-call setup); samples min/max %d/%d; mean finite %s" %
      (8192 * 8192 * 16, dt, 8192 * 8192 * 16 / dt, n.min(), n.max(), bool(np.isfinite(pipe.frame.mean).all())))
