import os, sys, time, cProfile, pstats
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from source_amd import api as ns, scenes
from source_amd.device import get_context
world = scenes.build_c2(ns, n=132)[0]
cam, pipe = scenes.c2_camera(ns, world, (1024, 1024), spp=1, bins=15)
cam.frame_sampler = ns.RectFrameSampler2D()
cam.render_engine = ns.HipEngine(rng="philox", seed=1)
ctx = get_context()
world.build_accelerator()
for k in range(50):
    cam.observe()
ctx.synchronize()
t0 = time.perf_counter()
for k in range(500):
    cam.observe()
t1 = time.perf_counter()
ctx.synchronize()
t2 = time.perf_counter()
print("host issue %.3f ms per observe, with drain %.3f ms" % ((t1 - t0) / 500 * 1e3, (t2 - t0) / 500 * 1e3))
pr = cProfile.Profile()
pr.enable()
for k in range(300):
    cam.observe()
pr.disable()
ctx.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
# the same call with an idle device: what the host side alone costs
import numpy as np
ts = []
for rep in range(50):
    ctx.synchronize()
    t0 = time.perf_counter()
    cam.observe()
    ts.append(time.perf_counter() - t0)
print("observe() on an idle device: host %.3f ms median, %.3f min" % (np.median(ts) * 1e3, min(ts) * 1e3))
for depth in (1, 2, 3, 4, 6):
    os.environ["RSX_PIPELINE"] = str(depth)
