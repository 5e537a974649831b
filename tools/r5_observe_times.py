"""round 5 (GPU box): wall time of every observe() of a configs[1] loop (which call waits for the device?)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from source_amd import api as ns, scenes
from source_amd.device import get_context
world = scenes.build_c2(ns, n=132)[0]
cam, pipe = scenes.c2_camera(ns, world, (1024, 1024), spp=1, bins=15)
cam.frame_sampler = ns.RectFrameSampler2D()
eng = ns.HipEngine(rng="philox", seed=1)
eng.eager_batch = os.environ.get("RSX_EAGER_BATCH", "1") != "0"
cam.render_engine = eng
ctx = get_context()
world.build_accelerator()
for _ in range(20): cam.observe()
ctx.synchronize()
ts = []
t0 = time.perf_counter()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for k in range(N):
    a = time.perf_counter(); cam.observe(); ts.append((time.perf_counter() - a) * 1e3)
tl = time.perf_counter(); ctx.synchronize(); te = time.perf_counter()
print("eager", eng.eager_batch, eng.eager_min, "N", N, "rays/s %.3g" % (N * 1048576 / (te - t0)), "loop %.3f ms, final sync %.3f ms, total %.3f ms" % ((tl - t0) * 1e3, (te - tl) * 1e3, (te - t0) * 1e3))
print(" ".join("%.2f" % t for t in ts)); sys.exit(0)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for k in range(64): cam.observe()
pr.disable(); ctx.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
