import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from source_amd import api as ns, scenes
from source_amd.device import get_context
world = scenes.build_prism(ns)[0]
cam, pipe = scenes.prism_camera(ns, world, (1024, 1024), 1, 512, 512)
cam.frame_sampler = ns.RectFrameSampler2D()
cam.render_engine = ns.HipEngine(rng="philox", seed=1)
ctx = get_context()
world.build_accelerator()
cam.observe(); ctx.synchronize()
t0 = time.perf_counter(); cam.observe(); ctx.synchronize(); dt = time.perf_counter() - t0
tr, ac = ctx.render_history(512)
print("512 slices x 1024^2 x 1 spp: %.2f s per pass; kernels: trace %.1f ms + accumulate %.2f ms per slice (sum %.2f s)" % (dt, sum(tr) / len(tr), sum(ac) / len(ac), (sum(tr) + sum(ac)) / 1e3))
