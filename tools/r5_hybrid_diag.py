import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from source_amd import api as ns, scenes
from source_amd.device import get_context
from source_amd.optical import hybrid
def rss():
    for line in open("/proc/self/status"):
        if line.startswith("VmRSS"): return line.split()[1]
world, prims = scenes.build_cornell(ns)
cam, pipe = scenes.cornell_camera(ns, world, (256, 256), spp=4, bins=15)
cam.frame_sampler = ns.RectFrameSampler2D()
cam.render_engine = ns.HipEngine(rng="philox", seed=5, host_materials=True)
scene = world.build_accelerator()
rng = np.random.default_rng(0)
def bench(tag, n=4096, reps=50):
    o = np.tile(np.array([[0.0, 0.0, -3.0]]), (n, 1)); d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1)[:, None]; m = np.full(n, np.inf)
    scene.hit_batch(o, d, m, geometry=True)
    t0 = time.perf_counter()
    for _ in range(reps): hybrid.trace_wave(scene, None, o, d, m)
    dt = (time.perf_counter() - t0) / reps
    print("%-40s n=%d: %.3f ms per trace_wave, rss %s kB" % (tag, n, dt * 1e3, rss()), flush=True)
bench("fresh")
bench("fresh", 40000, 10)
pids = []
for _ in range(16):
    pid = os.fork()
    if pid == 0:
        time.sleep(3); os._exit(0)
    pids.append(pid)
bench("16 idle children alive")
for p in pids: os.waitpid(p, 0)
bench("children gone")
cam.observe()
bench("after a host pass")
bench("after a host pass", 40000, 10)
pids = []
for _ in range(16):
    pid = os.fork()
    if pid == 0:
        time.sleep(3); os._exit(0)
    pids.append(pid)
bench("host pass + 16 idle children alive")
for p in pids: os.waitpid(p, 0)
bench("host pass, children gone")
import gc; gc.collect()
bench("after gc.collect")
