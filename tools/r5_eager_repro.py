"""round 5 (GPU box): eager partial batches (HipEngine.eager_batch) must leave the frames of full batches — stress over random groupings."""
import os, sys, hashlib, random, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from source_amd import api as ns, scenes
from source_amd.optical import observer as O
counts = []
orig = O.PinholeCamera._flush_lazy
def flush(self):
    p = getattr(self, "_lazy", None)
    if p is not None: counts.append(p["count"])
    return orig(self)
O.PinholeCamera._flush_lazy = flush
def run(eager, n, pixels=(40, 24), two=True, K=1, jitter=0.0):
    world = scenes.build_c2(ns, n=24)[0]
    pipes = [ns.SpectralRadiancePipeline2D(), ns.SpectralPowerPipeline2D()] if two else [ns.SpectralRadiancePipeline2D()]
    cam, _ = scenes.c2_camera(ns, world, pixels, spp=1, bins=6)
    cam.pipelines = pipes
    cam.frame_sampler = ns.RectFrameSampler2D()
    eng = ns.HipEngine(rng="philox", seed=11, passes_per_call=K)
    eng.eager_batch = eager
    cam.render_engine = eng
    counts.clear()
    for _ in range(n // K):
        cam.observe()
        if jitter: time.sleep(random.random() * jitter)
    h = hashlib.sha256()
    for p in pipes:
        for a in (p.frame.mean, p.frame.variance, p.frame.samples):
            h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:12], list(counts)
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    n = random.choice([100, 37, 64, 20])
    pixels = random.choice([(40, 24), (72, 50), (33, 21)])
    ref = run(False, n, pixels)[0]
    tog = run(False, n, pixels, K=n)[0]
    got, groups = run(True, n, pixels, jitter=random.choice([0.0, 0.0002, 0.001]))
    if not (ref == tog == got):
        bad += 1
        print("MISMATCH n=%d pixels=%s full batches %s, one call %s, eager %s groups %s" % (n, pixels, ref, tog, got, groups))
print("trials done, mismatches:", bad)
