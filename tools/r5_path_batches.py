#!/usr/bin/env python3
"""Small path-traced passes, several per library call (HipEngine(passes_per_call=K) on a Cornell box, spectral pipeline, full-frame
sampler): wall time per pass for K = 1, 2, 4, 8 and the frames' digests (equal for every K).
usage: python tools/r5_path_batches.py [pixels] [spp] [passes] [K,K,...]"""
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from source_amd import api as ns, scenes  # noqa: E402
from source_amd.device import get_context  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 96
SPP = int(sys.argv[2]) if len(sys.argv) > 2 else 16
PASSES = int(sys.argv[3]) if len(sys.argv) > 3 else 32
KS = [int(k) for k in sys.argv[4].split(',')] if len(sys.argv) > 4 else [1, 2, 4, 8]
if len(KS) > 1:                                             # one process per K: a second world in a process meets the runtime's ~80 ms stalls (lab notebook)
    import subprocess
    for K in KS:
        subprocess.run([sys.executable, os.path.abspath(__file__), str(N), str(SPP), str(PASSES), str(K)], check=True)
    sys.exit(0)
for K in KS:
    world, _ = scenes.build_cornell(ns)
    cam, pipe = scenes.cornell_camera(ns, world, (N, N), spp=SPP, bins=15)
    cam.frame_sampler = ns.RectFrameSampler2D()
    cam.render_engine = ns.HipEngine(rng="philox", seed=5, passes_per_call=K, auto_batch=False)
    world.build_accelerator()
    cam.observe()
    get_context().synchronize()
    t0 = time.perf_counter()
    for _ in range(PASSES // K):
        cam.observe()
    get_context().synchronize()
    dt = time.perf_counter() - t0
    digest = hashlib.sha256(np.array(pipe.frame.mean).tobytes()).hexdigest()[:12]
    print("%dx%d x %d spp, %d passes, %d per call: %.2f ms per pass, %.3g paths/s, frame %s (after %d passes)" %
          (N, N, SPP, PASSES, K, 1e3 * dt / PASSES, N * N * SPP * PASSES / dt, digest, PASSES + K), flush=True)
