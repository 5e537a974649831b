#!/usr/bin/env python3
"""README's path-tracing loop (tools/readme_example_paths.py: Cornell box, RGB pipeline, adaptive sampler, 16 spp per pass) timed pass by
pass with a cProfile of the loop: what a small adaptive pass costs on the host next to its kernels.
usage: python tools/r5_paths_profile.py [pixels] [passes]"""
import cProfile
import io
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from source_amd import api as rs, scenes  # noqa: E402
from source_amd.device import get_context  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 96
PASSES = int(sys.argv[2]) if len(sys.argv) > 2 else 40


def loop(profile):
    world, _ = scenes.build_cornell(rs)
    rgb = rs.RGBPipeline2D()
    cam, _ = scenes.cornell_camera(rs, world, (N, N), spp=16, bins=15, pipelines=[rgb])
    cam.frame_sampler = rs.RGBAdaptiveSampler2D(rgb, ratio=10, fraction=0.2, min_samples=64, cutoff=0.05)
    cam.render_engine = rs.MulticoreEngine()
    cam.observe()
    get_context().synchronize()
    prof = cProfile.Profile() if profile else None
    t0 = time.perf_counter()
    if prof:
        prof.enable()
    passes = 0
    while not cam.render_complete and passes < PASSES:
        cam.observe()
        passes += 1
    get_context().synchronize()
    if prof:
        prof.disable()
    dt = time.perf_counter() - t0
    print("%dx%d, %d passes of 16 spp: %.2f ms per pass (%s)" % (N, N, passes, 1e3 * dt / max(1, passes), "profiled" if profile else "plain"), flush=True)
    if prof:
        out = io.StringIO()
        pstats.Stats(prof, stream=out).sort_stats("tottime").print_stats(28)
        print(out.getvalue())


loop(False)
loop(False)
loop(True)
