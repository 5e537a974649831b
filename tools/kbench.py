#!/usr/bin/env python3
"""Kernel iteration harness (GPU box): renders BASELINE configs[1] K times with the library named by $RSX_LIB and prints the
per-launch kernel times plus a SHA-256 of the resulting frame (any variant must reproduce the same digest)."""
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from source_amd import api as ns, scenes  # noqa: E402
from source_amd.device import get_context  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
config = sys.argv[2] if len(sys.argv) > 2 else "c2"
if config == "c2":
    world = scenes.build_c2(ns, n=132)[0]
    cam, pipe = scenes.c2_camera(ns, world, (1024, 1024), spp=1, bins=15)
elif config == "c3":
    world = scenes.build_c3(ns, n=132)[0]
    cam, pipe = scenes.c3_camera(ns, world, (1024, 1024), spp=int(os.environ.get("KB_SPP", "4")), bins=15)
elif config == "c3full":
    world = scenes.build_c3(ns, n=132)[0]
    cam, pipe = scenes.c3_camera(ns, world, (2048, 2048), spp=64, bins=15)
elif config == "c3s100":                  # the reference's default pixel_samples
    world = scenes.build_c3(ns, n=132)[0]
    cam, pipe = scenes.c3_camera(ns, world, (1024, 1024), spp=100, bins=15)
elif config == "c3s24":
    world = scenes.build_c3(ns, n=132)[0]
    cam, pipe = scenes.c3_camera(ns, world, (1024, 1024), spp=24, bins=15)
elif config == "lambert":                 # diffuse inter-reflection room (fixture F13's scene), observer-default roulette
    world = scenes.build_lambert(ns)[0]
    cam, pipe = scenes.lambert_camera(ns, world, (1024, 1024), 16, 15, (0.01, 3, 500))
elif config == "lambert_csg":             # ... with the CSG solid, without volumes
    world = scenes.build_lambert(ns, with_volume=False, csg=True)[0]
    cam, pipe = scenes.lambert_camera(ns, world, (1024, 1024), 16, 15, (0.01, 3, 500))
elif config == "lambert_vol":             # ... with the volumes, without the CSG solid
    world = scenes.build_lambert(ns, with_volume=True, csg=False)[0]
    cam, pipe = scenes.lambert_camera(ns, world, (1024, 1024), 16, 15, (0.01, 3, 500))
elif config == "lambert_plain":           # the same room without the CSG solid and without volumes
    world = scenes.build_lambert(ns, with_volume=False, csg=False)[0]
    cam, pipe = scenes.lambert_camera(ns, world, (1024, 1024), 16, 15, (0.01, 3, 500))
elif config == "cornell":                 # BASELINE configs[0]'s scene (demos/cornell_box.py variant): Lambert + glass, importance sampling
    world = scenes.build_cornell(ns)[0]
    cam, pipe = scenes.cornell_camera(ns, world, (1024, 1024), 16, 15)
elif config == "prism":                   # BASELINE configs[4]'s scene (demos/prism.py variant), 32 one-bin spectral slices per pass
    world = scenes.build_prism(ns)[0]
    cam, pipe = scenes.prism_camera(ns, world, (1024, 1024), int(os.environ.get("KB_SPP", "4")), 32, 32)
elif config == "glass":                   # refraction scene (fixture F14's), 3 spectral slices
    world = scenes.build_glass(ns)[0]
    cam, pipe = scenes.glass_camera(ns, world, (1024, 1024), 16, 15, 3, (0.01, 3, 500))
elif config == "flat":
    world = scenes.build_flat(ns, n=512)[0]
    cam, pipe = scenes.c2_camera(ns, world, (2048, 2048), spp=64, bins=15)
elif config == "c4full":
    world = scenes.build_csg_demo(ns)[0]
    cam, pipe = scenes.csg_camera(ns, world, (1024, 1024), spp=16, bins=15)
elif config == "csg":
    world = scenes.build_csg_demo(ns)[0]
    cam, pipe = scenes.csg_camera(ns, world, (1024, 1024), spp=1, bins=15)
cam.frame_sampler = ns.RectFrameSampler2D()
eng = ns.HipEngine(rng="philox", seed=20250905, timing=False)
cam.render_engine = eng
ctx = get_context()
world.build_accelerator()
import time
warm = int(os.environ.get("KB_WARM", "3"))
for k in range(warm):
    eng.sample_offset = k * cam.pixel_samples
    cam.observe()
ctx.synchronize()
t0 = time.perf_counter()
for k in range(warm, warm + steps):
    eng.sample_offset = k * cam.pixel_samples
    cam.observe()
ctx.synchronize()
wall = (time.perf_counter() - t0) / steps
rays = cam.pixels[0] * cam.pixels[1] * cam.pixel_samples
tr, ac = ctx.render_history(min(steps, 512))
mean = pipe.frame.mean
digest = hashlib.sha256(np.ascontiguousarray(mean).tobytes()).hexdigest()[:16]
print(json.dumps({"lib": os.path.basename(os.environ.get("RSX_LIB", "librsx.so")), "config": config,
                  "trace_ms": round(float(np.mean(tr)), 4), "trace_min": round(float(np.min(tr)), 4), "trace_max": round(float(np.max(tr)), 4),
                  "accum_ms": round(float(np.mean(ac)), 4), "wall_ms": round(wall * 1e3, 4), "rays_per_s": round(rays / wall, 1),
                  "all_rays_per_pass": int(cam.stats.get("rays", 0)), "all_rays_per_s": round(cam.stats.get("rays", 0) / wall, 1), "digest": digest}))
