#!/usr/bin/env python3
"""
Development-container only: times the COMPILED REFERENCE (built by build_reference.sh) and the CPU oracle on the same scene, so that
the oracle's rays/s measured on the GPU box (bench.py cpu_baseline, kind "port") can be related to the reference's own speed.

    bash tests/golden/build_reference.sh && python tests/golden/time_reference.py [pixels]

Scene: BASELINE configs[1] (69 432-triangle mesh, pinhole camera), `pixels` x `pixels`, 1 sample / pixel, 15 bins.
"""
import json
import os
import sys
import time
import types

os.environ.setdefault("MPLBACKEND", "Agg")
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("RAYSECT_REF_BUILD", "/tmp/rs_oracle")
sys.path.insert(0, REF)
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import raysect  # noqa: E402
from raysect.core import SerialEngine, MulticoreEngine, Point3D, Vector3D, translate, rotate  # noqa: E402
from raysect.primitive import Mesh, Sphere, Box, Cylinder, Union, Intersect, Subtract  # noqa: E402
from raysect.optical import World, ConstantSF, InterpolatedSF  # noqa: E402
from raysect.optical.material import AbsorbingSurface, UniformSurfaceEmitter  # noqa: E402
from raysect.optical.material.debug import Light  # noqa: E402
from raysect.optical.observer import PinholeCamera, FullFrameSampler2D, SpectralRadiancePipeline2D, SpectralPowerPipeline2D  # noqa: E402

from source_amd import scenes, api as ns  # noqa: E402
from oracle import oracle as orc  # noqa: E402

REFNS = types.SimpleNamespace(
    World=World, Mesh=Mesh, Sphere=Sphere, Box=Box, Cylinder=Cylinder, Union=Union, Intersect=Intersect, Subtract=Subtract,
    Point3D=Point3D, Vector3D=Vector3D, translate=translate, rotate=rotate, ConstantSF=ConstantSF, InterpolatedSF=InterpolatedSF,
    AbsorbingSurface=AbsorbingSurface, UniformSurfaceEmitter=UniformSurfaceEmitter, Light=Light, PinholeCamera=PinholeCamera,
    FullFrameSampler2D=FullFrameSampler2D, SpectralRadiancePipeline2D=SpectralRadiancePipeline2D,
    SpectralPowerPipeline2D=SpectralPowerPipeline2D)

px = int(sys.argv[1]) if len(sys.argv) > 1 else 256
out = {"pixels": px, "host_cpus": os.cpu_count()}

world, mesh, box = scenes.build_c2(REFNS, n=132)
cam, pipe = scenes.c2_camera(REFNS, world, (px, px), spp=1, bins=15)
world.build_accelerator()
for name, engine in (("reference_serial", SerialEngine()), ("reference_multicore_8", MulticoreEngine(processes=8))):
    cam.render_engine = engine
    t0 = time.perf_counter()
    cam.observe()
    out[name + "_rays_per_s"] = round(px * px / (time.perf_counter() - t0), 1)

world, mesh, box = scenes.build_c2(ns, n=132)
cam, pipe = scenes.c2_camera(ns, world, (px, px), spp=1, bins=15)
flat = world.flatten()
keep = []
desc = cam.render_desc(world, None, cam._slice_spectrum()[0], ns.HipEngine(rng="philox", seed=1), keep, rect=(0, 0, px, px))
for threads in (1, 8):
    orc.render_pinhole(flat, desc, threads=threads)
    t0 = time.perf_counter()
    m, v, rays = orc.render_pinhole(flat, desc, threads=threads)
    out["oracle_%d_threads_rays_per_s" % threads] = round(rays / (time.perf_counter() - t0), 1)
out["oracle_1_over_reference_serial"] = round(out["oracle_1_threads_rays_per_s"] / out["reference_serial_rays_per_s"], 2)
print(json.dumps(out))
