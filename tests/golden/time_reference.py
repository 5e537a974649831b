#!/usr/bin/env python3
"""
Development-container only: times the COMPILED REFERENCE (built by build_reference.sh) and the CPU oracle on the same scene, so that
the oracle's rays/s measured on the GPU box (bench.py cpu_baseline, kind "port") can be related to the reference's own speed.

    bash tests/golden/build_reference.sh && python tests/golden/time_reference.py [pixels] [c2|c3] [spp]

Scene: BASELINE configs[1] (c2: 69 432-triangle mesh), configs[2] (c3: 15 instances of it + floor box) or configs[0]'s Cornell box
(c1: path traced; the rates are primary rays per second), pinhole camera,
`pixels` x `pixels`, `spp` samples / pixel, 15 bins. The result is merged into tests/golden/reference_timing.json (committed:
bench.py reports it as cpu_baseline.reference next to the oracle's rate on the GPU box).
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
from raysect.optical.material import AbsorbingSurface, UniformSurfaceEmitter, Lambert, Dielectric, Sellmeier  # noqa: E402
from raysect.optical.material.debug import Light  # noqa: E402
from raysect.optical.observer import PinholeCamera, FullFrameSampler2D, SpectralRadiancePipeline2D, SpectralPowerPipeline2D  # noqa: E402

from source_amd import scenes, api as ns  # noqa: E402
from oracle import oracle as orc  # noqa: E402

REFNS = types.SimpleNamespace(
    World=World, Mesh=Mesh, Sphere=Sphere, Box=Box, Cylinder=Cylinder, Union=Union, Intersect=Intersect, Subtract=Subtract,
    Point3D=Point3D, Vector3D=Vector3D, translate=translate, rotate=rotate, ConstantSF=ConstantSF, InterpolatedSF=InterpolatedSF,
    AbsorbingSurface=AbsorbingSurface, UniformSurfaceEmitter=UniformSurfaceEmitter, Light=Light, PinholeCamera=PinholeCamera,
    Lambert=Lambert, Dielectric=Dielectric, Sellmeier=Sellmeier,
    FullFrameSampler2D=FullFrameSampler2D, SpectralRadiancePipeline2D=SpectralRadiancePipeline2D,
    SpectralPowerPipeline2D=SpectralPowerPipeline2D)

px = int(sys.argv[1]) if len(sys.argv) > 1 else 256
wl = sys.argv[2] if len(sys.argv) > 2 else "c2"
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 1
out = {"workload": wl, "pixels": px, "spp": spp, "host_cpus": os.cpu_count(), "where": "development container (8 vCPU Xeon 2.1 GHz), compiled Cython reference"}


class UserLambert(Lambert):
    """c1user: what a Raysect user's own material costs IN THE REFERENCE — the Cornell box's walls as a Python subclass of the compiled
    Lambert with evaluate_shading written in Python (the counterpart of tools/host_material_rate.py's `user` case on the device)."""

    def __init__(self, reflectivity=None):
        super().__init__(reflectivity)
        self.refl = reflectivity                            # (the compiled class keeps its own copy private)

    def evaluate_shading(self, world, ray, s_incoming, s_outgoing, w_reflection_origin, w_transmission_origin, back_face,
                         world_to_surface, surface_to_world, intersection):
        pdf = s_outgoing.z * 0.3183098861837907 if s_outgoing.z > 0 else 0.0
        if pdf == 0.0:
            return ray.new_spectrum()
        spectrum = ray.spawn_daughter(w_reflection_origin, s_outgoing.transform(surface_to_world)).trace(world)
        spectrum.samples[:] *= self.refl.sample(spectrum.min_wavelength, spectrum.max_wavelength, spectrum.bins)   # (mul_array / mul_scalar are cdef:
        spectrum.samples[:] *= pdf                                                                                #  a Python material uses numpy)
        return spectrum


def build(api):
    if wl == "c1user":
        if api is REFNS:
            api = types.SimpleNamespace(**dict(vars(REFNS), Lambert=UserLambert))
        world, prims = scenes.build_cornell(api)
        return world, scenes.cornell_camera(api, world, (px, px), spp=spp, bins=15)
    if wl == "c1":                                          # the Cornell box of configs[0]: path traced (rays/s below = PRIMARY rays/s)
        world = scenes.build_cornell(api)[0]
        return world, scenes.cornell_camera(api, world, (px, px), spp=spp, bins=15)
    if wl == "c3":
        world = scenes.build_c3(api, n=132)[0]
        return world, scenes.c3_camera(api, world, (px, px), spp=spp, bins=15)
    world = scenes.build_c2(api, n=132)[0]
    return world, scenes.c2_camera(api, world, (px, px), spp=spp, bins=15)


world, (cam, pipe) = build(REFNS)
world.build_accelerator()
for name, engine in (("reference_serial", SerialEngine()), ("reference_multicore_8", MulticoreEngine(processes=8))):
    cam.render_engine = engine
    t0 = time.perf_counter()
    cam.observe()
    out[name + "_rays_per_s"] = round(px * px * spp / (time.perf_counter() - t0), 1)

if wl == "c1user":                                          # (no oracle leg: the oracle has no user materials)
    print(json.dumps(out))
    path = os.path.join(HERE, "reference_timing.json")
    table = json.load(open(path)) if os.path.exists(path) else {}
    table[wl] = out
    json.dump(table, open(path, "w"), indent=1, sort_keys=True)
    sys.exit(0)
world, (cam, pipe) = build(ns)
flat = world.flatten()
keep = []
desc = cam.render_desc(world, None, cam._slice_spectrum()[0], ns.HipEngine(rng="philox", seed=1), keep, rect=(0, 0, px, px))
for threads in (1, 8):
    orc.render_pinhole(flat, desc, threads=threads)
    t0 = time.perf_counter()
    m, v, rays = orc.render_pinhole(flat, desc, threads=threads)
    out["oracle_%d_threads_rays_per_s" % threads] = round(px * px * spp / (time.perf_counter() - t0), 1)
out["oracle_1_over_reference_serial"] = round(out["oracle_1_threads_rays_per_s"] / out["reference_serial_rays_per_s"], 2)
out["oracle_8_over_reference_multicore_8"] = round(out["oracle_8_threads_rays_per_s"] / out["reference_multicore_8_rays_per_s"], 2)
print(json.dumps(out))
path = os.path.join(HERE, "reference_timing.json")
table = json.load(open(path)) if os.path.exists(path) else {}
table[wl] = out
json.dump(table, open(path, "w"), indent=1, sort_keys=True)
