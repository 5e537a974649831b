#!/usr/bin/env python3
"""Rays/s of the host-callback material path (source_amd/optical/hybrid.py: materials without a device lowering are evaluated in Python,
their rays traced on the GPU in waves) next to the device path, on the Cornell box (GPU box):
  device     every material lowered (k_render_trace_path)
  host       the library's own host forms of the same materials (HipEngine(host_materials=True)): array forms over whole waves
  user       the five walls re-implemented by a user subclass of Lambert (evaluate_shading in Python) — what a Raysect user's own
             material costs; Python evaluated by forked worker processes (default: min(cores, 16))
  user1      the same in one process
  pernode    every material called per node through evaluate_surface / evaluate_volume, one process (the plugin API's own cost)
and a cProfile of the `user1` run. The compiled reference's serial rate on this scene (tests/golden/reference_timing.json) is printed
beside them.   usage: python tools/host_material_rate.py [pixels] [spp]"""
import cProfile
import io
import json
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from source_amd import api as ns, scenes                    # noqa: E402
from source_amd.device import get_context                   # noqa: E402
from source_amd.optical.material import hemisphere_cosine_pdf   # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
SPP = int(sys.argv[2]) if len(sys.argv) > 2 else 2


class MyLambert(ns.Lambert):
    def evaluate_shading(self, world, ray, s_in, s_out, w_refl, w_trans, back_face, w2s, s2w, intersection):
        pdf = hemisphere_cosine_pdf(s_out)
        if pdf == 0.0:
            return ray.new_spectrum()
        spectrum = ray.spawn_daughter(w_refl, s_out.transform(s2w)).trace(world)
        spectrum.mul_array(self.reflectivity.sample(spectrum.min_wavelength, spectrum.max_wavelength, spectrum.bins))
        spectrum.mul_scalar(pdf)
        return spectrum


def run(kind, profile=False):
    world, prims = scenes.build_cornell(ns)
    if kind.startswith("user"):
        for p in prims:
            if isinstance(p.material, ns.Lambert):
                p.material = MyLambert(p.material.reflectivity)
    cam, pipe = scenes.cornell_camera(ns, world, (N, N), spp=SPP, bins=15)
    cam.frame_sampler = ns.RectFrameSampler2D()
    cam.render_engine = ns.HipEngine(rng="philox", seed=5, host_materials=(kind in ("host", "pernode")), per_node_materials=(kind == "pernode"),
                                     host_workers=1 if kind in ("user1", "pernode") else None)
    world.build_accelerator()
    cam.observe()                                           # warm-up (allocations, first launches)
    get_context().synchronize()
    prof = cProfile.Profile() if profile else None
    from source_amd.optical import hybrid
    del hybrid.last_stats[:]
    t0 = time.perf_counter()
    if prof:
        prof.enable()
    cam.observe()
    get_context().synchronize()
    if prof:
        prof.disable()
    dt = time.perf_counter() - t0
    rays = cam.stats["rays"]
    print("%-7s %4dx%-4d x %d spp: %8.3f s per pass, %.4g primary rays/s, %.4g rays/s (all rays: %d)" % (kind, N, N, SPP, dt, N * N * SPP / dt, rays / dt, rays), flush=True)
    from source_amd.optical import hybrid
    for st in hybrid.last_stats:
        print("   workers:", st)
    del hybrid.last_stats[:]
    if prof:
        out = io.StringIO()
        pstats.Stats(prof, stream=out).sort_stats("cumulative").print_stats(22)
        print(out.getvalue())
    return pipe.frame.mean.copy()


from source_amd.optical import hybrid                       # noqa: E402
print("cores usable: %d" % hybrid.usable_cores())
dev = run("device")
host = run("host")
user = run("user")
user1 = run("user1", profile=True)
pernode = run("pernode")
print("frames equal (device == host == user == user1 == pernode):", bool((dev == host).all() and (dev == user).all() and (dev == user1).all() and (dev == pernode).all()))
ref = os.path.join(ROOT, "tests", "golden", "reference_timing.json")
if os.path.exists(ref):
    table = json.load(open(ref))
    r = table.get("c1")
    u = table.get("c1user")
    if u:
        print("compiled reference with the same user-written Python material: serial %.4g primary rays/s" % u["reference_serial_rays_per_s"])
    if r:
        print("compiled reference on this scene (%s): serial %.4g primary rays/s, MulticoreEngine(8) %.4g" % (r.get("where", "development container"), r["reference_serial_rays_per_s"], r["reference_multicore_8_rays_per_s"]))
