#!/usr/bin/env python3
"""One-off stress run on the GPU box: device frames against the oracle (same Philox paths) on larger frames of the demo scenes than the
test-suite uses, to hit rare cases (exact CSG ties, grazing roots handed to the stream merge, very long paths). Prints the number of
differing frame entries per scene (must be 0; the tinted-glass scene is compared at 1e-12).   python tools/stress_parity.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from oracle import oracle as orc  # noqa: E402
from source_amd import api as ns, scenes  # noqa: E402


def run(name, world, cam, pipe, tol=0.0):
    w, h = cam.pixels
    bins = cam.spectral_bins
    cam.frame_sampler = ns.RectFrameSampler2D()
    cam.render_engine = ns.HipEngine(rng="philox", seed=2024)
    t0 = time.perf_counter(); cam.observe(); m = np.array(pipe.frame.mean); t_dev = time.perf_counter() - t0
    ref = np.zeros((w, h, bins)); rays = 0
    t0 = time.perf_counter()
    flat = world.flatten()
    for sl in cam._slice_spectrum():
        keep = []
        desc = cam.render_desc(world, None, sl, cam.render_engine, keep, rect=(0, 0, w, h))
        om, ov, n = orc.render_pinhole(flat, desc, threads=orc.max_threads())
        rays += n
        ref[:, :, sl.offset:sl.offset + sl.bins] = om.reshape(h, w, sl.bins).transpose(1, 0, 2)
    t_orc = time.perf_counter() - t0
    if tol == 0.0:
        bad = int((m != ref).sum())
    else:
        bad = int((~np.isclose(m, ref, rtol=tol, atol=0)).sum())
    print("%-10s %4dx%-4d rays %11d  device %.2f s  oracle %.1f s  ray counts %s  differing entries %d of %d" %
          (name, w, h, rays, t_dev, t_orc, "equal" if cam.stats["rays"] == rays else "DIFFER", bad, m.size), flush=True)
    return bad


K = int(sys.argv[1]) if len(sys.argv) > 1 else 1         # linear scale of the frames
bad = 0
world, _ = scenes.build_prism(ns)
cam, pipe = scenes.prism_camera(ns, world, (384 * K, 256 * K), 8, 8, 8)
bad += run("prism", world, cam, pipe)
world, _ = scenes.build_cornell(ns)
cam, pipe = scenes.cornell_camera(ns, world, (384 * K, 384 * K), 8, 6)
bad += run("cornell", world, cam, pipe)
world, _ = scenes.build_lambert(ns)
cam, pipe = scenes.lambert_camera(ns, world, (384 * K, 320 * K), 8, 5, (0.01, 3, 500))
cam.ray_importance_sampling = True
bad += run("lambert", world, cam, pipe)
world, _ = scenes.build_glass(ns)
cam, pipe = scenes.glass_camera(ns, world, (256 * K, 192 * K), 8, 6, 3, (0.01, 3, 500))
bad += run("glass", world, cam, pipe, tol=1e-12)
world, _ = scenes.build_csg_demo(ns)
cam, pipe = scenes.csg_camera(ns, world, (512 * K, 512 * K), spp=8, bins=5)
bad += run("csg", world, cam, pipe)
cam, pipe = scenes.csg_camera(ns, world, (384 * K, 384 * K), spp=16, bins=5)     # (16 spp and more: the packet walk is the fast pass)
bad += run("csg 16spp", world, cam, pipe)
# primary-ray mesh scenes through the render kernels (coherent-pass instantiation at 16 spp, 1-spp instantiation), whole frames
world = scenes.build_c3(ns, n=132)[0]
cam, pipe = scenes.c3_camera(ns, world, (256 * K, 256 * K), spp=16, bins=3)
bad += run("c3 16spp", world, cam, pipe)
world = scenes.build_c2(ns, n=132)[0]
cam, pipe = scenes.c2_camera(ns, world, (512 * K, 512 * K), spp=1, bins=3)
bad += run("c2 1spp", world, cam, pipe)
world = scenes.build_c2(ns, n=132, smoothing=True, with_normals=True)[0]
cam, pipe = scenes.c2_camera(ns, world, (256 * K, 256 * K), spp=5, bins=3)
bad += run("c2 smooth", world, cam, pipe)
sys.exit(1 if bad else 0)
