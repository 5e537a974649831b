#!/usr/bin/env python3
"""One-off stress run on the GPU box for the world level of the traversal kernels (wide primitives, wide-only leaf tags, the cull by
the nearest wide answer): random worlds of 3 .. 28 spheres, boxes and cylinders — grid-snapped so that faces coincide and exact ties
between primitives are common, some huge (floors, shells: they sit in most world leaves), optionally a small mesh — hit by random,
axis-parallel and grid-aligned rays (rsx_hit_batch: two wide slots) and path traced with scattering / refracting / emitting materials
(k_render_trace_path: eight wide slots, cull bits) against the oracle: ids, distances, geometry and whole frames bit for bit.
python tools/stress_world.py [worlds] [rays_per_world]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from oracle import oracle as orc  # noqa: E402
from source_amd import api as ns, scenes  # noqa: E402

n_worlds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n_rays = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
rng = np.random.RandomState(2468)
GRID = [-1.0, -0.5, -0.25, 0.0, 0.25, 0.5, 1.0]
P = ns.Point3D


def snap(scale=1.0):
    return scale * (float(rng.choice(GRID)) if rng.rand() < 0.6 else float(rng.uniform(-1.0, 1.0)))


def primitive(world, huge):
    kind = rng.randint(3)
    t = ns.translate(snap(1.5), snap(1.5), snap(1.5))
    if rng.rand() < 0.35:
        t = t * ns.rotate(float(rng.choice([0, 90, 30, 45])), float(rng.choice([0, 90, 30])), float(rng.choice([0, 90])))
    s = float(rng.choice([2.0, 3.0, 4.0])) if huge else 1.0
    if kind == 0:
        return ns.Sphere(s * float(rng.choice([0.25, 0.5, 0.4])), world, t)
    if kind == 1:
        lo = [s * float(rng.choice([-0.5, -0.25])) for _ in range(3)]
        hi = [s * float(rng.choice([0.25, 0.5])) for _ in range(3)]
        if huge and rng.rand() < 0.5:                       # a slab: floor / wall
            ax = rng.randint(3)
            lo[ax], hi[ax] = -0.05, 0.0
        return ns.Box(P(*lo), P(*hi), world, t)
    return ns.Cylinder(s * float(rng.choice([0.25, 0.5])), s * float(rng.choice([0.5, 1.0])), world, t)


total = bad = hits = frames = 0
for wi in range(n_worlds):
    world = ns.World()
    n_prims = int(rng.choice([3, 5, 8, 9, 12, 20, 28]))
    for k in range(n_prims):
        primitive(world, huge=rng.rand() < 0.25)
    if rng.rand() < 0.3:                                    # a mesh among them: subtrees that hold it cannot be culled
        v, t = scenes.displaced_sphere(4, radius=0.4)
        ns.Mesh(v, t, parent=world, transform=ns.translate(snap(), snap(), snap()))
    scene = world.build_accelerator()
    n = n_rays
    o = rng.uniform(-3, 3, size=(n, 3))
    d = rng.normal(size=(n, 3))
    k8 = n // 8
    o[:k8] = rng.choice(GRID + [2.0, -2.0, 3.0], size=(k8, 3))                 # grid-aligned origins
    axis = rng.randint(3, size=k8)
    d[k8:2 * k8] = 0.0
    d[np.arange(k8, 2 * k8), axis] = rng.choice([-1.0, 1.0], size=k8)          # axis-parallel rays (zero components)
    d /= np.linalg.norm(d, axis=1)[:, None]
    m = np.where(rng.rand(n) < 0.2, rng.uniform(0.1, 4.0, size=n), np.inf)
    dev = scene.hit_batch(o, d, m, geometry=True)
    ref = orc.hit_batch(world.flatten(), o, d, m, geometry=True)
    hit = ref["prim"] >= 0
    same = (dev["prim"] == ref["prim"]) & ((dev["t"] == ref["t"]) | ~hit) & ((dev["exiting"] == ref["exiting"]) | ~hit)
    same &= (np.all((dev["geom"] == ref["geom"]) | np.isnan(ref["geom"]), axis=1) | ~hit)
    total += n
    hits += int(hit.sum())
    bad += int((~same).sum())
    if not same.all():
        i = int(np.argmin(same))
        print("world %d (%d primitives): %d mismatches, first at ray %d: device prim %d t %r, oracle prim %d t %r" %
              (wi, n_prims, int((~same).sum()), i, dev["prim"][i], dev["t"][i], ref["prim"][i], ref["t"][i]), flush=True)
    if wi % int(os.environ.get("STRESS_FRAME_EVERY", "2")) == 0:
        mats = [ns.Lambert(ns.ConstantSF(0.8)), ns.Dielectric(ns.ConstantSF(1.5), ns.ConstantSF(1.0)), ns.UniformVolumeEmitter(ns.ConstantSF(1.0), 0.5),
                ns.Lambert(ns.ConstantSF(0.5)), ns.NullMaterial(), ns.UniformSurfaceEmitter(ns.ConstantSF(1.0), 0.7), ns.Lambert(ns.ConstantSF(0.3))]
        for prim in list(world._primitives):
            prim.material = mats[rng.randint(len(mats))]
            if rng.rand() < 0.2:
                prim.material.importance = float(rng.choice([1.0, 4.0]))
        ns.Box(P(-6, -6, -6), P(6, 6, 6), world, material=ns.UniformSurfaceEmitter(ns.ConstantSF(1.0), 1.0))
        pipe = ns.SpectralRadiancePipeline2D()
        cam = ns.PinholeCamera((96, 96), fov=60, parent=world, pipelines=[pipe], frame_sampler=ns.RectFrameSampler2D(), transform=ns.translate(0.2, 0.1, -4.5))
        cam.pixel_samples, cam.spectral_bins, cam.spectral_rays, cam.quiet = 4, 3, 1, True
        cam.ray_extinction_prob, cam.ray_extinction_min_depth, cam.ray_max_depth = 0.05, 2, 60
        cam.ray_importance_sampling = bool(rng.rand() < 0.5)
        cam.render_engine = ns.HipEngine(rng="philox", seed=wi)
        cam.observe()
        keep = []
        desc = cam.render_desc(world, None, cam._slice_spectrum()[0], cam.render_engine, keep, rect=(0, 0, 96, 96))
        om, ov, rays = orc.render_pinhole(world.flatten(), desc, threads=orc.max_threads())
        fm = np.array(pipe.frame.mean)
        rm = om.reshape(96, 96, 3).transpose(1, 0, 2)
        diff = int(((fm != rm) & ~(np.isnan(fm) & np.isnan(rm))).sum())
        frames += 1
        if diff or cam.stats["rays"] != rays:
            print("world %d (%d primitives): path-traced frame differs in %d entries (ray counts %d / %d)" % (wi, n_prims, diff, cam.stats["rays"], rays), flush=True)
            bad += max(diff, 1)
print("%d worlds, %d rays, %d hits, %d path-traced frames, mismatches: %d" % (n_worlds, total, hits, frames, bad))
