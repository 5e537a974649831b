#!/usr/bin/env python3
"""One-off stress run on the GPU box for the state-free CSG evaluator: random operand trees of spheres, boxes and cylinders — many
with deliberately coplanar faces and shared centres, so that exact ties between operand roots are common — hit by random, axis-parallel
and grid-aligned rays; rsx_hit_batch (fast pass + stream-merge redo pass) against the oracle's stream merge: primitive id, distance,
exiting flag and the full intersection geometry must be identical.   python tools/stress_csg.py [worlds] [rays_per_world] [depth]
depth (default 3: trees of at most 8 leaves, the state-free evaluator's domain) up to 12 and beyond: deeper, bigger trees go through the
stream merge alone (one loop over an explicit frame stack, node states in the scene's arena beyond 16 nodes)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from oracle import oracle as orc  # noqa: E402
from source_amd import api as ns  # noqa: E402

n_worlds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n_rays = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
DEPTH = int(sys.argv[3]) if len(sys.argv) > 3 else 3
MAX_LEAVES = 8 if DEPTH <= 3 else 80
rng = np.random.RandomState(12345)
GRID = [-0.5, -0.25, 0.0, 0.25, 0.5]                         # coordinates snap to a grid: coincident faces and centres


def snap():
    return float(rng.choice(GRID)) if rng.rand() < 0.6 else float(rng.uniform(-0.5, 0.5))


def leaf():
    kind = rng.randint(3)
    t = ns.translate(snap(), snap(), snap())
    if rng.rand() < 0.4:
        t = t * ns.rotate(float(rng.choice([0, 90, 30, 45])), float(rng.choice([0, 90, 30])), float(rng.choice([0, 90])))
    if kind == 0:
        return ns.Sphere(float(rng.choice([0.25, 0.5, 0.4])), transform=t)
    if kind == 1:
        lo = [float(rng.choice([-0.5, -0.25])) for _ in range(3)]
        hi = [float(rng.choice([0.25, 0.5])) for _ in range(3)]
        return ns.Box(ns.Point3D(*lo), ns.Point3D(*hi), transform=t)
    return ns.Cylinder(float(rng.choice([0.25, 0.5])), float(rng.choice([0.5, 1.0])), transform=t)


def tree(depth):
    if depth == 0 or rng.rand() < (0.3 if DEPTH <= 3 else 0.12):
        return leaf()
    op = [ns.Union, ns.Intersect, ns.Subtract][rng.randint(3)]
    t = ns.translate(snap(), snap(), snap()) if rng.rand() < 0.5 else None
    return op(tree(depth - 1), tree(depth - 1), transform=t) if t is not None else op(tree(depth - 1), tree(depth - 1))


def count_leaves(p):
    return count_leaves(p.primitive_a) + count_leaves(p.primitive_b) if hasattr(p, "primitive_a") else 1


total = bad = hits = inside = frames = 0
for wi in range(n_worlds):
    world = ns.World()
    made = 0
    while made < 4:
        obj = tree(DEPTH)
        if not hasattr(obj, "primitive_a") or count_leaves(obj) > MAX_LEAVES:
            continue
        obj.parent = world
        obj.transform = ns.translate(float(rng.choice([-1.5, 0, 1.5])), float(rng.choice([-1.5, 0, 1.5])), 0.0) * (obj.transform or ns.translate(0, 0, 0))
        obj.material = ns.AbsorbingSurface()
        made += 1
    n = n_rays
    o = rng.uniform(-3, 3, (n, 3))
    d = rng.normal(size=(n, 3))
    k = n // 4                                                # a quarter axis-parallel from grid points: ties galore
    axis = rng.randint(3, size=k)
    o[:k] = rng.choice(GRID + [1.0, 1.5, 1.75, -1.5, -1.25], size=(k, 3))
    d[:k] = 0.0
    d[np.arange(k), axis] = rng.choice([-1.0, 1.0], size=k)
    o[np.arange(k), axis] = -4.0 * d[np.arange(k), axis]
    o[k:2 * k] = rng.choice(GRID + [1.5, -1.5], size=(k, 3)) + rng.choice([0.0, 3.0, -3.0], size=(k, 3))   # through grid points
    d[k:2 * k] = rng.choice(GRID + [1.5, -1.5], size=(k, 3)) - o[k:2 * k] + 1e-300
    norm = np.linalg.norm(d, axis=1)
    d[norm < 1e-200] = [0.0, 0.0, 1.0]
    d /= np.linalg.norm(d, axis=1)[:, None]
    m = np.where(rng.rand(n) < 0.2, rng.uniform(0.5, 6.0, n), np.inf)
    scene = world.build_accelerator()
    dev = scene.hit_batch(o, d, m, geometry=True)
    ref = orc.hit_batch(world.flatten(), o, d, m, geometry=True, threads=orc.max_threads())
    same = (dev["prim"] == ref["prim"])
    hit = ref["prim"] >= 0
    same &= np.where(hit, dev["t"] == ref["t"], True) & np.where(hit, dev["exiting"] == ref["exiting"], True)
    geq = (dev["geom"] == ref["geom"]) | (np.isnan(dev["geom"]) & np.isnan(ref["geom"]))     # degenerate rays give NaN points on both sides
    same &= np.where(hit[:, None], geq, True).all(axis=1)
    # contains(): the flattened operand program against the oracle's recursion, random and grid points (points on faces included)
    pts = rng.uniform(-2.2, 2.2, (n // 2, 3))
    pts[: n // 8] = rng.choice(GRID + [1.0, 1.5, 1.75, 2.0, -1.0, -1.5, -1.25, -2.0], size=(n // 8, 3))
    cd, cr = scene.contains_batch(pts), orc.contains_batch(world.flatten(), pts)
    if not np.array_equal(cd, cr):
        print("world %d: contains() differs for %d points" % (wi, int((cd != cr).any(axis=1).sum())), flush=True)
        bad += int((cd != cr).any(axis=1).sum())
    inside += int(cr.sum())
    # the same solids path traced: random scattering / refracting / emitting materials, an emitting shell around them — the path
    # kernel's fast pass (state-free hit and contains), its redo pass and the term replay against the oracle, bit for bit
    if wi % int(os.environ.get("STRESS_FRAME_EVERY", "4")) == 0:
        mats = [ns.Lambert(ns.ConstantSF(0.8)), ns.Dielectric(ns.ConstantSF(1.5), ns.ConstantSF(1.0)), ns.UniformVolumeEmitter(ns.ConstantSF(1.0), 0.5),
                ns.Lambert(ns.ConstantSF(0.5)), ns.NullMaterial(), ns.Dielectric(ns.ConstantSF(1.3), ns.ConstantSF(1.0), transmission_only=True)]
        for prim in list(world._primitives):
            prim.material = mats[rng.randint(len(mats))]
        ns.Box(ns.Point3D(-4, -4, -4), ns.Point3D(4, 4, 4), world, material=ns.UniformSurfaceEmitter(ns.ConstantSF(1.0), 1.0))
        pipe = ns.SpectralRadiancePipeline2D()
        cam = ns.PinholeCamera((96, 96), fov=60, parent=world, pipelines=[pipe], frame_sampler=ns.RectFrameSampler2D(), transform=ns.translate(0.2, 0.1, -3.5))
        cam.pixel_samples, cam.spectral_bins, cam.spectral_rays, cam.quiet = 4, 3, 1, True
        cam.ray_extinction_prob, cam.ray_extinction_min_depth, cam.ray_max_depth = 0.05, 2, 60
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
            print("world %d: path-traced frame differs in %d entries (ray counts %d / %d)" % (wi, diff, cam.stats["rays"], rays), flush=True)
            bad += max(diff, 1)
    total += n
    hits += int(hit.sum())
    bad += int((~same).sum())
    if not same.all():
        i = int(np.argmin(same))
        print("world %d: %d mismatches, first at ray %d: device prim %d t %r, oracle prim %d t %r" %
              (wi, int((~same).sum()), i, dev["prim"][i], dev["t"][i], ref["prim"][i], ref["t"][i]), flush=True)
        print("   exiting", dev["exiting"][i], ref["exiting"][i], "\n   dev geom", dev["geom"][i].tolist(), "\n   ref geom", ref["geom"][i].tolist(),
              "\n   ray", o[i].tolist(), d[i].tolist(), m[i])

        def show(p, ind=0):
            tr = p.transform
            print("   " + " " * ind + type(p).__name__, [round(v, 4) for v in (tr.m if tr is not None else [])][3:12:4],
                  getattr(p, "radius", ""), getattr(p, "height", ""), getattr(p, "lower", ""), getattr(p, "upper", ""))
            if hasattr(p, "primitive_a"):
                show(p.primitive_a, ind + 2); show(p.primitive_b, ind + 2)
        show(world._primitives[int(ref["prim"][i])])
print("%d worlds, %d rays, %d hits, %d point-in-solid positives, %d path-traced frames, mismatches: %d" % (n_worlds, total, hits, inside, frames, bad))
sys.exit(1 if bad else 0)
