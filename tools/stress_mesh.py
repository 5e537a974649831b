#!/usr/bin/env python3
"""One-off stress run on the GPU box for the watertight triangle test and the mesh KD traversal: rays aimed exactly at mesh vertices,
edge midpoints and triangle centroids of the 69 432-triangle mesh (on-edge and on-vertex hits: the f64 fallback of the barycentrics,
first-wins tie order inside leaves), grazing rays, axis-parallel rays and rays with a finite reach — rsx_hit_batch against the oracle:
primitive, triangle, t, u, v, w, exiting and the full geometry must be identical.   python tools/stress_mesh.py [million_rays]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from oracle import oracle as orc  # noqa: E402
from source_amd import api as ns, scenes  # noqa: E402

n = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 4000000
rng = np.random.RandomState(777)
bad = total = 0
for name, builder in (("single mesh", lambda: scenes.build_c2(ns, n=132)), ("smooth + normals", lambda: scenes.build_c2(ns, n=48, smoothing=True, with_normals=True)),
                      ("15 instances", lambda: scenes.build_c3(ns, n=132))):
    built = builder()
    world = built[0]
    mesh = next(p for p in world._primitives if isinstance(p, ns.Mesh))
    v = np.asarray(mesh.data.vertices, dtype=np.float64)
    t = np.asarray(mesh.data.triangles)[:, :3]
    m = np.array(mesh.to_root().m).reshape(4, 4)
    vw = v @ m[:3, :3].T + m[:3, 3]                                        # world-space vertices of (one instance of) the mesh
    k = n // 5
    tri = t[rng.randint(len(t), size=4 * k)]
    targets = np.concatenate([
        vw[rng.randint(len(vw), size=k)],                                     # vertices
        0.5 * (vw[tri[:k, 0]] + vw[tri[:k, 1]]),                              # edge midpoints
        (vw[tri[k:2 * k, 0]] + vw[tri[k:2 * k, 1]] + vw[tri[k:2 * k, 2]]) / 3.0,   # centroids
        vw[tri[2 * k:3 * k, 0]] + 1e-7 * rng.normal(size=(k, 3)),             # a hair off a vertex
        rng.uniform(-0.3, 0.3, (n - 4 * k, 3)) + vw.mean(axis=0),             # anywhere near the mesh (grazing included)
    ])
    o = rng.normal(size=(n, 3)); o = 0.8 * o / np.linalg.norm(o, axis=1)[:, None] + vw.mean(axis=0)
    axis = rng.rand(n) < 0.1                                                 # a tenth axis-parallel through the targets
    a = rng.randint(3, size=n)
    o[axis] = targets[axis]; o[axis, a[axis]] += 1.0
    d = targets - o
    d /= np.linalg.norm(d, axis=1)[:, None]
    reach = np.where(rng.rand(n) < 0.15, np.linalg.norm(targets - o, axis=1) * rng.choice([0.999999, 1.0, 1.000001], size=n), np.inf)
    scene = world.build_accelerator()
    dev = scene.hit_batch(o, d, reach, geometry=True)
    ref = orc.hit_batch(world.flatten(), o, d, reach, geometry=True, threads=orc.max_threads())
    hit = ref["prim"] >= 0
    same = (dev["prim"] == ref["prim"]) & np.where(hit, (dev["tri"] == ref["tri"]) & (dev["t"] == ref["t"]) & (dev["exiting"] == ref["exiting"]), True)
    same &= np.where(hit[:, None], dev["uvw"] == ref["uvw"], True).all(axis=1)
    geq = (dev["geom"] == ref["geom"]) | (np.isnan(dev["geom"]) & np.isnan(ref["geom"]))
    same &= np.where(hit[:, None], geq, True).all(axis=1)
    edge = hit & ((ref["uvw"] == 0).any(axis=1))
    print("%-18s %9d rays  %9d hits  %7d with a barycentric exactly 0  mismatches %d" % (name, n, int(hit.sum()), int(edge.sum()), int((~same).sum())), flush=True)
    total += n
    bad += int((~same).sum())
print("total %d rays, mismatches %d" % (total, bad))
sys.exit(1 if bad else 0)
