#!/usr/bin/env python3
"""
Executes INTEGRATION.md's plug-point #1 against STOCK Raysect (development container only, like make_golden.py).

Imports the compiled reference from /tmp/rs_oracle (tests/golden/build_reference.sh), builds the worlds of fixtures F04 / F05 / F06 /
F07 / F11 out of stock raysect objects, asks every ray and point through `World.hit` / `World.contains` twice — once with the stock
`KDTree` accelerator, once after `world.accelerator = HipAccelerator()` (integration/raysect_hip.py: librsx behind
`raysect.core.acceleration.Accelerator`) — and compares the Intersection objects field by field, bit for bit: primitive, ray_distance,
exiting, hit / inside / outside points, normal, both transforms, and triangle / u / v / w of MeshIntersections; contains() lists in order.

Writes tests/golden/f19_binding.npz (per case: rays, hits, mismatches and a SHA-256 of the compared records — data only) which the CPU
suite recomputes through source_amd's own host walk (tests/test_host.py::test_binding_under_stock_raysect_fixture), and prints the log
kept as profiles/r06_bind_reference.txt. Nothing of the reference travels.

    bash tests/golden/build_reference.sh && python tests/golden/bind_reference.py | tee profiles/r06_bind_reference.txt
"""
import hashlib
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
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(REPO, "integration"))

import numpy as np  # noqa: E402

import raysect  # noqa: E402
assert os.path.realpath(raysect.__file__).startswith(os.path.realpath(REF)), raysect.__file__
from raysect.core import Ray, Point3D, Vector3D, translate, rotate  # noqa: E402
from raysect.core.acceleration import KDTree  # noqa: E402
from raysect.primitive import Mesh, Sphere, Box, Cylinder, Union, Intersect, Subtract  # noqa: E402
from raysect.primitive.mesh.mesh import MeshIntersection  # noqa: E402
from raysect.optical import World, ConstantSF  # noqa: E402
from raysect.optical.material import AbsorbingSurface, UniformSurfaceEmitter  # noqa: E402
from raysect.optical.material.debug import Light  # noqa: E402

import raysets  # noqa: E402
from raysect_hip import HipAccelerator  # noqa: E402
from source_amd import scenes  # noqa: E402

NS = types.SimpleNamespace(World=World, Mesh=Mesh, Sphere=Sphere, Box=Box, Cylinder=Cylinder, Union=Union, Intersect=Intersect, Subtract=Subtract,
                           Point3D=Point3D, Vector3D=Vector3D, translate=translate, rotate=rotate, ConstantSF=ConstantSF,
                           AbsorbingSurface=AbsorbingSurface, UniformSurfaceEmitter=UniformSurfaceEmitter, Light=Light)


def mat16(m):
    return [m[i, j] for i in range(4) for j in range(4)]


def record(world, prims, hit):
    """One Intersection as numbers: primitive index, t, exiting, hit / inside / outside, normal, both matrices, triangle, u, v, w."""
    if hit is None:
        return [-1.0] + [np.nan] * 50
    mesh = isinstance(hit, MeshIntersection)
    return ([float(prims.index(hit.primitive)), hit.ray_distance, float(hit.exiting)] +
            [c for p in (hit.hit_point, hit.inside_point, hit.outside_point, hit.normal) for c in (p.x, p.y, p.z)] +
            mat16(hit.world_to_primitive) + mat16(hit.primitive_to_world) +
            [float(hit.triangle) if mesh else -1.0, hit.u if mesh else 0.0, hit.v if mesh else 0.0, hit.w if mesh else 0.0])


def ask(world, prims, o, d, m, pts):
    hits = np.array([record(world, prims, world.hit(Ray(Point3D(*o[k]), Vector3D(*d[k]), float(m[k])))) for k in range(len(o))])
    inside = [[prims.index(p) for p in world.contains(Point3D(*q))] for q in pts]
    return hits, inside


def digest(hits, inside, n_world):
    """What the CPU suite recomputes: ids, t, exiting, the four geometry vectors (the transforms and MeshIntersection extras are compared
    here, between the two accelerators) and the containment lists as a [points, primitives] table."""
    table = np.zeros((len(inside), max(1, n_world)), dtype=np.uint8)
    for i, row in enumerate(inside):
        table[i, row] = 1
    h = hashlib.sha256()
    hit = hits[:, 0] >= 0
    h.update(hits[:, 0].astype(np.int32).tobytes())
    h.update(np.ascontiguousarray(hits[hit, 1:15]).tobytes())
    h.update(table.tobytes())
    return h.hexdigest()


def cases():
    m70k_v, m70k_t = scenes.displaced_sphere(132)                                    # F04's 69 432-triangle mesh as a one-primitive world
    w = World()
    mesh = Mesh(m70k_v, m70k_t, smoothing=False, closed=True, parent=w, material=AbsorbingSurface())
    o1, d1, m1 = raysets.random_outside(3000, 41)
    o2, d2, m2 = raysets.through_vertices(m70k_v, 1500, 44)
    o3, d3, m3 = raysets.along_edges(m70k_v, m70k_t, 1000, 45)
    o4, d4, m4 = raysets.axis_aligned(1000, 46, 0.1, m70k_v)
    yield ("f04_mesh_world", w, [mesh], np.concatenate([o1, o2, o3, o4]), np.concatenate([d1, d2, d3, d4]), np.concatenate([m1, m2, m3, m4]),
           raysets.points(1500, 49, 0.1))
    tr = translate(0.1, -0.2, 0.3) * rotate(25, -35, 45)                             # F05: one-primitive worlds of the analytic primitives
    f05 = {"sphere": Sphere(0.8, transform=tr), "sphere_id": Sphere(1.0),
           "box": Box(Point3D(-0.5, -0.7, -0.4), Point3D(0.6, 0.5, 0.9), transform=tr), "box_id": Box(Point3D(-0.6, -0.6, -0.6), Point3D(0.6, 0.6, 0.6)),
           "cylinder": Cylinder(0.5, 1.2, transform=tr), "cylinder_id": Cylinder(0.6, 1.0, transform=translate(0, 0, -0.5))}
    for k, (name, prim) in enumerate(f05.items()):
        w = World()
        prim.parent = w
        o, d, m = raysets.primitive_rays(3000, 70 + k)
        yield ("f05_" + name, w, [prim], o, d, m, raysets.points(1500, 90 + k, 1.2))
    w, prims = scenes.build_csg_demo(NS)                                             # F06: demos/csg.py
    o, d, m = raysets.scene_rays(6000, 101, 9.0, 4.5)
    og, dg, mg = raysets.pinhole_grid(48, (0.0, 0.0, -4.0), 75.0)
    yield ("f06_csg_demo", w, list(w.primitives), np.concatenate([o, og]), np.concatenate([d, dg]), np.concatenate([m, mg]), raysets.points(3000, 102, 4.5))
    w, prims = scenes.build_mixed(NS)                                                # F07: instances + analytic + CSG, coincident spheres
    o, d, m = raysets.scene_rays(12000, 111, 6.0, 2.2)
    yield ("f07_mixed_world", w, list(w.primitives), o, d, m, raysets.points(4000, 112, 2.0))
    for name, (w, prims) in scenes.build_edge_worlds(NS).items():                    # F11: Appendix B edge semantics
        o, d, m = scenes.edge_rays(name)
        yield ("f11_" + name, w, list(w.primitives), o, d, m, np.concatenate([o, o + 0.25 * d]))


def main():
    out = {}
    total = bad = 0
    print("stock Raysect: %s (version %s)" % (os.path.dirname(raysect.__file__), getattr(raysect, "__version__", "?")))
    print("%-22s %7s %7s %9s %9s  %12s %12s  %s" % ("case", "rays", "hits", "hit diff", "cont diff", "stock us/call", "librsx us/call", "sha256[:16]"))
    for name, world, prims, o, d, m, pts in cases():
        assert isinstance(world.accelerator, KDTree)
        world.build_accelerator()
        t0 = time.perf_counter()
        ref_hits, ref_inside = ask(world, prims, o, d, m, pts)
        t_ref = time.perf_counter() - t0
        world.accelerator = HipAccelerator()                                         # world.pyx:67-70
        assert isinstance(world.accelerator, HipAccelerator)
        world.build_accelerator()                                                    # Accelerator.build(primitives): flatten + rsx_host_scene_create
        t0 = time.perf_counter()
        hip_hits, hip_inside = ask(world, prims, o, d, m, pts)
        t_hip = time.perf_counter() - t0
        same = (ref_hits == hip_hits) | (np.isnan(ref_hits) & np.isnan(hip_hits))    # every field, bit for bit (== on f64: -0.0 / +0.0 differ nowhere here: checked below)
        signs = np.signbit(ref_hits) == np.signbit(hip_hits)
        hit_diff = int((~(same & signs).all(axis=1)).sum())
        cont_diff = sum(1 for a, b in zip(ref_inside, hip_inside) if a != b)
        dg = digest(ref_hits, ref_inside, len(prims))
        n_calls = len(o) + len(pts)
        print("%-22s %7d %7d %9d %9d  %12.2f %12.2f  %s" % (name, len(o), int((ref_hits[:, 0] >= 0).sum()), hit_diff, cont_diff,
                                                          1e6 * t_ref / n_calls, 1e6 * t_hip / n_calls, dg[:16]))
        out[name] = np.array([len(o), int((ref_hits[:, 0] >= 0).sum()), hit_diff, cont_diff, len(pts)], dtype=np.int64)
        out[name + "_sha"] = np.frombuffer(bytes.fromhex(dg), dtype=np.uint8)
        total += len(o) + len(pts)
        bad += hit_diff + cont_diff
    print("%d World.hit / World.contains calls through each accelerator, %d differences" % (total, bad))
    print("(us/call: wall time of the script's loop per call, i.e. with the Ray / Point3D construction and the 51-number record it keeps of every Intersection)")
    assert bad == 0
    np.savez_compressed(os.path.join(HERE, "f19_binding.npz"), **out)
    print("wrote tests/golden/f19_binding.npz")


if __name__ == "__main__":
    main()
