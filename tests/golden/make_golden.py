#!/usr/bin/env python3
"""
Golden-vector generator (development container only).

Imports the *compiled reference* (raysect/source built out-of-tree by build_reference.sh,
default /tmp/rs_oracle) and records input->output vectors for every row of SURVEY.md §8(a)
into tests/golden/*.npz. Only these data files are committed; no reference source, bytecode
or binaries enter the repo, and nothing in the GPU tests reads /root/reference.

    bash tests/golden/build_reference.sh && python tests/golden/make_golden.py
"""
import hashlib
import io
import os
import random as pyrandom
import sys
import types

os.environ.setdefault("MPLBACKEND", "Agg")
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("RAYSECT_REF_BUILD", "/tmp/rs_oracle")
sys.path.insert(0, REF)
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402

import raysect  # noqa: E402
assert os.path.realpath(raysect.__file__).startswith(os.path.realpath(REF)), raysect.__file__
from raysect.core import Ray as CoreRay, Point3D, Vector3D, AffineMatrix3D, translate, rotate, rotate_x, rotate_y, rotate_z, rotate_vector  # noqa: E402
from raysect.core import SerialEngine, BoundingBox3D, StatsArray1D, StatsArray3D  # noqa: E402
from raysect.core.math import random as rsrandom  # noqa: E402
from raysect.core.acceleration.kdtree import _PrimitiveKDTree  # noqa: E402
from raysect.primitive import Mesh, Sphere, Box, Cylinder, Union, Intersect, Subtract  # noqa: E402
from raysect.optical import World, ConstantSF, InterpolatedSF, Ray as OpticalRay  # noqa: E402
from raysect.optical.material import AbsorbingSurface, UniformSurfaceEmitter, UniformVolumeEmitter, NullMaterial, Lambert, Dielectric, Sellmeier  # noqa: E402
from raysect.optical.material.debug import Light  # noqa: E402
from raysect.optical.observer import PinholeCamera, FullFrameSampler2D, SpectralRadiancePipeline2D, SpectralPowerPipeline2D  # noqa: E402
from raysect.optical.observer import RGBPipeline2D, RGBAdaptiveSampler2D  # noqa: E402

from source_amd import scenes  # noqa: E402
import raysets  # noqa: E402

NS = types.SimpleNamespace(
    World=World, Mesh=Mesh, Sphere=Sphere, Box=Box, Cylinder=Cylinder, Union=Union, Intersect=Intersect,
    Subtract=Subtract, Point3D=Point3D, Vector3D=Vector3D, translate=translate, rotate=rotate,
    ConstantSF=ConstantSF, InterpolatedSF=InterpolatedSF, AbsorbingSurface=AbsorbingSurface,
    UniformSurfaceEmitter=UniformSurfaceEmitter, UniformVolumeEmitter=UniformVolumeEmitter, NullMaterial=NullMaterial, Light=Light, Lambert=Lambert, Dielectric=Dielectric, Sellmeier=Sellmeier,
    PinholeCamera=PinholeCamera,
    FullFrameSampler2D=FullFrameSampler2D, SpectralRadiancePipeline2D=SpectralRadiancePipeline2D,
    SpectralPowerPipeline2D=SpectralPowerPipeline2D, RGBPipeline2D=RGBPipeline2D, RGBAdaptiveSampler2D=RGBAdaptiveSampler2D)


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("%-28s %8.1f KiB" % (name, os.path.getsize(path) / 1024))


def mat(m):
    return np.array([[m[i, j] for j in range(4)] for i in range(4)], dtype=np.float64)


def mkray(o, d, m):
    return CoreRay(Point3D(*o), Vector3D(*d), float(m))


# ------------------------------------------------------------------ F0 host math
def f00_math():
    rng = np.random.RandomState(11)
    n = 64
    p = rng.uniform(-3, 3, (n, 9))
    p[:, 3:6] = rng.uniform(-180, 180, (n, 3))
    p[:8, 3:6] = [[0, 0, 0], [90, 0, 0], [0, 90, 0], [0, 0, 90], [165, 0, 0], [0, -12, 0], [30, -20, 0], [50, 50, 0]]
    tr, inv, chain, chain_inv, rvec = [], [], [], [], []
    pts, vecs = [], []
    for k in range(n):
        m = translate(*p[k, 0:3]) * rotate(*p[k, 3:6])
        tr.append(mat(m)); inv.append(mat(m.inverse()))
        c = m * rotate_x(p[k, 6] * 30) * translate(*p[k, 6:9]) * rotate_z(p[k, 7] * 50) * rotate_y(p[k, 8] * 70)
        chain.append(mat(c)); chain_inv.append(mat(c.inverse()))
        rvec.append(mat(rotate_vector(p[k, 3], Vector3D(*p[k, 0:3]))))
        q = Point3D(*p[k, 6:9]).transform(c.inverse())
        v = Vector3D(*p[k, 0:3]).transform(c)
        vn = Vector3D(*p[k, 0:3]).normalise()
        pts.append([q.x, q.y, q.z]); vecs.append([v.x, v.y, v.z, vn.x, vn.y, vn.z])
    save("f00_math", params=p, tr=np.array(tr), inv=np.array(inv), chain=np.array(chain),
         chain_inv=np.array(chain_inv), rvec=np.array(rvec), pts=np.array(pts), vecs=np.array(vecs))


# ------------------------------------------------------------------ F1 MT19937-64
def f01_mt():
    seeds = np.array([1, 7, 1234567890, 2**63 + 12345], dtype=np.uint64)
    out = np.empty((len(seeds), 1000))
    for i, s in enumerate(seeds):
        rsrandom.seed(int(s))
        out[i] = [rsrandom.uniform() for _ in range(1000)]
    save("f01_mt", seeds=seeds, uniforms=out)


# ------------------------------------------------------------------ F2 AABB slab
def f02_aabb():
    rng = np.random.RandomState(2)
    n = 4000
    lo = rng.uniform(-1, 0, (n, 3)); hi = lo + rng.uniform(0, 2, (n, 3))
    o, d, _ = raysets.scene_rays(n, 21, 3.0, 1.5)
    d[:200, 0] = 0.0; d[200:400, 1] = 0.0; d[400:500, :2] = 0.0; d[400:500, 2] = 1.0
    o[500:600] = 0.5 * (lo[500:600] + hi[500:600])            # origin inside
    o[600:650, 0] = lo[600:650, 0]                           # origin on a face plane
    res = np.empty((n, 3))
    for k in range(n):
        b = BoundingBox3D(Point3D(*lo[k]), Point3D(*hi[k]))
        h, f, bk = b.full_intersection(mkray(o[k], d[k], np.inf))
        res[k] = [h, f, bk]
    save("f02_aabb", lower=lo, upper=hi, origin=o, direction=d, result=res)


# ------------------------------------------------------------------ F3 KD build (RSM blobs)
def rsm_bytes(mesh):
    f = io.BytesIO(); mesh.data.save(f); return f.getvalue()


def small_meshes():
    return {
        "cube": scenes.cube_mesh(),
        "sphere8": scenes.displaced_sphere(8, radius=1.0),
        "blob24": scenes.displaced_sphere(24, radius=0.5),
        "fan500": scenes.fan_mesh(500),
    }


def f03_kd():
    out = {}
    for name, (v, t) in small_meshes().items():
        m = Mesh(v, t, smoothing=False, closed=(name != "fan500"))
        out[name] = np.frombuffer(rsm_bytes(m), dtype=np.uint8)
        fn = m.data.face_normals
        out[name + "_face_normals"] = fn
    # non-default kd parameters
    v, t = scenes.displaced_sphere(24, radius=0.5)
    m = Mesh(v, t, smoothing=False, kdtree_max_depth=9, kdtree_min_items=4, kdtree_hit_cost=20.0, kdtree_empty_bonus=0.1)
    out["blob24_params"] = np.frombuffer(rsm_bytes(m), dtype=np.uint8)
    # 70k stand-in: digest only (blob is ~8 MB)
    v, t = scenes.displaced_sphere(132)
    m = Mesh(v, t, smoothing=False)
    b = rsm_bytes(m)
    out["m70k_sha256"] = np.frombuffer(hashlib.sha256(b).digest(), dtype=np.uint8)
    out["m70k_len"] = np.array([len(b)])
    out["m70k_ntri"] = np.array([m.data.face_normals.shape[0]])
    out["m70k_face_normals_sha256"] = np.frombuffer(hashlib.sha256(m.data.face_normals.tobytes()).digest(), dtype=np.uint8)
    # world-level trees (kd blob after the pickled primitive list)
    for name, builder in (("mixed", scenes.build_mixed), ("csg", scenes.build_csg_demo)):
        world = builder(NS)[0]
        tree = _PrimitiveKDTree(world.primitives)
        out["world_" + name] = np.frombuffer(tree.__getstate__()[1], dtype=np.uint8)
        boxes = []
        for p in world.primitives:
            bb = p.bounding_box()
            boxes.append([bb.lower.x, bb.lower.y, bb.lower.z, bb.upper.x, bb.upper.y, bb.upper.z])
        out["world_" + name + "_boxes"] = np.array(boxes)
    save("f03_kd", **out)
    return m  # the 70k mesh, reused below


# ------------------------------------------------------------------ F4 mesh hits
def mesh_hit_arrays(mesh, o, d, m, full=False):
    n = len(o)
    tri = np.full(n, -1, dtype=np.int32); t = np.full(n, np.nan)
    uvw = np.zeros((n, 3), dtype=np.float32); ex = np.zeros(n, dtype=np.uint8)
    extra = np.full((n, 12), np.nan) if full else None
    for k in range(n):
        i = mesh.hit(mkray(o[k], d[k], m[k]))
        if i is not None:
            tri[k] = i.triangle; t[k] = i.ray_distance; uvw[k] = (i.u, i.v, i.w); ex[k] = i.exiting
            if full:
                extra[k] = [i.hit_point.x, i.hit_point.y, i.hit_point.z, i.inside_point.x, i.inside_point.y, i.inside_point.z,
                            i.outside_point.x, i.outside_point.y, i.outside_point.z, i.normal.x, i.normal.y, i.normal.z]
    return tri, t, uvw, ex, extra


def root_sequences(prim, o, d, m, cap=64):
    """hit() + repeated next_intersection(): ragged list of (t, exiting) per ray."""
    counts = np.zeros(len(o), dtype=np.int32); ts = []; exs = []
    for k in range(len(o)):
        i = prim.hit(mkray(o[k], d[k], m[k]))
        c = 0
        while i is not None and c < cap:
            ts.append(i.ray_distance); exs.append(i.exiting); c += 1
            i = prim.next_intersection()
        counts[k] = c
    return counts, np.array(ts), np.array(exs, dtype=np.uint8)


def f04_mesh(mesh70k):
    v, t = scenes.displaced_sphere(132)
    sets = {
        "grid": raysets.pinhole_grid(96),
        "outside": raysets.random_outside(6000, 41),
        "outside_raw": raysets.random_outside(2000, 42, unit=False),
        "interior": raysets.random_interior(4000, 43),
        "vertices": raysets.through_vertices(v, 4000, 44),
        "edges": raysets.along_edges(v, t, 3000, 45),
        "axis": raysets.axis_aligned(3000, 46, 0.1, v),
    }
    out = {}
    for name, (o, d, m) in sets.items():
        tri, tt, uvw, ex, extra = mesh_hit_arrays(mesh70k, o, d, m, full=(name in ("outside", "interior")))
        out[name + "_tri"] = tri; out[name + "_t"] = tt; out[name + "_uvw"] = uvw; out[name + "_ex"] = ex
        if extra is not None:
            out[name + "_extra"] = extra
        print("   ", name, "hits", int((tri >= 0).sum()), "/", len(tri))
    # result-dependent rays: origin on the surface, max_distance around the hit distance
    o, d, m = sets["outside"]
    tt = out["outside_t"]; hit = np.where(out["outside_tri"] >= 0)[0][:1500]
    o2 = o[hit] + d[hit] * tt[hit, None]
    out["surf_o"] = o2; out["surf_d"] = d[hit]
    tri, t2, uvw, ex, _ = mesh_hit_arrays(mesh70k, o2, d[hit], np.full(len(hit), np.inf))
    out["surf_tri"] = tri; out["surf_t"] = t2; out["surf_uvw"] = uvw; out["surf_ex"] = ex
    md = np.concatenate([tt[hit[:500]], tt[hit[500:1000]] * (1 - 1e-7), tt[hit[1000:1500]] * (1 + 1e-7)])
    out["maxd_m"] = md
    tri, t3, uvw, ex, _ = mesh_hit_arrays(mesh70k, o[hit], d[hit], md)
    out["maxd_idx"] = hit; out["maxd_tri"] = tri; out["maxd_t"] = t3; out["maxd_uvw"] = uvw
    # full root sequences (next_intersection protocol)
    o, d, m = raysets.random_outside(1500, 47)
    c, ts, exs = root_sequences(mesh70k, o, d, m)
    out["seq_counts"] = c; out["seq_t"] = ts; out["seq_ex"] = exs
    # instanced + transformed + smoothed mesh (normals path)
    vn = scenes.vertex_normals(v, t)
    sm = Mesh(v, np.concatenate([t, t], axis=1), vn, smoothing=True, transform=translate(0.01, -0.02, 0.03) * rotate(33, 21, -14))
    o, d, m = raysets.random_outside(3000, 48)
    tri, tt, uvw, ex, extra = mesh_hit_arrays(sm, o, d, m, full=True)
    out["smooth_tri"] = tri; out["smooth_t"] = tt; out["smooth_uvw"] = uvw; out["smooth_ex"] = ex; out["smooth_extra"] = extra
    out["smooth_to_local"] = mat(sm.to_local()); out["smooth_to_root"] = mat(sm.to_root())
    # contains()
    pts = raysets.points(4000, 49, 0.1)
    out["contains"] = np.array([mesh70k.contains(Point3D(*p)) for p in pts], dtype=np.uint8)
    # 1M-ray digest
    o, d, m = raysets.random_outside(1000000, 50)
    tri, tt, uvw, ex, _ = mesh_hit_arrays(mesh70k, o, d, m)
    h = hashlib.sha256(); h.update(tri.tobytes()); h.update(tt.tobytes()); h.update(uvw.tobytes()); h.update(ex.tobytes())
    out["digest_1m"] = np.frombuffer(h.digest(), dtype=np.uint8)
    out["digest_1m_hits"] = np.array([(tri >= 0).sum()])
    save("f04_mesh", **out)


def f04b_small_meshes():
    out = {}
    for name, (v, t) in small_meshes().items():
        mesh = Mesh(v, t, smoothing=False, closed=(name != "fan500"))
        sc = float(np.abs(v).max())
        o, d, m = raysets.random_outside(1500, 61, 3.0 * sc, 1.0 * sc)
        o2, d2, m2 = raysets.through_vertices(v, 600, 62)
        o3, d3, m3 = raysets.axis_aligned(600, 63, sc, v)
        o3[np.arange(600), np.argmax(np.abs(d3), axis=1)] *= 8 * sc
        o = np.concatenate([o, o2 * 4 * sc, o3]); d = np.concatenate([d, (v[np.random.RandomState(62).randint(0, len(v), 600)] - o2 * 4 * sc), d3]); m = np.concatenate([m, m2, m3])
        tri, tt, uvw, ex, _ = mesh_hit_arrays(mesh, o, d, m)
        out[name + "_o"] = o; out[name + "_d"] = d
        out[name + "_tri"] = tri; out[name + "_t"] = tt; out[name + "_uvw"] = uvw; out[name + "_ex"] = ex
    save("f04b_small_meshes", **out)


# ------------------------------------------------------------------ F5 analytic primitives
def prim_records(prim, o, d, m):
    n = len(o)
    rec = np.full((n, 2, 14), np.nan)   # [first|next]: t, exiting, hit(3), inside(3), outside(3), normal(3)
    for k in range(n):
        i = prim.hit(mkray(o[k], d[k], m[k]))
        for j in range(2):
            if i is None:
                break
            rec[k, j] = [i.ray_distance, i.exiting, i.hit_point.x, i.hit_point.y, i.hit_point.z,
                         i.inside_point.x, i.inside_point.y, i.inside_point.z,
                         i.outside_point.x, i.outside_point.y, i.outside_point.z, i.normal.x, i.normal.y, i.normal.z]
            i = prim.next_intersection()
    return rec


def f05_primitives():
    out = {}
    tr = translate(0.1, -0.2, 0.3) * rotate(25, -35, 45)
    prims = {
        "sphere": Sphere(0.8, transform=tr),
        "sphere_id": Sphere(1.0),
        "box": Box(Point3D(-0.5, -0.7, -0.4), Point3D(0.6, 0.5, 0.9), transform=tr),
        "box_id": Box(Point3D(-0.6, -0.6, -0.6), Point3D(0.6, 0.6, 0.6)),
        "cylinder": Cylinder(0.5, 1.2, transform=tr),
        "cylinder_id": Cylinder(0.6, 1.0, transform=translate(0, 0, -0.5)),
    }
    for k, (name, p) in enumerate(prims.items()):
        o, d, m = raysets.primitive_rays(3000, 70 + k)
        out[name] = prim_records(p, o, d, m)
        out[name + "_to_local"] = mat(p.to_local())
        bb = p.bounding_box()
        out[name + "_bbox"] = np.array([bb.lower.x, bb.lower.y, bb.lower.z, bb.upper.x, bb.upper.y, bb.upper.z])
        pts = raysets.points(2000, 90 + k, 1.2)
        out[name + "_contains"] = np.array([p.contains(Point3D(*q)) for q in pts], dtype=np.uint8)
    save("f05_primitives", **out)


# ------------------------------------------------------------------ F6/F7 CSG + world level
def world_records(world, o, d, m):
    n = len(o)
    prims = list(world.primitives)
    idx = np.full(n, -1, dtype=np.int32); rec = np.full((n, 14), np.nan)
    for k in range(n):
        i = world.hit(mkray(o[k], d[k], m[k]))
        if i is not None:
            idx[k] = prims.index(i.primitive)
            rec[k] = [i.ray_distance, i.exiting, i.hit_point.x, i.hit_point.y, i.hit_point.z,
                      i.inside_point.x, i.inside_point.y, i.inside_point.z,
                      i.outside_point.x, i.outside_point.y, i.outside_point.z, i.normal.x, i.normal.y, i.normal.z]
    return idx, rec


def contains_records(world, pts):
    prims = list(world.primitives)
    out = np.zeros((len(pts), len(prims)), dtype=np.uint8)
    for k, p in enumerate(pts):
        for q in world.contains(Point3D(*p)):
            out[k, prims.index(q)] = 1
    return out


def f06_csg():
    out = {}
    world, prims = scenes.build_csg_demo(NS)
    o, d, m = raysets.scene_rays(12000, 101, 9.0, 4.5)
    og, dg, mg = raysets.pinhole_grid(64, (0.0, 0.0, -4.0), 75.0)
    o = np.concatenate([o, og]); d = np.concatenate([d, dg]); m = np.concatenate([m, mg])
    idx, rec = world_records(world, o, d, m)
    out["world_idx"] = idx; out["world_rec"] = rec
    print("    csg world hits by prim:", np.bincount(idx[idx >= 0]))
    # direct root sequences on the first CSG object and the lens
    for name, p in (("obj0", prims[0]), ("lens", prims[4])):
        c, ts, exs = root_sequences(p, o[:4000], d[:4000], np.full(4000, np.inf))
        out[name + "_counts"] = c; out[name + "_t"] = ts; out[name + "_ex"] = exs
    pts = raysets.points(4000, 102, 4.5)
    out["contains"] = contains_records(world, pts)
    save("f06_csg", **out)


def f07_world():
    out = {}
    world, prims = scenes.build_mixed(NS)
    o, d, m = raysets.scene_rays(20000, 111, 6.0, 2.2)
    idx, rec = world_records(world, o, d, m)
    out["idx"] = idx; out["rec"] = rec
    print("    mixed world hits by prim:", np.bincount(idx[idx >= 0]))
    pts = raysets.points(6000, 112, 2.0)
    out["contains"] = contains_records(world, pts)
    out["to_local"] = np.array([mat(p.to_local()) for p in prims])
    save("f07_world", **out)


# ------------------------------------------------------------------ F8 camera rays
def f08_camera():
    world = World()
    cam = PinholeCamera((48, 32), fov=52.0, parent=world, transform=translate(0.3, -0.2, 1.0) * rotate(20, 10, 5))
    tmpl = OpticalRay()
    rsrandom.seed(77)
    rows = []
    for (x, y) in [(0, 0), (47, 31), (13, 7), (24, 16), (5, 30)]:
        for ray, w in cam._generate_rays(x, y, tmpl, 16):
            rows.append([x, y, ray.origin.x, ray.origin.y, ray.origin.z, ray.direction.x, ray.direction.y, ray.direction.z, w])
    rsrandom.seed(77)
    u = np.array([rsrandom.uniform() for _ in range(160)])
    save("f08_camera", rows=np.array(rows), uniforms=u, to_root=mat(cam.to_root()))


# ------------------------------------------------------------------ F9 statistics
def f09_stats():
    rng = np.random.RandomState(9)
    nseq, length = 40, 50
    x = rng.lognormal(size=(nseq, length)) * rng.choice([1e-6, 1.0, 1e6], (nseq, 1))
    x[3] = 2.5; x[4, 1:] = x[4, 0]
    states = np.empty((nseq, length, 3))
    for s in range(nseq):
        a = StatsArray1D(1)
        for k in range(length):
            a.add_sample(0, x[s, k])
            states[s, k] = [a.mean[0], a.variance[0], a.samples[0]]
    m = 400
    ma, mb = rng.normal(size=m), rng.normal(size=m)
    va, vb = rng.uniform(0, 2, m), rng.uniform(-0.1, 2, m)
    na, nb = rng.randint(0, 6, m), rng.randint(1, 6, m)
    na[:50] = rng.randint(100, 10000, 50); nb[:50] = rng.randint(100, 10000, 50)
    comb = np.empty((m, 3))
    for k in range(m):
        f = StatsArray3D(1, 1, 1)
        f.mean[0, 0, 0] = ma[k]; f.variance[0, 0, 0] = va[k] if na[k] > 1 else 0.0; f.samples[0, 0, 0] = na[k]
        f.combine_samples(0, 0, 0, mb[k], vb[k], int(nb[k]))
        comb[k] = [f.mean[0, 0, 0], f.variance[0, 0, 0], f.samples[0, 0, 0]]
    va = np.where(na > 1, va, 0.0)
    save("f09_stats", x=x, states=states, ma=ma, va=va, na=na, mb=mb, vb=vb, nb=nb, comb=comb)


# ------------------------------------------------------------------ F10 frames (observe())
def observe_frame(cam, pipe, seed):
    pyrandom.seed(seed); rsrandom.seed(seed)
    cam.render_engine = SerialEngine()
    cam.observe()
    f = pipe.frame
    return np.array(f.mean), np.array(f.variance), np.array(f.samples)


def f10_frames():
    out = {}
    world, mesh, box = scenes.build_c2(NS, n=132)
    cam, pipe = scenes.c2_camera(NS, world, (40, 40), spp=4, bins=15)
    out["c2_mean"], out["c2_var"], out["c2_n"] = observe_frame(cam, pipe, 1)
    # second accumulate pass (combine_samples path with n>1 on both sides)
    pyrandom.seed(2); rsrandom.seed(2); cam.observe()
    out["c2_mean2"], out["c2_var2"], out["c2_n2"] = np.array(pipe.frame.mean), np.array(pipe.frame.variance), np.array(pipe.frame.samples)
    # 1 spp, spectral slicing (3 rays over 7 bins), power pipeline with sensitivity, smoothing normals
    world, mesh, box = scenes.build_c2(NS, n=48, smoothing=True, with_normals=True)
    pipe = SpectralPowerPipeline2D()
    cam = PinholeCamera((24, 36), fov=45, sensitivity=2.5, parent=world, pipelines=[pipe], frame_sampler=FullFrameSampler2D(),
                        transform=translate(0, 0.16, -0.4) * rotate(0, -12, 0))
    cam.pixel_samples = 1; cam.spectral_bins = 7; cam.spectral_rays = 3; cam.quiet = True
    cam.min_wavelength = 400.0; cam.max_wavelength = 700.0
    out["c2s_mean"], out["c2s_var"], out["c2s_n"] = observe_frame(cam, pipe, 3)
    # csg demo scene
    world, prims = scenes.build_csg_demo(NS)
    cam, pipe = scenes.csg_camera(NS, world, (32, 32), spp=6, bins=5)
    out["csg_mean"], out["csg_var"], out["csg_n"] = observe_frame(cam, pipe, 4)
    # instanced scene
    world = scenes.build_c3(NS, n=32)[0]
    cam, pipe = scenes.c3_camera(NS, world, (32, 32), spp=3, bins=4)
    out["c3_mean"], out["c3_var"], out["c3_n"] = observe_frame(cam, pipe, 5)
    # spectral function sampling used for the material tables
    sf = InterpolatedSF([300, 490, 510, 590, 610, 800], np.array([0.0, 0.1, 1.0, 0.7, 0.2, 0.4]))
    out["sf_interp_15"] = sf.sample(375.0, 740.0, 15)
    out["sf_interp_3"] = sf.sample(480.0, 520.0, 3)
    out["sf_interp_wide"] = sf.sample(200.0, 900.0, 9)
    out["sf_const"] = ConstantSF(0.75).sample(375.0, 740.0, 4)
    save("f10_frames", **out)


def f11_edges():
    """Edge semantics: empty world, coincident primitives, t == max_distance, origins on surfaces,
    axis-parallel grazing rays; odd frame shapes (1x1, 3x5 with one bin) and a masked frame sampler (ragged task list)."""
    out = {}
    for name, (world, prims) in scenes.build_edge_worlds(NS).items():
        o, d, m = scenes.edge_rays(name)
        out[name + "_idx"], out[name + "_rec"] = world_records(world, o, d, m)
        pts = np.concatenate([o, o + 0.25 * d])
        out[name + "_contains"] = contains_records(world, pts) if prims else np.zeros((len(pts), 0), dtype=np.uint8)
    world, mesh, box = scenes.build_c2(NS, n=24)
    for tag, pixels, spp, bins, mask in (("one", (1, 2), 3, 2, None), ("odd", (3, 5), 2, 1, None),
                                         ("mask", (9, 7), 2, 3, (np.add.outer(np.arange(9), np.arange(7)) % 3 != 0))):
        cam, pipe = scenes.edge_camera(NS, world, pixels, spp, bins, mask)
        out[tag + "_mean"], out[tag + "_var"], out[tag + "_n"] = observe_frame(cam, pipe, 21)
    save("f11_edges", **out)


def f12_volumes():
    """Transparent boundaries and volume emission (NullMaterial, UniformVolumeEmitter): deterministic continuation rays
    (material.pyx:118-178) and the per-segment volume pass (ray.pyx:422-455, homogeneous.pyx:55-102)."""
    out = {}
    world, prims = scenes.build_volumes(NS)
    cam, pipe = scenes.volumes_camera(NS, world)
    out["mean"], out["var"], out["n"] = observe_frame(cam, pipe, 31)
    pyrandom.seed(32); rsrandom.seed(32); cam.observe()                       # accumulate pass
    out["mean2"], out["var2"], out["n2"] = np.array(pipe.frame.mean), np.array(pipe.frame.variance), np.array(pipe.frame.samples)
    world, prims = scenes.build_volumes(NS, enclosed=False)                   # paths that leave the scene (ray.pyx:389-391)
    cam, pipe = scenes.volumes_camera(NS, world)
    out["open_mean"], out["open_var"], out["open_n"] = observe_frame(cam, pipe, 33)
    save("f12_volumes", **out)


def f13_lambert():
    """Stochastic secondary rays: Lambert (lambert.pyx:76-104 under ContinuousBSDF.evaluate_surface, material.pyx:286-361, importance
    sampling off), Russian roulette and the depth limit (ray.pyx:382-388), mixed with null surfaces and volume emission. SerialEngine
    consumes ONE MT19937-64 stream in execution order, which the oracle's orc_render_pinhole_mt replays."""
    out = {}
    world, prims = scenes.build_lambert(NS)
    cam, pipe = scenes.lambert_camera(NS, world)
    out["mean"], out["var"], out["n"] = observe_frame(cam, pipe, 41)
    pyrandom.seed(42); rsrandom.seed(42); cam.observe()                       # accumulate pass
    out["mean2"], out["var2"], out["n2"] = np.array(pipe.frame.mean), np.array(pipe.frame.variance), np.array(pipe.frame.samples)
    world, prims = scenes.build_lambert(NS, with_volume=False)                # observer defaults: roulette 0.01 from depth 3, max depth 500
    cam, pipe = scenes.lambert_camera(NS, world, pixels=(12, 10), spp=3, bins=3, extinction=(0.01, 3, 500))
    out["deep_mean"], out["deep_var"], out["deep_n"] = observe_frame(cam, pipe, 43)
    cam, pipe = scenes.lambert_camera(NS, world, pixels=(12, 10), spp=3, bins=4, extinction=(0.3, 1, 3))   # roulette from the first daughter on, shallow depth limit
    cam.spectral_rays = 2
    out["rr0_mean"], out["rr0_var"], out["rr0_n"] = observe_frame(cam, pipe, 44)
    save("f13_lambert", **out)


def f14_glass():
    """Dielectric (dielectric.pyx:125-328): Fresnel-weighted refraction / reflection, total internal reflection, transmission_only,
    Sellmeier index averaged per spectral slice, Beer-Lambert attenuation in the volume pass; mixed with Lambert and emitters."""
    out = {}
    world, prims = scenes.build_glass(NS)
    cam, pipe = scenes.glass_camera(NS, world)
    out["mean"], out["var"], out["n"] = observe_frame(cam, pipe, 51)
    pyrandom.seed(52); rsrandom.seed(52); cam.observe()
    out["mean2"], out["var2"], out["n2"] = np.array(pipe.frame.mean), np.array(pipe.frame.variance), np.array(pipe.frame.samples)
    world, prims = scenes.build_glass(NS, unit_transmission=True)
    cam, pipe = scenes.glass_camera(NS, world, pixels=(16, 12), spp=3, bins=4, spectral_rays=2, extinction=(0.01, 3, 500))
    out["clear_mean"], out["clear_var"], out["clear_n"] = observe_frame(cam, pipe, 53)
    bk7 = Sellmeier(1.03961212, 0.231792344, 1.01046945, 6.00069867e-3, 2.00179144e-2, 1.03560653e2)
    out["sellmeier_avg"] = np.array([bk7.average(375.0, 740.0), bk7.average(375.0, 496.0), bk7.average(700.0, 703.5), bk7.evaluate(589.3)])
    out["sellmeier_sample"] = bk7.sample(375.0, 740.0, 7)
    save("f14_glass", **out)


def f15_importance():
    """Multiple importance sampling (ContinuousBSDF.evaluate_surface, material.pyx:327-352; ImportanceManager, world.pyx:47-230):
    emitters carry importance 1 by default, a second emitter gets importance 3, the camera starts inside one bounding sphere's reach."""
    out = {}
    world, prims = scenes.build_lambert(NS)
    prims[5].material.importance = 3.0
    world2_light = NS.Sphere(0.12, world, NS.translate(-0.6, 0.5, 1.2), NS.UniformSurfaceEmitter(NS.ConstantSF(1.0), 4.0))
    cam, pipe = scenes.lambert_camera(NS, world)
    cam.ray_importance_sampling = True
    cam.ray_important_path_weight = 0.25
    out["mean"], out["var"], out["n"] = observe_frame(cam, pipe, 61)
    pyrandom.seed(62); rsrandom.seed(62); cam.observe()
    out["mean2"], out["var2"], out["n2"] = np.array(pipe.frame.mean), np.array(pipe.frame.variance), np.array(pipe.frame.samples)
    cam.ray_important_path_weight = 0.9                                        # mostly light-directed samples, many below the surface
    cam.ray_extinction_prob, cam.ray_extinction_min_depth, cam.ray_max_depth = 0.01, 3, 500
    pipe2 = SpectralRadiancePipeline2D(); cam.pipelines = [pipe2]
    out["heavy_mean"], out["heavy_var"], out["heavy_n"] = observe_frame(cam, pipe2, 63)
    # ImportanceManager.sample / pdf on their own (world.pyx:150-230): 40 origins (the first inside a bounding sphere), seeded stream
    rng = np.random.RandomState(5)
    pts = rng.uniform(-0.9, 0.9, (40, 3)); pts[:, 2] += 1.0
    pts[0] = [0.3, 0.3, 1.05]
    dirs = rng.normal(size=(40, 3)); dirs /= np.linalg.norm(dirs, axis=1)[:, None]
    out["im_pts"], out["im_dirs"] = pts, dirs
    out["im_pdf"] = np.array([world.important_direction_pdf(Point3D(*p), Vector3D(*d)) for p, d in zip(pts, dirs)])
    rsrandom.seed(99)
    samples, pdf2 = [], []
    for p in pts:
        v = world.important_direction_sample(Point3D(*p))
        samples.append([v.x, v.y, v.z]); pdf2.append(world.important_direction_pdf(Point3D(*p), v))
    out["im_samples"], out["im_pdf2"] = np.array(samples), np.array(pdf2)
    save("f15_importance", **out)


def f16_rgb():
    """RGBPipeline2D / XYZPixelProcessor (rgb.pyx:48-289, 534-562; colour.pyx:123-187) next to a spectral pipeline on the same
    rays, three spectral slices, sensitivity, two accumulating passes; and RGBAdaptiveSampler2D's task lists (sampler2d.pyx:697-896)
    for the resulting frames."""
    from raysect.optical.colour import resample_ciexyz, ciexyz_to_srgb
    out = {}
    out["xyz_7"] = np.array(resample_ciexyz(375.0, 740.0, 7))
    out["xyz_slice"] = np.array(resample_ciexyz(496.0, 618.0, 5))
    out["srgb"] = np.array([ciexyz_to_srgb(*v) for v in ((0.2, 0.3, 0.1), (0.001, 0.002, 0.0005), (0.9, 1.0, 1.2), (0.0, 0.0, 0.0))])
    world, mesh, box = scenes.build_c2(NS, n=48, smoothing=True, with_normals=True)
    rgb = RGBPipeline2D(display_progress=False)
    spectral = SpectralPowerPipeline2D(display_progress=False) if "display_progress" in SpectralPowerPipeline2D.__init__.__doc__ else SpectralPowerPipeline2D()
    cam = PinholeCamera((20, 16), fov=45, sensitivity=2.5, parent=world, pipelines=[rgb, spectral], frame_sampler=FullFrameSampler2D(),
                        transform=translate(0, 0.16, -0.4) * rotate(0, -12, 0))
    cam.pixel_samples = 3; cam.spectral_bins = 9; cam.spectral_rays = 3; cam.quiet = True
    cam.min_wavelength = 400.0; cam.max_wavelength = 700.0
    pyrandom.seed(71); rsrandom.seed(71); cam.render_engine = SerialEngine(); cam.observe()
    f = rgb.xyz_frame
    out["xyz_mean"], out["xyz_var"], out["xyz_n"] = np.array(f.mean), np.array(f.variance), np.array(f.samples)
    out["spec_mean"] = np.array(spectral.frame.mean)
    sampler = RGBAdaptiveSampler2D(rgb, ratio=2, fraction=0.3, min_samples=5, cutoff=0.05)
    pyrandom.seed(72); out["tasks1"] = np.array(sampler.generate_tasks((20, 16)))
    cam.frame_sampler = sampler
    pyrandom.seed(73); rsrandom.seed(73); cam.observe()
    out["xyz_mean2"], out["xyz_var2"], out["xyz_n2"] = np.array(f.mean), np.array(f.variance), np.array(f.samples)
    pyrandom.seed(74); out["tasks2"] = np.array(sampler.generate_tasks((20, 16)))
    save("f16_rgb", **out)


def f17_scenes():
    """Whole demo scenes as the reference renders them (SerialEngine): the dispersive prism (demos/prism.py variant: nested CSG, two
    Sellmeier glasses, importance 9 on the prism, path weight 0.75, four one-bin spectral slices) and the Cornell box with glass
    (demos/cornell_box.py variant)."""
    out = {}
    world, prims = scenes.build_prism(NS)
    cam, pipe = scenes.prism_camera(NS, world, (24, 18), 2, 4, 4)
    out["prism_mean"], out["prism_var"], out["prism_n"] = observe_frame(cam, pipe, 81)
    world, prims = scenes.build_cornell(NS)
    cam, pipe = scenes.cornell_camera(NS, world, (20, 20), 3, 5)
    out["cornell_mean"], out["cornell_var"], out["cornell_n"] = observe_frame(cam, pipe, 82)
    save("f17_scenes", **out)


# ------------------------------------------------------------------ F18 the `flat` workload: ONE mesh of 1 048 576 triangles
def f18_flat():
    """SURVEY.md 8(d) "M1M-flat" (scenes.build_flat: the bench's `flat` workload, the one tree that leaves the caches): the reference's own
    SAH tree over 1 048 576 triangles as the SHA-256 of MeshData.save()'s bytes (kdtree3d.pyx:126-486, mesh.pyx:864-931), and one million
    Mesh.hit() answers as a digest of (triangle, t, u, v, w, exiting), the form of F04's."""
    import time
    v, t = scenes.displaced_sphere(512)
    t0 = time.time()
    mesh = Mesh(v, t, smoothing=False, closed=True)
    print("    reference built the 1M-triangle tree in %.1f s" % (time.time() - t0))
    b = rsm_bytes(mesh)
    out = {"flat_sha256": np.frombuffer(hashlib.sha256(b).digest(), dtype=np.uint8), "flat_len": np.array([len(b)]),
           "flat_ntri": np.array([mesh.data.face_normals.shape[0]]),
           "flat_face_normals_sha256": np.frombuffer(hashlib.sha256(mesh.data.face_normals.tobytes()).digest(), dtype=np.uint8)}
    o, d, m = raysets.random_outside(1000000, 51)
    tri, tt, uvw, ex, _ = mesh_hit_arrays(mesh, o, d, m)
    h = hashlib.sha256(); h.update(tri.tobytes()); h.update(tt.tobytes()); h.update(uvw.tobytes()); h.update(ex.tobytes())
    out["digest_1m"] = np.frombuffer(h.digest(), dtype=np.uint8)
    out["digest_1m_hits"] = np.array([(tri >= 0).sum()])
    # a few hundred answers in full, so that a mismatch can be located
    out["first_tri"] = tri[:512]; out["first_t"] = tt[:512]; out["first_uvw"] = uvw[:512]; out["first_ex"] = ex[:512]
    save("f18_flat", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["all"]
    run = lambda k: "all" in which or k in which  # noqa: E731
    if run("f00"): f00_math()
    if run("f01"): f01_mt()
    if run("f02"): f02_aabb()
    m70k = None
    if run("f03") or run("f04"):
        m70k = f03_kd() if run("f03") else Mesh(*scenes.displaced_sphere(132), smoothing=False)
    if run("f04"): f04_mesh(m70k)
    if run("f04b"): f04b_small_meshes()
    if run("f05"): f05_primitives()
    if run("f06"): f06_csg()
    if run("f07"): f07_world()
    if run("f08"): f08_camera()
    if run("f09"): f09_stats()
    if run("f10"): f10_frames()
    if run("f11"): f11_edges()
    if run("f12"): f12_volumes()
    if run("f13"): f13_lambert()
    if run("f14"): f14_glass()
    if run("f15"): f15_importance()
    if run("f16"): f16_rgb()
    if run("f17"): f17_scenes()
    if run("f18"): f18_flat()
