"""
Pins the CPU oracle (oracle/rsx_oracle.c) — and the product's host-side builders that feed it (math, KD build,
mesh preprocessing, scene flattening) — against golden vectors captured from the compiled reference
(tests/golden/make_golden.py). Bit-exact unless stated. CPU only.
"""
import hashlib

import numpy as np
import pytest

import raysets
from source_amd import scenes
from source_amd._flatten import FlatScene


def eq(a, b):
    return np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


# ---------------------------------------------------------------------------------------- F1 / F2
def test_mt19937_64_stream(orc, golden):
    g = golden("f01_mt")
    for s, ref in zip(g["seeds"], g["uniforms"]):
        assert eq(orc.mt_uniform(int(s), 1000), ref)
    # the reference's own known-answer vector (raysect/core/math/tests/test_random.py:38-39)
    assert orc.mt_uniform(1234567890, 1)[0] == 0.8114659955555504


def test_aabb_slabs(orc, golden):
    g = golden("f02_aabb")
    assert eq(orc.aabb_intersect(g["lower"], g["upper"], g["origin"], g["direction"]), g["result"])


# ---------------------------------------------------------------------------------------- F3 KD build
def _rsm_kd_blob(rsm):
    """Split an RSM v1.0 byte string into (header fields, kd blob) — SURVEY.md Appendix A."""
    b = bytes(rsm)
    assert b[:3] == b"RSM" and b[3] == 1 and b[4] == 0
    nv, nn, nt = np.frombuffer(b[8:20], dtype="<i4")
    stride = 6 if nn else 3
    off = 20 + 12 * nv + 12 * nn + 4 * stride * nt
    return (nv, nn, nt), b[off:]


@pytest.mark.parametrize("name", ["cube", "sphere8", "blob24", "fan500"])
def test_kd_build_small_meshes(orc, ns, golden, name):
    g = golden("f03_kd")
    v, t = {"cube": scenes.cube_mesh, "sphere8": lambda: scenes.displaced_sphere(8, radius=1.0),
            "blob24": lambda: scenes.displaced_sphere(24, radius=0.5), "fan500": lambda: scenes.fan_mesh(500)}[name]()
    (nv, nn, nt), ref_blob = _rsm_kd_blob(g[name])
    # oracle
    n, tris, fn, boxes = orc.mesh_prepare(v, t)
    assert n == nt
    assert eq(fn, g[name + "_face_normals"])
    assert orc.kd_build(boxes, 0, 1, 5.0, 0.25)[5] == ref_blob
    # product host builders (librsx C++), full RSM byte equality
    mesh = ns.Mesh(v, t, smoothing=False, closed=(name != "fan500"))
    assert eq(mesh.data.face_normals, g[name + "_face_normals"])
    import io
    f = io.BytesIO()
    mesh.save(f)
    assert f.getvalue() == bytes(g[name])


def test_kd_build_nondefault_params(orc, ns, golden):
    g = golden("f03_kd")
    v, t = scenes.displaced_sphere(24, radius=0.5)
    _, ref_blob = _rsm_kd_blob(g["blob24_params"])
    n, tris, fn, boxes = orc.mesh_prepare(v, t)
    assert orc.kd_build(boxes, 9, 4, 20.0, 0.1)[5] == ref_blob
    mesh = ns.Mesh(v, t, smoothing=False, kdtree_max_depth=9, kdtree_min_items=4, kdtree_hit_cost=20.0, kdtree_empty_bonus=0.1)
    assert mesh.data.kd.serialise() == ref_blob


def test_kd_build_70k_digest(orc, golden, m70k):
    g = golden("f03_kd")
    mesh, v, t = m70k
    import io
    f = io.BytesIO()
    mesh.save(f)
    b = f.getvalue()
    assert len(b) == int(g["m70k_len"][0])
    assert mesh.data._triangles.shape[0] == int(g["m70k_ntri"][0])
    assert hashlib.sha256(b).digest() == bytes(g["m70k_sha256"])
    assert hashlib.sha256(mesh.data.face_normals.tobytes()).digest() == bytes(g["m70k_face_normals_sha256"])
    # oracle's own builder produces the same tree
    n, tris, fn, boxes = orc.mesh_prepare(v, t)
    assert orc.kd_build(boxes, 0, 1, 5.0, 0.25)[5] == mesh.data.kd.serialise()


@pytest.mark.parametrize("name,builder", [("mixed", scenes.build_mixed), ("csg", scenes.build_csg_demo)])
def test_world_tree_and_boxes(orc, ns, golden, name, builder):
    g = golden("f03_kd")
    world = builder(ns)[0]
    flat = world.flatten()
    assert eq(flat.boxes, g["world_" + name + "_boxes"])        # every primitive.bounding_box(), bit-exact
    assert flat.world_kd.serialise() == bytes(g["world_" + name])
    assert orc.kd_build(flat.boxes, 0, 1, 80.0, 0.2)[5] == bytes(g["world_" + name])


# ---------------------------------------------------------------------------------------- F4 mesh
def _check_mesh(r, g, key):
    tri = np.where(r["prim"] >= 0, r["tri"], -1)
    assert eq(tri, g[key + "_tri"]), key
    hit = tri >= 0
    assert eq(r["t"][hit], g[key + "_t"][hit]), key
    assert eq(r["uvw"][hit], g[key + "_uvw"][hit]), key
    if key + "_ex" in g:
        assert eq(r["exiting"][hit], g[key + "_ex"][hit]), key
    return hit


def test_mesh_hits(orc, golden, m70k):
    g = golden("f04_mesh")
    mesh, v, t = m70k
    flat = FlatScene([mesh])
    sets = {"grid": raysets.pinhole_grid(96), "outside": raysets.random_outside(6000, 41),
            "outside_raw": raysets.random_outside(2000, 42, unit=False), "interior": raysets.random_interior(4000, 43),
            "vertices": raysets.through_vertices(v, 4000, 44), "edges": raysets.along_edges(v, t, 3000, 45),
            "axis": raysets.axis_aligned(3000, 46, 0.1, v)}
    for name, (o, d, m) in sets.items():
        r = orc.prim_hit_batch(flat, 0, o, d, m, geometry=True)
        hit = _check_mesh(r, g, name)
        if name + "_extra" in g:
            assert eq(r["geom"][hit], g[name + "_extra"][hit]), name
    # origin on the surface / max_distance just short, exact and long
    r = orc.prim_hit_batch(flat, 0, g["surf_o"], g["surf_d"])
    _check_mesh(r, g, "surf")
    o, d, _ = sets["outside"]
    r = orc.prim_hit_batch(flat, 0, o[g["maxd_idx"]], d[g["maxd_idx"]], g["maxd_m"])
    _check_mesh(r, g, "maxd")
    # contains()
    assert eq(orc.prim_contains_batch(flat, 0, raysets.points(4000, 49, 0.1)), g["contains"])


def test_mesh_next_intersection_sequences(orc, golden, m70k):
    g = golden("f04_mesh")
    flat = FlatScene([m70k[0]])
    o, d, m = raysets.random_outside(1500, 47)
    counts, t, ex = orc.roots_batch(flat, 0, o, d, m, max_roots=64)
    assert eq(counts, g["seq_counts"])
    mask = np.arange(64)[None, :] < counts[:, None]
    assert eq(t[mask], g["seq_t"])
    assert eq(ex[mask], g["seq_ex"])


def test_mesh_smoothing_instance_transform(orc, ns, golden, m70k):
    g = golden("f04_mesh")
    _, v, t = m70k
    vn = scenes.vertex_normals(v, t)
    sm = ns.Mesh(v, np.concatenate([t, t], axis=1), vn, smoothing=True, transform=ns.translate(0.01, -0.02, 0.03) * ns.rotate(33, 21, -14))
    assert eq(np.array(sm.to_local().m).reshape(4, 4), g["smooth_to_local"])
    assert eq(np.array(sm.to_root().m).reshape(4, 4), g["smooth_to_root"])
    flat = FlatScene([sm])
    o, d, m = raysets.random_outside(3000, 48)
    r = orc.prim_hit_batch(flat, 0, o, d, m, geometry=True)
    hit = _check_mesh(r, g, "smooth")
    assert eq(r["geom"][hit], g["smooth_extra"][hit])


def test_mesh_1m_ray_digest(orc, golden, m70k):
    g = golden("f04_mesh")
    flat = FlatScene([m70k[0]])
    o, d, m = raysets.random_outside(1000000, 50)
    r = orc.prim_hit_batch(flat, 0, o, d, m, threads=orc.max_threads())
    tri = np.where(r["prim"] >= 0, r["tri"], -1).astype(np.int32)
    t = np.where(tri >= 0, r["t"], np.nan)
    h = hashlib.sha256()
    for a in (tri, t, r["uvw"], r["exiting"]):
        h.update(np.ascontiguousarray(a).tobytes())
    assert int((tri >= 0).sum()) == int(g["digest_1m_hits"][0])
    assert h.digest() == bytes(g["digest_1m"])


def test_flat_1m_triangle_tree_and_digest(orc, golden, m1m):
    """F18, the bench's `flat` workload (ONE mesh of 1 048 576 triangles, SURVEY.md 8d "M1M-flat"): librsx's host builder writes the
    reference's own 66.7 MB of MeshData.save() — vertices, triangles and the SAH tree of kdtree3d.pyx:126-486, node for node — and the
    pinned oracle answers one million rays with the reference's (triangle, t, u, v, w, exiting)."""
    import io
    g = golden("f18_flat")
    mesh, v, t = m1m
    f = io.BytesIO()
    mesh.save(f)
    b = f.getvalue()
    assert len(b) == int(g["flat_len"][0]) and mesh.data._triangles.shape[0] == int(g["flat_ntri"][0]) == 1047536    # (1 048 576 less the degenerate ones at the poles, _filter_triangles mesh.pyx:363-399)
    assert hashlib.sha256(b).digest() == bytes(g["flat_sha256"])
    assert hashlib.sha256(mesh.data.face_normals.tobytes()).digest() == bytes(g["flat_face_normals_sha256"])
    flat = FlatScene([mesh])
    o, d, m = raysets.random_outside(1000000, 51)
    r = orc.prim_hit_batch(flat, 0, o, d, m, threads=orc.max_threads())
    tri = np.where(r["prim"] >= 0, r["tri"], -1).astype(np.int32)
    tt = np.where(tri >= 0, r["t"], np.nan)
    assert eq(tri[:512], g["first_tri"]) and eq(tt[:512], g["first_t"]) and eq(r["uvw"][:512], g["first_uvw"]) and eq(r["exiting"][:512], g["first_ex"])
    h = hashlib.sha256()
    for a in (tri, tt, r["uvw"], r["exiting"]):
        h.update(np.ascontiguousarray(a).tobytes())
    assert int((tri >= 0).sum()) == int(g["digest_1m_hits"][0])
    assert h.digest() == bytes(g["digest_1m"])


@pytest.mark.parametrize("name", ["cube", "sphere8", "blob24", "fan500"])
def test_small_mesh_hits(orc, ns, golden, name):
    g = golden("f04b_small_meshes")
    v, t = {"cube": scenes.cube_mesh, "sphere8": lambda: scenes.displaced_sphere(8, radius=1.0),
            "blob24": lambda: scenes.displaced_sphere(24, radius=0.5), "fan500": lambda: scenes.fan_mesh(500)}[name]()
    mesh = ns.Mesh(v, t, smoothing=False, closed=(name != "fan500"))
    r = orc.prim_hit_batch(FlatScene([mesh]), 0, g[name + "_o"], g[name + "_d"], None)
    _check_mesh(r, g, name)


# ---------------------------------------------------------------------------------------- F5 analytic
def _prims(ns):
    tr = ns.translate(0.1, -0.2, 0.3) * ns.rotate(25, -35, 45)
    P = ns.Point3D
    return {"sphere": ns.Sphere(0.8, transform=tr), "sphere_id": ns.Sphere(1.0),
            "box": ns.Box(P(-0.5, -0.7, -0.4), P(0.6, 0.5, 0.9), transform=tr), "box_id": ns.Box(P(-0.6, -0.6, -0.6), P(0.6, 0.6, 0.6)),
            "cylinder": ns.Cylinder(0.5, 1.2, transform=tr), "cylinder_id": ns.Cylinder(0.6, 1.0, transform=ns.translate(0, 0, -0.5))}


def test_analytic_primitives(orc, ns, golden):
    g = golden("f05_primitives")
    for k, (name, prim) in enumerate(_prims(ns).items()):
        assert eq(np.array(prim.to_local().m).reshape(4, 4), g[name + "_to_local"]), name
        assert eq(prim.bounding_box().as_list(), g[name + "_bbox"]), name
        flat = FlatScene([prim])
        o, d, m = raysets.primitive_rays(3000, 70 + k)
        counts, t, ex, geom = orc.roots_batch(flat, 0, o, d, m, max_roots=2, geometry=True)
        ref = g[name]                                       # [n, 2, 14]: t, exiting, hit, inside, outside, normal
        valid = ~np.isnan(ref[:, :, 0])
        assert eq(counts, valid.sum(axis=1)), name
        assert eq(t[valid], ref[:, :, 0][valid]), name
        assert eq(ex[valid], ref[:, :, 1][valid]), name
        assert eq(geom[valid], ref[:, :, 2:][valid]), name
        assert eq(orc.prim_contains_batch(flat, 0, raysets.points(2000, 90 + k, 1.2)), g[name + "_contains"]), name


# ---------------------------------------------------------------------------------------- F6 / F7 world level
def _check_world(r, idx, rec):
    assert eq(r["prim"], idx)
    hit = idx >= 0
    assert eq(r["t"][hit], rec[hit, 0])
    assert eq(r["exiting"][hit], rec[hit, 1])
    assert eq(r["geom"][hit], rec[hit, 2:])


def test_csg_world(orc, ns, golden):
    g = golden("f06_csg")
    world, prims = scenes.build_csg_demo(ns)
    flat = world.flatten()
    o, d, m = raysets.scene_rays(12000, 101, 9.0, 4.5)
    og, dg, mg = raysets.pinhole_grid(64, (0.0, 0.0, -4.0), 75.0)
    o, d, m = np.concatenate([o, og]), np.concatenate([d, dg]), np.concatenate([m, mg])
    _check_world(orc.hit_batch(flat, o, d, m, geometry=True), g["world_idx"], g["world_rec"])
    for name, index in (("obj0", 0), ("lens", 4)):
        counts, t, ex = orc.roots_batch(flat, index, o[:4000], d[:4000], None, max_roots=64)
        assert eq(counts, g[name + "_counts"]), name
        mask = np.arange(64)[None, :] < counts[:, None]
        assert eq(t[mask], g[name + "_t"]), name
        assert eq(ex[mask], g[name + "_ex"]), name
    assert eq(orc.contains_batch(flat, raysets.points(4000, 102, 4.5)), g["contains"])


def test_mixed_world(orc, ns, golden):
    g = golden("f07_world")
    world, prims = scenes.build_mixed(ns)
    flat = world.flatten()
    assert eq(np.array([np.array(p.to_local().m).reshape(4, 4) for p in prims]), g["to_local"])
    o, d, m = raysets.scene_rays(20000, 111, 6.0, 2.2)
    r = orc.hit_batch(flat, o, d, m, geometry=True)
    _check_world(r, g["idx"], g["rec"])
    assert not np.isin(r["prim"], [2, 3]).any()              # coincident spheres: the last registered wins (App. B19)
    assert eq(orc.contains_batch(flat, raysets.points(6000, 112, 2.0)), g["contains"])


# ---------------------------------------------------------------------------------------- F8 camera / F9 stats / F10 frames
def test_pinhole_rays(orc, ns, golden):
    from source_amd import _lib
    g = golden("f08_camera")
    world = ns.World()
    cam = ns.PinholeCamera((48, 32), fov=52.0, parent=world, transform=ns.translate(0.3, -0.2, 1.0) * ns.rotate(20, 10, 5))
    assert eq(np.array(cam.to_root().m).reshape(4, 4), g["to_root"])
    desc = _lib.RenderDesc()
    desc.camera = cam.device_camera()
    for i, v in enumerate(ns.AffineMatrix3D().m):            # _generate_rays returns camera-local rays (pinhole.pyx:169-204);
        desc.camera.to_root[i] = v                            # the to_root() transform happens later (observer.pyx:403-404)
    tasks = np.array([(0, 0), (47, 31), (13, 7), (24, 16), (5, 30)], dtype=np.int32)
    u = np.ascontiguousarray(g["uniforms"])
    desc.tasks, desc.n_tasks, desc.spp, desc.uniforms = _lib.ptr(tasks), 5, 16, _lib.ptr(u)
    rays = orc.pinhole_rays(desc)
    assert eq(rays, g["rows"][:, 2:])


def test_welford_and_combine(orc, golden):
    g = golden("f09_stats")
    for x, states in zip(g["x"], g["states"]):
        assert eq(orc.add_samples(x), states)
    m, v, n = orc.frame_combine(g["ma"], g["va"], g["na"], g["mb"], g["vb"], g["nb"])
    assert eq(np.stack([m, v, n.astype(float)], axis=1), g["comb"])


def _observe_oracle(orc, ns, cam, pipe, seed, frame=None):
    """Runs the oracle over the tasks / uniforms the reference's SerialEngine would consume (stream parity)."""
    import random as pyrandom
    from source_amd.core import random as rsrandom
    from source_amd.optical.observer import HipEngine
    pyrandom.seed(seed)
    rsrandom.seed(seed)
    world = cam.root
    flat = world.flatten()
    slices = cam._slice_spectrum()
    tasks = cam._generate_tasks()
    nx, ny = cam.pixels
    bins = cam.spectral_bins
    if frame is None:
        frame = [np.zeros((nx, ny, bins)), np.zeros((nx, ny, bins)), np.zeros((nx, ny, bins), dtype=np.int32)]
    t = np.array(tasks)
    eng = HipEngine(rng="stream")
    for sl in slices:
        keep = []
        desc = cam.render_desc(world, tasks, sl, eng, keep)
        desc.power = 1 if pipe.power else 0
        mean, var, rays = orc.render_pinhole(flat, desc)
        assert rays >= len(tasks) * cam.pixel_samples          # + one per daughter ray (null surfaces, scattering)
        z = slice(sl.offset, sl.offset + sl.bins)
        sub = [f[t[:, 0], t[:, 1], z] for f in frame]
        m, v, n = orc.frame_combine(sub[0], sub[1], sub[2], mean, np.maximum(var, 0), np.full(mean.shape, cam.pixel_samples, dtype=np.int32))
        frame[0][t[:, 0], t[:, 1], z], frame[1][t[:, 0], t[:, 1], z], frame[2][t[:, 0], t[:, 1], z] = m, v, n
    return frame


def test_frames_c2(orc, ns, golden):
    g = golden("f10_frames")
    world, mesh, box = scenes.build_c2(ns, n=132)
    cam, pipe = scenes.c2_camera(ns, world, (40, 40), spp=4, bins=15)
    f = _observe_oracle(orc, ns, cam, pipe, 1)
    assert eq(f[0], g["c2_mean"]) and eq(f[1], g["c2_var"]) and eq(f[2], g["c2_n"])
    f = _observe_oracle(orc, ns, cam, pipe, 2, f)            # accumulate pass: combine_samples with n > 1 on both sides
    assert eq(f[0], g["c2_mean2"]) and eq(f[1], g["c2_var2"]) and eq(f[2], g["c2_n2"])


def test_frames_sliced_power_smoothing(orc, ns, golden):
    g = golden("f10_frames")
    world, mesh, box = scenes.build_c2(ns, n=48, smoothing=True, with_normals=True)
    pipe = ns.SpectralPowerPipeline2D()
    cam = ns.PinholeCamera((24, 36), fov=45, sensitivity=2.5, parent=world, pipelines=[pipe], frame_sampler=ns.FullFrameSampler2D(),
                           transform=ns.translate(0, 0.16, -0.4) * ns.rotate(0, -12, 0))
    cam.pixel_samples, cam.spectral_bins, cam.spectral_rays, cam.quiet = 1, 7, 3, True
    cam.min_wavelength, cam.max_wavelength = 400.0, 700.0
    f = _observe_oracle(orc, ns, cam, pipe, 3)
    assert eq(f[0], g["c2s_mean"]) and eq(f[1], g["c2s_var"]) and eq(f[2], g["c2s_n"])


def test_frames_csg_and_instanced(orc, ns, golden):
    g = golden("f10_frames")
    world, prims = scenes.build_csg_demo(ns)
    cam, pipe = scenes.csg_camera(ns, world, (32, 32), spp=6, bins=5)
    f = _observe_oracle(orc, ns, cam, pipe, 4)
    assert eq(f[0], g["csg_mean"]) and eq(f[1], g["csg_var"]) and eq(f[2], g["csg_n"])
    world = scenes.build_c3(ns, n=32)[0]
    cam, pipe = scenes.c3_camera(ns, world, (32, 32), spp=3, bins=4)
    f = _observe_oracle(orc, ns, cam, pipe, 5)
    assert eq(f[0], g["c3_mean"]) and eq(f[1], g["c3_var"]) and eq(f[2], g["c3_n"])


def test_spectral_function_sampling(ns, golden):
    g = golden("f10_frames")
    sf = ns.InterpolatedSF([300, 490, 510, 590, 610, 800], np.array([0.0, 0.1, 1.0, 0.7, 0.2, 0.4]))
    assert eq(sf.sample(375.0, 740.0, 15), g["sf_interp_15"])
    assert eq(sf.sample(480.0, 520.0, 3), g["sf_interp_3"])
    assert eq(sf.sample(200.0, 900.0, 9), g["sf_interp_wide"])
    assert eq(ns.ConstantSF(0.75).sample(375.0, 740.0, 4), g["sf_const"])


# ---------------------------------------------------------------------------------------- F11 edge semantics
EDGE_FRAMES = (("one", (1, 2), 3, 2, None), ("odd", (3, 5), 2, 1, None),
               ("mask", (9, 7), 2, 3, (np.add.outer(np.arange(9), np.arange(7)) % 3 != 0)))


def test_edge_semantics(orc, ns, golden):
    """Empty world (App. B.16), coincident primitives (B.19), t == max_distance on analytic vs mesh surfaces (B.18), origins on
    surfaces, axis-parallel rays along faces / edges / a cylinder axis; 1x2, 3x5x1-bin and masked (ragged) frames."""
    g = golden("f11_edges")
    for name, (world, prims) in scenes.build_edge_worlds(ns).items():
        flat = world.flatten()
        o, d, m = scenes.edge_rays(name)
        _check_world(orc.hit_batch(flat, o, d, m, geometry=True), g[name + "_idx"], g[name + "_rec"])
        pts = np.concatenate([o, o + 0.25 * d])
        assert eq(orc.contains_batch(flat, pts), g[name + "_contains"]), name
    assert (g["empty_idx"] == -1).all() and (g["coincident_idx"][[0, 3]] == 3).all() and g["coincident_idx"][-1] == 2
    assert g["limits_idx"][0] == 0 and g["limits_idx"][1] == -1 and g["limits_idx"][7] == -1 and g["limits_idx"][8] == 1   # B.18
    world, mesh, box = scenes.build_c2(ns, n=24)
    for tag, pixels, spp, bins, mask in EDGE_FRAMES:
        cam, pipe = scenes.edge_camera(ns, world, pixels, spp, bins, mask)
        f = _observe_oracle(orc, ns, cam, pipe, 21)
        assert eq(f[0], g[tag + "_mean"]) and eq(f[1], g[tag + "_var"]) and eq(f[2], g[tag + "_n"]), tag
    with pytest.raises(RuntimeError):
        ns.PinholeCamera((1, 1), parent=world)                 # pinhole.pyx:166-167


# ---------------------------------------------------------------------------------------- F12 transparent boundaries / volume emission
def test_frames_volume_emitters(orc, ns, golden):
    """NullMaterial + UniformVolumeEmitter (first slice of SURVEY.md §8f row 1): null surfaces continue the ray deterministically,
    each path segment integrates the emission of the volumes containing its origin in world.contains() order."""
    g = golden("f12_volumes")
    world, prims = scenes.build_volumes(ns)
    cam, pipe = scenes.volumes_camera(ns, world)
    f = _observe_oracle(orc, ns, cam, pipe, 31)
    assert eq(f[0], g["mean"]) and eq(f[1], g["var"]) and eq(f[2], g["n"])
    f = _observe_oracle(orc, ns, cam, pipe, 32, f)
    assert eq(f[0], g["mean2"]) and eq(f[1], g["var2"]) and eq(f[2], g["n2"])
    world, prims = scenes.build_volumes(ns, enclosed=False)      # most paths end in a miss: that segment contributes nothing
    cam, pipe = scenes.volumes_camera(ns, world)
    f = _observe_oracle(orc, ns, cam, pipe, 33)
    assert eq(f[0], g["open_mean"]) and eq(f[1], g["open_var"]) and eq(f[2], g["open_n"])


# ---------------------------------------------------------------------------------------- F13 Lambert: stochastic secondary rays
def _observe_oracle_mt(orc, ns, cam, pipe, seed, frame=None):
    """The reference's SerialEngine on a scene with scattering materials: jitter, hemisphere samples and roulette draws interleave
    in ONE MT19937-64 stream, so the oracle replays the stream itself (orc_render_pinhole_mt) instead of taking a jitter table."""
    import random as pyrandom
    from source_amd.optical.observer import HipEngine
    pyrandom.seed(seed)
    state = orc.mt_state(seed)
    world = cam.root
    flat = world.flatten()
    tasks = cam._generate_tasks()
    nx, ny = cam.pixels
    bins = cam.spectral_bins
    if frame is None:
        frame = [np.zeros((nx, ny, bins)), np.zeros((nx, ny, bins)), np.zeros((nx, ny, bins), dtype=np.int32)]
    t = np.array(tasks)
    eng = HipEngine(rng="philox")
    for sl in cam._slice_spectrum():
        keep = []
        desc = cam.render_desc(world, tasks, sl, eng, keep)
        mean, var, rays = orc.render_pinhole_mt(flat, desc, state)
        z = slice(sl.offset, sl.offset + sl.bins)
        sub = [f[t[:, 0], t[:, 1], z] for f in frame]
        m, v, n = orc.frame_combine(sub[0], sub[1], sub[2], mean, np.maximum(var, 0), np.full(mean.shape, cam.pixel_samples, dtype=np.int32))
        frame[0][t[:, 0], t[:, 1], z], frame[1][t[:, 0], t[:, 1], z], frame[2][t[:, 0], t[:, 1], z] = m, v, n
    return frame


def test_frames_lambert(orc, ns, golden):
    """Lambert + Russian roulette + depth limit, mixed with null surfaces and volume emission, bit for bit against the reference's
    SerialEngine frames (importance sampling off)."""
    g = golden("f13_lambert")
    world, prims = scenes.build_lambert(ns)
    cam, pipe = scenes.lambert_camera(ns, world)
    f = _observe_oracle_mt(orc, ns, cam, pipe, 41)
    assert eq(f[0], g["mean"]) and eq(f[1], g["var"]) and eq(f[2], g["n"])
    f = _observe_oracle_mt(orc, ns, cam, pipe, 42, f)
    assert eq(f[0], g["mean2"]) and eq(f[1], g["var2"]) and eq(f[2], g["n2"])
    world, prims = scenes.build_lambert(ns, with_volume=False)
    cam, pipe = scenes.lambert_camera(ns, world, pixels=(12, 10), spp=3, bins=3, extinction=(0.01, 3, 500))
    f = _observe_oracle_mt(orc, ns, cam, pipe, 43)
    assert eq(f[0], g["deep_mean"]) and eq(f[1], g["deep_var"]) and eq(f[2], g["deep_n"])
    cam, pipe = scenes.lambert_camera(ns, world, pixels=(12, 10), spp=3, bins=4, extinction=(0.3, 1, 3))
    cam.spectral_rays = 2
    f = _observe_oracle_mt(orc, ns, cam, pipe, 44)
    assert eq(f[0], g["rr0_mean"]) and eq(f[1], g["rr0_var"]) and eq(f[2], g["rr0_n"])


# ---------------------------------------------------------------------------------------- F14 Dielectric
def test_frames_dielectric(orc, ns, golden):
    """Dielectric: refraction / reflection choice, total internal reflection, transmission_only, per-slice Sellmeier index and the
    Beer-Lambert volume pass, bit for bit against the reference's SerialEngine frames (the oracle calls the same libm pow)."""
    g = golden("f14_glass")
    bk7 = ns.Sellmeier(1.03961212, 0.231792344, 1.01046945, 6.00069867e-3, 2.00179144e-2, 1.03560653e2)
    assert eq(np.array([bk7.average(375.0, 740.0), bk7.average(375.0, 496.0), bk7.average(700.0, 703.5), bk7.evaluate(589.3)]), g["sellmeier_avg"])
    assert eq(bk7.sample(375.0, 740.0, 7), g["sellmeier_sample"])
    world, prims = scenes.build_glass(ns)
    cam, pipe = scenes.glass_camera(ns, world)
    f = _observe_oracle_mt(orc, ns, cam, pipe, 51)
    assert eq(f[0], g["mean"]) and eq(f[1], g["var"]) and eq(f[2], g["n"])
    f = _observe_oracle_mt(orc, ns, cam, pipe, 52, f)
    assert eq(f[0], g["mean2"]) and eq(f[1], g["var2"]) and eq(f[2], g["n2"])
    world, prims = scenes.build_glass(ns, unit_transmission=True)
    cam, pipe = scenes.glass_camera(ns, world, pixels=(16, 12), spp=3, bins=4, spectral_rays=2, extinction=(0.01, 3, 500))
    f = _observe_oracle_mt(orc, ns, cam, pipe, 53)
    assert eq(f[0], g["clear_mean"]) and eq(f[1], g["clear_var"]) and eq(f[2], g["clear_n"])


# ---------------------------------------------------------------------------------------- F15 multiple importance sampling
def _importance_scene(ns):
    world, prims = scenes.build_lambert(ns)
    prims[5].material.importance = 3.0
    ns.Sphere(0.12, world, ns.translate(-0.6, 0.5, 1.2), ns.UniformSurfaceEmitter(ns.ConstantSF(1.0), 4.0))
    cam, pipe = scenes.lambert_camera(ns, world)
    cam.ray_importance_sampling = True
    cam.ray_important_path_weight = 0.25
    return world, cam, pipe


def test_frames_importance_sampling(orc, ns, golden):
    """ContinuousBSDF multiple importance sampling against the reference's SerialEngine frames: bounding spheres and selection CDF
    of the important primitives (host), sphere selection, cone / full-sphere sampling, the combined pdf."""
    g = golden("f15_importance")
    world, cam, pipe = _importance_scene(ns)
    f = _observe_oracle_mt(orc, ns, cam, pipe, 61)
    assert eq(f[0], g["mean"]) and eq(f[1], g["var"]) and eq(f[2], g["n"])
    f = _observe_oracle_mt(orc, ns, cam, pipe, 62, f)
    assert eq(f[0], g["mean2"]) and eq(f[1], g["var2"]) and eq(f[2], g["n2"])
    cam.ray_important_path_weight = 0.9
    cam.ray_extinction_prob, cam.ray_extinction_min_depth, cam.ray_max_depth = 0.01, 3, 500
    f = _observe_oracle_mt(orc, ns, cam, pipe, 63)
    assert eq(f[0], g["heavy_mean"]) and eq(f[1], g["heavy_var"]) and eq(f[2], g["heavy_n"])
    # ImportanceManager.sample / pdf on their own: selection by CDF, cone and full-sphere sampling, the cimported rotate_basis
    import ctypes as C
    from source_amd.optical.observer import HipEngine
    keep = []
    desc = cam.render_desc(world, [(0, 0)], cam._slice_spectrum()[0], HipEngine(), keep)
    L = orc.lib()
    L.orc_important_pdf.restype = C.c_double
    L.orc_important_pdf.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_important_sample.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_int, C.c_void_p]
    u = orc.mt_uniform(99, 3 * 40)
    for i, (p, d) in enumerate(zip(g["im_pts"], g["im_dirs"])):
        p, d = np.ascontiguousarray(p), np.ascontiguousarray(d)
        assert L.orc_important_pdf(C.byref(desc), orc.p(p), orc.p(d)) == g["im_pdf"][i]
        out = np.zeros(3)
        L.orc_important_sample(C.byref(desc), orc.p(p), u[3 * i], u[3 * i + 1], u[3 * i + 2], 1, orc.p(out))
        assert eq(out, g["im_samples"][i])
        assert L.orc_important_pdf(C.byref(desc), orc.p(p), orc.p(out)) == g["im_pdf2"][i]


# ---------------------------------------------------------------------------------------- F16 RGB pipeline / adaptive sampler
def _rgb_camera(ns, world, rgb):
    cam = ns.PinholeCamera((20, 16), fov=45, sensitivity=2.5, parent=world, pipelines=[rgb], frame_sampler=ns.FullFrameSampler2D(),
                           transform=ns.translate(0, 0.16, -0.4) * ns.rotate(0, -12, 0))
    cam.pixel_samples, cam.spectral_bins, cam.spectral_rays, cam.quiet = 3, 9, 3, True
    cam.min_wavelength, cam.max_wavelength = 400.0, 700.0
    return cam


def _observe_oracle_rgb(orc, ns, cam, rgb, seed):
    """One observe() of an RGBPipeline2D with the oracle standing in for the device: the host half (working frames, finalise,
    adaptive sampler) is the product's own code."""
    import random as pyrandom
    from source_amd.core import random as rsrandom
    from source_amd.optical.observer import HipEngine
    pyrandom.seed(seed)
    rsrandom.seed(seed)
    world = cam.root
    flat = world.flatten()
    slices = cam._slice_spectrum()
    cam._initialise_pipelines(cam.min_wavelength, cam.max_wavelength, cam.spectral_bins, slices, True)
    tasks = cam._generate_tasks()
    t = np.array(tasks)
    eng = HipEngine(rng="stream")
    for slice_id, sl in enumerate(slices):
        keep = []
        desc = cam.render_desc(world, tasks, sl, eng, keep)
        mean, var, _ = orc.render_pinhole_xyz(flat, desc, rgb._resampled[slice_id], rgb._deltas[slice_id])
        rgb.update_block(t[:, 0], t[:, 1], mean, var)
    cam._finalise_pipelines()
    return tasks


def test_rgb_pipeline_and_adaptive_sampler(orc, ns, golden):
    """RGBPipeline2D / XYZPixelProcessor and RGBAdaptiveSampler2D against the reference: resampled CIE curves,
    sRGB conversion, XYZ frames over three spectral slices and two accumulating passes (the second over the adaptive sampler's
    tasks), and the sampler's task lists."""
    import random as pyrandom
    from source_amd.optical import colour
    g = golden("f16_rgb")
    assert eq(colour.resample_ciexyz(375.0, 740.0, 7), g["xyz_7"]) and eq(colour.resample_ciexyz(496.0, 618.0, 5), g["xyz_slice"])
    srgb = np.array([colour.ciexyz_to_srgb(*v) for v in ((0.2, 0.3, 0.1), (0.001, 0.002, 0.0005), (0.9, 1.0, 1.2), (0.0, 0.0, 0.0))])
    assert np.allclose(srgb, g["srgb"], rtol=1e-15, atol=0)                    # one libm pow per channel
    world, mesh, box = scenes.build_c2(ns, n=48, smoothing=True, with_normals=True)
    rgb = ns.RGBPipeline2D()
    cam = _rgb_camera(ns, world, rgb)
    _observe_oracle_rgb(orc, ns, cam, rgb, 71)
    f = rgb.xyz_frame
    assert eq(f.mean, g["xyz_mean"]) and eq(f.variance, g["xyz_var"]) and eq(f.samples, g["xyz_n"])
    sampler = ns.RGBAdaptiveSampler2D(rgb, ratio=2, fraction=0.3, min_samples=5, cutoff=0.05)
    pyrandom.seed(72)
    assert np.array_equal(np.array(sampler.generate_tasks((20, 16))), g["tasks1"])
    cam.frame_sampler = sampler
    tasks = _observe_oracle_rgb(orc, ns, cam, rgb, 73)
    assert len(tasks) == 320                                                   # 3 samples per pixel < min_samples: everything again
    assert eq(f.mean, g["xyz_mean2"]) and eq(f.variance, g["xyz_var2"]) and eq(f.samples, g["xyz_n2"])
    pyrandom.seed(74)
    assert np.array_equal(np.array(sampler.generate_tasks((20, 16))), g["tasks2"])
    with pytest.raises(TypeError):
        ns.RGBAdaptiveSampler2D(ns.SpectralRadiancePipeline2D())


# ---------------------------------------------------------------------------------------- F17 whole demo scenes
def test_frames_demo_scenes(orc, ns, golden):
    """The prism scene (BASELINE configs[4]: nested CSG + dispersive glass + importance sampling + spectral slices) and the Cornell box
    with glass (configs[0]) against the reference's SerialEngine frames."""
    g = golden("f17_scenes")
    world, prims = scenes.build_prism(ns)
    cam, pipe = scenes.prism_camera(ns, world, (24, 18), 2, 4, 4)
    f = _observe_oracle_mt(orc, ns, cam, pipe, 81)
    assert eq(f[0], g["prism_mean"]) and eq(f[1], g["prism_var"]) and eq(f[2], g["prism_n"])
    assert (f[0] > 0).mean() > 0.05
    world, prims = scenes.build_cornell(ns)
    cam, pipe = scenes.cornell_camera(ns, world, (20, 20), 3, 5)
    f = _observe_oracle_mt(orc, ns, cam, pipe, 82)
    assert eq(f[0], g["cornell_mean"]) and eq(f[1], g["cornell_var"]) and eq(f[2], g["cornell_n"])


def test_portable_pow(orc):
    """The portable pow the oracle (counter mode), the device and the host-callback path share: against libm's pow (what the reference
    calls, correctly rounded to well under one unit in the last place) it stays within 2 ulp over the attenuation range and the
    whole exponent range; the Python restatement gives the C one's bits."""
    import math
    from source_amd.optical import _portable as P
    rng = np.random.RandomState(0)
    x = np.concatenate([rng.uniform(0, 1, 30000) ** rng.choice([1, 3, 8], 30000), rng.uniform(0.5, 2.5, 5000), 2.0 ** rng.randint(-1000, 1000, 5000) * rng.uniform(1, 2, 5000)])
    y = np.concatenate([rng.uniform(0, 50, 30000) ** rng.choice([1, 2], 30000), rng.uniform(-300, 300, 5000), rng.uniform(-1, 1, 5000)])
    x = np.where(x > 0, x, 0.25)
    got = orc.portable_pow(x, y)
    with np.errstate(over="ignore", under="ignore"):
        want = np.power(x, y)
    ok = np.isfinite(want) & (want > 2.3e-308) & np.isfinite(got)
    ulp = np.abs(got[ok] - want[ok]) / np.spacing(want[ok])
    assert ok.sum() > 30000 and ulp.max() <= 2.0, ulp.max()
    assert (got[ok] == want[ok]).mean() > 0.85
    assert orc.portable_pow(np.array([1.0, 0.3, 0.0]), np.array([7.5, 0.0, 2.0])).tolist() == [1.0, 1.0, 0.0]
    for i in range(0, len(x), 17):
        assert P.pow(float(x[i]), float(y[i])) == got[i] or (math.isnan(got[i]))
    phi = rng.uniform(0, 2 * np.pi, 2000)
    sn, cs = orc.portable_sincos(phi)
    assert all(P.sincos(float(a)) == (float(s), float(c)) for a, s, c in zip(phi[:300], sn, cs))
    t = rng.uniform(0, 1, 300)
    assert all(P.asin(float(a)) == float(b) for a, b in zip(t, orc.portable_asin(t)))
