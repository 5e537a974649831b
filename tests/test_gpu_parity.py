"""
GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C-ABI (librsx via ctypes), against
 (1) the CPU oracle on identical flattened scenes and seeded ray sets, and (2) the committed golden vectors captured
from the compiled reference. Bar: bit-exact for primitive / triangle ids, distances, barycentrics, intersection
geometry and — in RSX_RNG_STREAM mode — the rendered frames (mean, variance, samples).
"""
import hashlib
import os

import numpy as np
import pytest

import raysets
from source_amd import scenes
from source_amd._flatten import FlatScene

from source_amd.optical.observer import FrameSampler2D, RectTasks

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class RectSampler(FrameSampler2D):
    """Frame sampler that keeps the natural (unshuffled) iy-outer / ix-inner order."""

    def generate_tasks(self, pixels):
        return [(ix, iy) for iy in range(pixels[1]) for ix in range(pixels[0])]


def eq(a, b):
    return np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


def dev_scene(flat):
    from source_amd.device import DeviceScene
    return DeviceScene(flat)


def assert_hits_equal(dev, ref, geometry=True):
    assert eq(dev["prim"], ref["prim"])
    hit = ref["prim"] >= 0
    assert eq(dev["t"][hit], ref["t"][hit])
    assert eq(dev["exiting"][hit], ref["exiting"][hit])
    assert eq(dev["tri"][hit], ref["tri"][hit])
    assert eq(dev["uvw"][hit], ref["uvw"][hit])
    if geometry:
        assert eq(dev["geom"][hit], ref["geom"][hit])


def test_mesh_world_vs_oracle_and_golden(orc, golden, m70k):
    g = golden("f04_mesh")
    mesh, v, t = m70k
    flat = FlatScene([mesh])
    sc = dev_scene(flat)
    sets = {"grid": raysets.pinhole_grid(96), "outside": raysets.random_outside(6000, 41),
            "outside_raw": raysets.random_outside(2000, 42, unit=False), "interior": raysets.random_interior(4000, 43),
            "vertices": raysets.through_vertices(v, 4000, 44), "edges": raysets.along_edges(v, t, 3000, 45),
            "axis": raysets.axis_aligned(3000, 46, 0.1, v)}
    for name, (o, d, m) in sets.items():
        dev = sc.hit_batch(o, d, m, geometry=True)
        assert_hits_equal(dev, orc.hit_batch(flat, o, d, m, geometry=True))
        # straight against the reference's Mesh.hit() vectors
        tri = np.where(dev["prim"] >= 0, dev["tri"], -1)
        assert eq(tri, g[name + "_tri"]), name
        hit = tri >= 0
        assert eq(dev["t"][hit], g[name + "_t"][hit]) and eq(dev["uvw"][hit], g[name + "_uvw"][hit]), name
        assert eq(dev["exiting"][hit], g[name + "_ex"][hit]), name
        if name + "_extra" in g:
            assert eq(dev["geom"][hit], g[name + "_extra"][hit]), name
    dev = sc.hit_batch(g["surf_o"], g["surf_d"])
    assert eq(np.where(dev["prim"] >= 0, dev["tri"], -1), g["surf_tri"])
    o, d, _ = sets["outside"]
    dev = sc.hit_batch(o[g["maxd_idx"]], d[g["maxd_idx"]], g["maxd_m"])
    assert eq(np.where(dev["prim"] >= 0, dev["tri"], -1), g["maxd_tri"])
    hit = dev["prim"] >= 0
    assert eq(dev["t"][hit], g["maxd_t"][hit])


def test_mesh_1m_ray_digest(golden, m70k):
    """One million rays against the reference's SHA-256 of (triangle, t, u, v, w, exiting)."""
    g = golden("f04_mesh")
    sc = dev_scene(FlatScene([m70k[0]]))
    o, d, m = raysets.random_outside(1000000, 50)
    r = sc.hit_batch(o, d, m)
    tri = np.where(r["prim"] >= 0, r["tri"], -1).astype(np.int32)
    t = np.where(tri >= 0, r["t"], np.nan)
    uvw = np.where((tri >= 0)[:, None], r["uvw"], 0).astype(np.float32)
    ex = np.where(tri >= 0, r["exiting"], 0).astype(np.uint8)
    h = hashlib.sha256()
    for a in (tri, t, uvw, ex):
        h.update(np.ascontiguousarray(a).tobytes())
    assert int((tri >= 0).sum()) == int(g["digest_1m_hits"][0])
    assert h.digest() == bytes(g["digest_1m"])


def test_mesh_next_intersection_and_contains(orc, golden, m70k):
    g = golden("f04_mesh")
    flat = FlatScene([m70k[0]])
    sc = dev_scene(flat)
    o, d, m = raysets.random_outside(1500, 47)
    counts, t, ex, geom, tri, uvw = sc.roots_batch(0, o, d, m, max_roots=64, geometry=True)
    assert eq(counts, g["seq_counts"])
    mask = np.arange(64)[None, :] < counts[:, None]
    assert eq(t[mask], g["seq_t"]) and eq(ex[mask], g["seq_ex"])
    # geometry of every root (hit / inside / outside points, normal) == the oracle's next_intersection() sequence
    oc, ot, oe, og = orc.roots_batch(flat, 0, o, d, m, max_roots=64, geometry=True)
    assert eq(geom[mask], og[mask]) and (tri[mask] >= 0).all() and (tri[~mask] == -1).all()
    first = counts > 0
    ref = orc.hit_batch(flat, o, d, m)                                   # first root == World.hit on the single-mesh scene
    assert eq(tri[first, 0], ref["tri"][first]) and eq(uvw[first, 0], ref["uvw"][first])
    assert eq(sc.contains_batch(raysets.points(4000, 49, 0.1))[:, 0], g["contains"])


def test_smoothed_transformed_mesh(orc, ns, golden, m70k):
    g = golden("f04_mesh")
    _, v, t = m70k
    vn = scenes.vertex_normals(v, t)
    sm = ns.Mesh(v, np.concatenate([t, t], axis=1), vn, smoothing=True, transform=ns.translate(0.01, -0.02, 0.03) * ns.rotate(33, 21, -14))
    flat = FlatScene([sm])
    o, d, m = raysets.random_outside(3000, 48)
    dev = dev_scene(flat).hit_batch(o, d, m, geometry=True)
    assert_hits_equal(dev, orc.hit_batch(flat, o, d, m, geometry=True))
    hit = dev["prim"] >= 0
    assert eq(np.where(hit, dev["tri"], -1), g["smooth_tri"])
    assert eq(dev["geom"][hit], g["smooth_extra"][hit])


@pytest.mark.parametrize("name", ["cube", "sphere8", "blob24", "fan500"])
def test_small_meshes(ns, golden, name):
    g = golden("f04b_small_meshes")
    v, t = {"cube": scenes.cube_mesh, "sphere8": lambda: scenes.displaced_sphere(8, radius=1.0),
            "blob24": lambda: scenes.displaced_sphere(24, radius=0.5), "fan500": lambda: scenes.fan_mesh(500)}[name]()
    mesh = ns.Mesh(v, t, smoothing=False, closed=(name != "fan500"))
    r = dev_scene(FlatScene([mesh])).hit_batch(g[name + "_o"], g[name + "_d"])
    tri = np.where(r["prim"] >= 0, r["tri"], -1)
    assert eq(tri, g[name + "_tri"])
    hit = tri >= 0
    assert eq(r["t"][hit], g[name + "_t"][hit]) and eq(r["uvw"][hit], g[name + "_uvw"][hit]) and eq(r["exiting"][hit], g[name + "_ex"][hit])


def test_mesh_files_on_device(orc, ns, golden, tmp_path):
    """SURVEY.md §8(f) row 4 on the device: meshes that come from FILES are traced. (1) RSM blobs written by the compiled reference
    (fixture F03) load through Mesh.from_file — KD-tree taken from the file — and answer the F04b ray sets with the reference's own
    triangle ids, distances and barycentrics; (2) an OBJ export / import_obj round trip of the cube (vertices exactly representable
    in the file's %e text) gives the same answers; (3) a displaced-sphere mesh re-imported from OBJ (vertices rounded by the text
    format) is traced identically by the device and the oracle."""
    import io
    g3, g4 = golden("f03_kd"), golden("f04b_small_meshes")

    def check(mesh, name):
        r = dev_scene(FlatScene([mesh])).hit_batch(g4[name + "_o"], g4[name + "_d"])
        tri = np.where(r["prim"] >= 0, r["tri"], -1)
        assert eq(tri, g4[name + "_tri"]), name
        hit = tri >= 0
        assert hit.sum() > 100
        assert eq(r["t"][hit], g4[name + "_t"][hit]) and eq(r["uvw"][hit], g4[name + "_uvw"][hit]) and eq(r["exiting"][hit], g4[name + "_ex"][hit]), name
    for name in ("cube", "sphere8", "blob24", "fan500"):
        check(ns.Mesh.from_file(io.BytesIO(g3[name].tobytes())), name)
    path = str(tmp_path / "cube.rsm")
    with open(path, "wb") as f:
        f.write(g3["blob24"].tobytes())
    check(ns.Mesh.from_file(path), "blob24")                  # by file name, as users call it
    v, t = scenes.cube_mesh()
    obj = str(tmp_path / "cube.obj")
    ns.export_obj(ns.Mesh(v, t, smoothing=False), obj)
    check(ns.import_obj(obj, smoothing=False), "cube")
    v, t = scenes.displaced_sphere(24, radius=0.5)
    ns.export_obj(ns.Mesh(v, t, smoothing=False), obj)
    world = ns.World()
    back = ns.import_obj(obj, smoothing=False, parent=world, transform=ns.translate(0.1, 0, 0) * ns.rotate(20, 10, 0))
    o, d, m = raysets.primitive_rays(20000, 5)
    flat = world.flatten()
    dev = dev_scene(flat).hit_batch(o, d, m, geometry=True)
    assert_hits_equal(dev, orc.hit_batch(flat, o, d, m, geometry=True))
    assert (dev["prim"] >= 0).sum() > 2000
    # ... and through World.hit(), the single-ray API
    first = int(np.nonzero(dev["prim"] >= 0)[0][0])
    hit = world.hit(ns.Ray(ns.Point3D(*o[first]), ns.Vector3D(*d[first]), max_distance=float(m[first])))
    assert hit is not None and hit.primitive is back and hit.triangle == dev["tri"][first] and hit.ray_distance == dev["t"][first]


def test_consecutive_passes_draw_fresh_samples(ns):
    """Two default observe() passes of spp samples are the same sample set as one pass of 2 spp (Philox counters advance between
    passes): merged mean equal to rel 1e-12, and different from what a repeated pass would give."""
    world, mesh, box = scenes.build_c2(ns, n=48)
    cam, pipe = scenes.c2_camera(ns, world, (96, 64), spp=6, bins=5)
    cam.frame_sampler = ns.RectFrameSampler2D()
    cam.render_engine = ns.HipEngine(rng="philox", seed=21)
    cam.observe()
    one = pipe.frame.mean.copy()
    cam.observe()
    two = pipe.frame.mean.copy()
    assert (pipe.frame.samples == 12).all()
    cam2, pipe2 = scenes.c2_camera(ns, world, (96, 64), spp=12, bins=5)
    cam2.frame_sampler = ns.RectFrameSampler2D()
    cam2.render_engine = ns.HipEngine(rng="philox", seed=21)
    cam2.observe()
    np.testing.assert_allclose(two, pipe2.frame.mean, rtol=1e-12, atol=1e-300)
    assert (two != one).mean() > 0.2                           # a repeated pass would have left the mean where it was


def test_analytic_primitives(orc, ns, golden):
    from tests.test_oracle_golden import _prims
    g = golden("f05_primitives")
    for k, (name, prim) in enumerate(_prims(ns).items()):
        flat = FlatScene([prim])
        sc = dev_scene(flat)
        o, d, m = raysets.primitive_rays(3000, 70 + k)
        counts, t, ex, geom, tri, uvw = sc.roots_batch(0, o, d, m, max_roots=2, geometry=True)
        ref = g[name]                                       # [n, 2, 14]: t, exiting, hit, inside, outside, normal (compiled reference)
        valid = ~np.isnan(ref[:, :, 0])
        assert eq(counts, valid.sum(axis=1)), name
        assert eq(t[valid], ref[:, :, 0][valid]) and eq(ex[valid], ref[:, :, 1][valid]), name
        assert eq(geom[valid], ref[:, :, 2:][valid]) and (tri == -1).all(), name
        # first root with full geometry through the world path
        dev = sc.hit_batch(o, d, m, geometry=True)
        assert_hits_equal(dev, orc.hit_batch(flat, o, d, m, geometry=True))
        first = valid[:, 0] & (dev["prim"] >= 0)
        assert eq(dev["geom"][first], ref[:, 0, 2:][first]), name
        assert eq(sc.contains_batch(raysets.points(2000, 90 + k, 1.2))[:, 0], g[name + "_contains"]), name


def test_primitive_object_api(ns, golden):
    """The reference's per-object API on top of the device: Primitive.hit(ray) -> Intersection, repeated next_intersection(),
    contains(point), World.hit(ray) — objects carry the same numbers the batch entry points return (golden f05)."""
    from tests.test_oracle_golden import _prims
    g = golden("f05_primitives")
    for k, (name, prim) in enumerate(_prims(ns).items()):
        o, d, m = raysets.primitive_rays(3000, 70 + k)
        ref = g[name]
        picks = [i for i in range(len(o)) if not np.isnan(ref[i, 1, 0])][:3] + [i for i in range(len(o)) if np.isnan(ref[i, 0, 0])][:1]
        for i in picks:
            ray = ns.Ray(ns.Point3D(*o[i]), ns.Vector3D(*d[i]), max_distance=float(m[i]))
            hit = prim.hit(ray)
            for root in range(2):
                if np.isnan(ref[i, root, 0]):
                    assert hit is None
                    break
                assert hit is not None and hit.primitive is prim and hit.ray is ray
                assert hit.ray_distance == ref[i, root, 0] and hit.exiting == bool(ref[i, root, 1])
                got = [hit.hit_point.x, hit.hit_point.y, hit.hit_point.z, hit.inside_point.x, hit.inside_point.y, hit.inside_point.z,
                       hit.outside_point.x, hit.outside_point.y, hit.outside_point.z, hit.normal.x, hit.normal.y, hit.normal.z]
                assert eq(got, ref[i, root, 2:])
                hit = prim.next_intersection()
        pts = raysets.points(2000, 90 + k, 1.2)[:40]
        assert eq([prim.contains(ns.Point3D(*q)) for q in pts], g[name + "_contains"][:40].astype(bool))
    # World.hit on a small world: same object identity and distance as the batch path
    world = ns.World()
    a = ns.Sphere(0.5, world, ns.translate(0, 0, 2))
    b = ns.Box(ns.Point3D(-1, -1, 4), ns.Point3D(1, 1, 5), world)
    hit = world.hit(ns.Ray(ns.Point3D(0, 0, 0), ns.Vector3D(0, 0, 1)))
    assert hit.primitive is a and hit.ray_distance == 1.5 and not hit.exiting
    hit = world.hit(ns.Ray(ns.Point3D(0.9, 0, 0), ns.Vector3D(0, 0, 1)))
    assert hit.primitive is b and hit.ray_distance == 4.0
    assert world.hit(ns.Ray(ns.Point3D(3, 0, 0), ns.Vector3D(0, 0, 1))) is None
    assert world.contains(ns.Point3D(0, 0, 2)) == [a]


def test_instanced_world_vs_oracle(orc, ns):
    world = scenes.build_c3(ns, n=32)[0]
    flat = world.flatten()
    sc = world.build_accelerator()
    o, d, m = raysets.scene_rays(30000, 201, 2.5, 0.9)
    assert_hits_equal(sc.hit_batch(o, d, m, geometry=True), orc.hit_batch(flat, o, d, m, geometry=True, threads=orc.max_threads()))
    pts = raysets.points(5000, 202, 1.0)
    assert eq(sc.contains_batch(pts), orc.contains_batch(flat, pts))


def _observe(ns, cam, pipe, seed, rng="stream"):
    import random as pyrandom
    from source_amd.core import random as rsrandom
    pyrandom.seed(seed)
    rsrandom.seed(seed)
    cam.render_engine = ns.HipEngine(rng=rng, seed=seed)
    cam.observe()
    f = pipe.frame
    return f.mean.copy(), f.variance.copy(), f.samples.copy()


def test_frames_c2_stream_parity(ns, golden):
    """observe() on the GPU == the reference's SerialEngine frame, bit for bit (same MT19937-64 jitter stream)."""
    g = golden("f10_frames")
    world, mesh, box = scenes.build_c2(ns, n=132)
    cam, pipe = scenes.c2_camera(ns, world, (40, 40), spp=4, bins=15)
    m, v, n = _observe(ns, cam, pipe, 1)
    assert eq(m, g["c2_mean"]) and eq(v, g["c2_var"]) and eq(n, g["c2_n"])
    # the same script a Raysect user writes: seed both generators, camera.render_engine = SerialEngine(), observe()
    import random as pyrandom
    from source_amd.core import random as rsrandom
    cam2, pipe2 = scenes.c2_camera(ns, world, (40, 40), spp=4, bins=15)
    pyrandom.seed(1)
    rsrandom.seed(1)
    cam2.render_engine = ns.SerialEngine()
    cam2.observe()
    assert eq(pipe2.frame.mean, g["c2_mean"]) and eq(pipe2.frame.variance, g["c2_var"])
    # an oversized slice is cut into several library calls (here: 777 rays per call); the MT stream is consumed in the same order
    cam3, pipe3 = scenes.c2_camera(ns, world, (40, 40), spp=4, bins=15)
    cam3.MAX_RAYS_PER_CALL = 777
    m, v, n = _observe(ns, cam3, pipe3, 1)
    assert eq(m, g["c2_mean"]) and eq(v, g["c2_var"]) and eq(n, g["c2_n"])
    # ... and in rect / Philox mode the bands reproduce the single-call frame
    frames = []
    for limit in (1 << 29, 5000):
        cam4, pipe4 = scenes.c2_camera(ns, world, (40, 40), spp=4, bins=15)
        cam4.MAX_RAYS_PER_CALL = limit
        cam4.frame_sampler = ns.RectFrameSampler2D()
        cam4.render_engine = ns.HipEngine(rng="philox", seed=3)
        cam4.observe()
        frames.append((pipe4.frame.mean.copy(), pipe4.frame.variance.copy(), pipe4.frame.samples.copy()))
    assert all(eq(a, b) for a, b in zip(*frames)) and (frames[0][2] == 4).all()
    m, v, n = _observe(ns, cam, pipe, 2)                     # accumulate=True second pass
    assert eq(m, g["c2_mean2"]) and eq(v, g["c2_var2"]) and eq(n, g["c2_n2"])


def test_frames_sliced_power_smoothing(ns, golden):
    g = golden("f10_frames")
    world, mesh, box = scenes.build_c2(ns, n=48, smoothing=True, with_normals=True)
    pipe = ns.SpectralPowerPipeline2D()
    cam = ns.PinholeCamera((24, 36), fov=45, sensitivity=2.5, parent=world, pipelines=[pipe], frame_sampler=ns.FullFrameSampler2D(),
                           transform=ns.translate(0, 0.16, -0.4) * ns.rotate(0, -12, 0))
    cam.pixel_samples, cam.spectral_bins, cam.spectral_rays, cam.quiet = 1, 7, 3, True
    cam.min_wavelength, cam.max_wavelength = 400.0, 700.0
    m, v, n = _observe(ns, cam, pipe, 3)
    assert eq(m, g["c2s_mean"]) and eq(v, g["c2s_var"]) and eq(n, g["c2s_n"])


def test_frames_instanced(ns, golden):
    g = golden("f10_frames")
    world = scenes.build_c3(ns, n=32)[0]
    cam, pipe = scenes.c3_camera(ns, world, (32, 32), spp=3, bins=4)
    m, v, n = _observe(ns, cam, pipe, 5)
    assert eq(m, g["c3_mean"]) and eq(v, g["c3_var"]) and eq(n, g["c3_n"])


def test_unfused_engine_contract(ns, golden):
    """RenderEngine contract path: update((task, [(mean, var)], ray_count), slice_id) per task gives the same frame."""
    g = golden("f10_frames")
    import random as pyrandom
    from source_amd.core import random as rsrandom
    world, mesh, box = scenes.build_c2(ns, n=132)
    cam, pipe = scenes.c2_camera(ns, world, (40, 40), spp=4, bins=15)
    pyrandom.seed(1)
    rsrandom.seed(1)
    cam.render_engine = ns.HipEngine(rng="stream", fused=False)
    cam.observe()
    assert eq(pipe.frame.mean, g["c2_mean"]) and eq(pipe.frame.variance, g["c2_var"]) and eq(pipe.frame.samples, g["c2_n"])


def test_philox_frame_vs_oracle_full_size(orc, ns):
    """Throughput mode at BASELINE configs[1] size (1024x1024, 1 spp): device frame == oracle frame bit for bit, and is
    independent of how the frame is tiled (the multi-GPU sharding property)."""
    from source_amd import _lib
    world, mesh, box = scenes.build_c2(ns, n=132)
    cam, pipe = scenes.c2_camera(ns, world, (1024, 1024), spp=1, bins=15)
    cam.render_engine = ns.HipEngine(rng="philox", seed=7)
    cam.frame_sampler = RectSampler()
    cam.observe()
    mean = pipe.frame.mean
    assert (pipe.frame.samples == 1).all() and (pipe.frame.variance == 0).all()
    # oracle on a 1024 x 16 strip (bounded CPU time)
    flat = world.flatten()
    sl = cam._slice_spectrum()[0]
    keep = []
    desc = cam.render_desc(world, None, sl, cam.render_engine, keep, rect=(0, 0, 1024, 1024))     # the whole frame
    m, v, rays = orc.render_pinhole(flat, desc, threads=orc.max_threads())
    assert eq(mean, m.reshape(1024, 1024, 15).transpose(1, 0, 2))      # rect tasks are iy-outer / ix-inner
    assert mean.max() > 0 and (mean[512, 512] > 0).all()




@pytest.mark.parametrize("spp", [12, 16])
def test_packet_form_frame_vs_oracle_directly(orc, ns, spp):
    """The packet walk (dev_packet.hpp) tied to the oracle WITHOUT the per-lane kernel in between: whole frames of the instanced
    scene of configs[2] and of the single-mesh scene of configs[1] at 12 samples per pixel (units cut across pixels, separate
    Welford kernel) and 16 (four pixels per unit, Welford fused into the trace kernel) — mean and variance equal the oracle's
    bit for bit (kdtree3d.pyx:609-700, mesh.pyx:506-713, statsarray.pyx:743-776 per ray of each packet)."""
    for build, camera, size in ((lambda: scenes.build_c3(ns, n=40)[0], scenes.c3_camera, (160, 96)), (lambda: scenes.build_c2(ns, n=48)[0], scenes.c2_camera, (128, 96))):
        world = build()
        cam, pipe = camera(ns, world, size, spp=spp, bins=7)
        cam.render_engine = ns.HipEngine(rng="philox", seed=23, auto_batch=False)
        cam.frame_sampler = ns.RectFrameSampler2D()
        cam.observe()
        keep = []
        desc = cam.render_desc(world, None, cam._slice_spectrum()[0], cam.render_engine, keep, rect=(0, 0) + size)
        m, v, rays = orc.render_pinhole(world.flatten(), desc, threads=orc.max_threads())
        assert eq(pipe.frame.mean, m.reshape(size[1], size[0], 7).transpose(1, 0, 2)) and eq(pipe.frame.variance, v.reshape(size[1], size[0], 7).transpose(1, 0, 2))
        assert (pipe.frame.samples == spp).all() and pipe.frame.variance.max() > 0


def _full_size_properties(orc, ns, world, cam, pipe, make_camera, strip_rows, spp, seed):
    """Shared body of the BASELINE full-size checks: one Philox pass on the device, then
      (1) every frame element holds exactly spp samples, mean finite,
      (2) a strip of rows equals the oracle's render of the same rows bit for bit (mean AND variance: the per-pixel Welford
          order s = 0..spp-1 is fixed, statsarray.pyx:743-776),
      (3) tile sharding: rendering a column band alone reproduces that band of the full frame bit for bit,
      (4) sample sharding: two spp/2 passes with sample offsets 0 and spp/2 merged by combine_samples agree with the single pass
          to rel 1e-12 in the mean (different but equivalent summation order, statsarray.pyx:780-859)."""
    nx, ny = cam.pixels
    cam.render_engine = ns.HipEngine(rng="philox", seed=seed)
    cam.frame_sampler = ns.RectFrameSampler2D()
    cam.observe()
    f = pipe.frame
    mean, var, n = f.mean.copy(), f.variance.copy(), f.samples.copy()
    assert (n == spp).all() and np.isfinite(mean).all() and np.isfinite(var).all() and (var >= 0).all() and mean.max() > 0
    flat = world.flatten()
    sl = cam._slice_spectrum()[0]
    y0 = ny // 2 - strip_rows // 2
    keep = []
    desc = cam.render_desc(world, None, sl, cam.render_engine, keep, rect=(0, y0, nx, y0 + strip_rows))
    m, v, rays = orc.render_pinhole(flat, desc, threads=orc.max_threads())
    bins = mean.shape[2]
    assert rays == nx * strip_rows * spp
    assert eq(mean[:, y0:y0 + strip_rows, :], m.reshape(strip_rows, nx, bins).transpose(1, 0, 2))
    assert eq(var[:, y0:y0 + strip_rows, :], v.reshape(strip_rows, nx, bins).transpose(1, 0, 2))
    # (3) a band of columns rendered on its own (what one rank of a tile-sharded job does)
    x0, x1 = (3 * nx) // 8, (5 * nx) // 8
    cam2, pipe2 = make_camera()
    cam2.render_engine = ns.HipEngine(rng="philox", seed=seed)
    cam2.frame_sampler = ns.RectFrameSampler2D(rect=(x0, 0, x1, ny))
    cam2.observe()
    f2 = pipe2.frame
    assert eq(f2.mean[x0:x1], mean[x0:x1]) and eq(f2.variance[x0:x1], var[x0:x1]) and (f2.samples[x0:x1] == spp).all()
    assert (f2.samples[:x0] == 0).all() and (f2.samples[x1:] == 0).all()
    # (4) the same spp samples split over two passes (what two ranks of a sample-sharded job do), merged by combine_samples
    cam3, pipe3 = make_camera()
    cam3.pixel_samples = spp // 2
    cam3.frame_sampler = ns.RectFrameSampler2D(rect=(x0, 0, x1, ny))
    eng = ns.HipEngine(rng="philox", seed=seed)
    cam3.render_engine = eng
    cam3.observe()
    eng.sample_offset = spp // 2
    cam3.observe()
    f3 = pipe3.frame
    assert (f3.samples[x0:x1] == spp).all()
    np.testing.assert_allclose(f3.mean[x0:x1], mean[x0:x1], rtol=1e-12, atol=1e-300)
    # combine_samples forms E[x^2] - mean^2 (statsarray.pyx:815-819), so the merged variance carries an absolute error of a few
    # ulp of mean^2 whatever the arithmetic: tolerance = 16 eps (mean^2 + var)
    tol = 16 * np.finfo(np.float64).eps * (mean[x0:x1] ** 2 + var[x0:x1])
    assert (np.abs(f3.variance[x0:x1] - var[x0:x1]) <= tol).all()


def test_c3_full_size_instanced_1m_triangles(orc, ns):
    """BASELINE configs[2] at full size: 15 instances of the 69 432-triangle mesh (1 041 480 triangles) + floor box,
    2048x2048, 64 samples/pixel = 268 M primary rays in one pass."""
    world = scenes.build_c3(ns, n=132)[0]
    make = lambda: scenes.c3_camera(ns, world, (2048, 2048), spp=64, bins=15)
    cam, pipe = make()
    _full_size_properties(orc, ns, world, cam, pipe, make, strip_rows=2048, spp=64, seed=11)     # the WHOLE 268 M-ray frame against the oracle


def test_flat_1m_triangle_mesh_against_reference_digest_and_oracle(orc, ns, golden, m1m):
    """SURVEY.md 8(d) "M1M-flat" — the bench's `flat` workload, ONE mesh of 1 048 576 triangles: the tree that leaves the caches (66 MB of
    nodes + items on the host, 0.45 GB of node and leaf records on the device). F18 pins it to the compiled reference: (1) the device's
    answers to one million rays = the reference's SHA-256 of (triangle, t, u, v, w, exiting) — Mesh.hit of the reference's own tree over
    the same triangles (the tree bytes themselves are pinned on the CPU side, tests/test_oracle_golden.py); (2) the 2048 x 2048 x 64 spp
    frame of scenes.build_flat through the packet kernel: every element holds 64 samples and a strip of 64 rows (8.4 M rays) equals the
    oracle's mean and variance bit for bit."""
    g = golden("f18_flat")
    mesh, v, t = m1m
    sc = dev_scene(FlatScene([mesh]))
    o, d, m = raysets.random_outside(1000000, 51)
    r = sc.hit_batch(o, d, m)
    tri = np.where(r["prim"] >= 0, r["tri"], -1).astype(np.int32)
    tt = np.where(tri >= 0, r["t"], np.nan)
    uvw = np.where((tri >= 0)[:, None], r["uvw"], 0).astype(np.float32)
    ex = np.where(tri >= 0, r["exiting"], 0).astype(np.uint8)
    assert eq(tri[:512], g["first_tri"]) and eq(tt[:512], g["first_t"]) and eq(uvw[:512], g["first_uvw"]) and eq(ex[:512], g["first_ex"])
    h = hashlib.sha256()
    for a in (tri, tt, uvw, ex):
        h.update(np.ascontiguousarray(a).tobytes())
    assert int((tri >= 0).sum()) == int(g["digest_1m_hits"][0])
    assert h.digest() == bytes(g["digest_1m"])
    del sc
    world = scenes.build_flat(ns, n=512)[0]
    cam, pipe = scenes.c2_camera(ns, world, (2048, 2048), spp=64, bins=15)
    cam.render_engine = ns.HipEngine(rng="philox", seed=19)
    cam.frame_sampler = ns.RectFrameSampler2D()
    cam.observe()
    f = pipe.frame
    mean, var, n = f.mean, f.variance, f.samples
    assert (n == 64).all() and np.isfinite(mean).all() and (var >= 0).all() and mean.max() > 0
    strip_rows, nx, ny = 64, 2048, 2048
    y0 = ny // 2 - strip_rows // 2
    keep = []
    desc = cam.render_desc(world, None, cam._slice_spectrum()[0], cam.render_engine, keep, rect=(0, y0, nx, y0 + strip_rows))
    mo, vo, rays = orc.render_pinhole(world.flatten(), desc, threads=orc.max_threads())
    assert rays == nx * strip_rows * 64
    assert eq(mean[:, y0:y0 + strip_rows, :], mo.reshape(strip_rows, nx, 15).transpose(1, 0, 2))
    assert eq(var[:, y0:y0 + strip_rows, :], vo.reshape(strip_rows, nx, 15).transpose(1, 0, 2))


def test_c4_full_size_csg_demo(orc, ns):
    """BASELINE configs[3] at full size: the demos/csg.py Boolean tree, 1024x1024, 16 samples/pixel."""
    world = scenes.build_csg_demo(ns)[0]
    make = lambda: scenes.csg_camera(ns, world, (1024, 1024), spp=16, bins=15)
    cam, pipe = make()
    _full_size_properties(orc, ns, world, cam, pipe, make, strip_rows=1024, spp=16, seed=13)     # the whole frame against the oracle


def test_c5_shape_512_spectral_slices(orc, ns):
    """BASELINE configs[4]'s frame shape with the closed-form materials this round covers: 1024x1024, 512 spectral bins rendered
    as 512 one-bin slices (spectral_rays = 512, observer.pyx:311-340), 1 spp — a 10.7 GB device-resident frame and 537 M primary
    rays. Checks slice offsets over the whole frame, a strip against the oracle and the per-bin spectral table."""
    world, mesh, box = scenes.build_c2(ns, n=132)
    sf = ns.InterpolatedSF([300, 490, 510, 590, 610, 800], np.array([0.0, 0.1, 1.0, 0.7, 0.2, 0.4]))
    mesh.material = ns.Light(ns.Vector3D(-1, -1, 1), 1.0, sf)
    cam, pipe = scenes.c2_camera(ns, world, (1024, 1024), spp=1, bins=512)
    cam.spectral_rays = 512
    cam.render_engine = ns.HipEngine(rng="philox", seed=17)
    cam.frame_sampler = ns.RectFrameSampler2D()
    cam.observe()
    f = pipe.frame
    assert f.shape == (1024, 1024, 512)
    n = f.samples
    assert (n == 1).all()
    mean = f.mean
    assert np.isfinite(mean).all() and (f.variance == 0).all()
    # every slice is an independent 1-bin render with its own jitter: compare three slices' strips with the oracle
    flat = world.flatten()
    slices = cam._slice_spectrum()
    assert len(slices) == 512 and all(s.bins == 1 and s.offset == k for k, s in enumerate(slices))
    for k in (0, 255, 511):
        keep = []
        desc = cam.render_desc(world, None, slices[k], cam.render_engine, keep, rect=(0, 508, 1024, 516))
        m, v, rays = orc.render_pinhole(flat, desc, threads=orc.max_threads())
        assert eq(mean[:, 508:516, k], m.reshape(8, 1024).T)
    # slices draw independent jitter (one Philox key per slice): normalised by their table value two slices still differ
    table = sf.sample(cam.min_wavelength, cam.max_wavelength, 512)
    assert not np.allclose(mean[:, 508:516, 200] / table[200], mean[:, 508:516, 300] / table[300], rtol=1e-9)
    f.release()


def test_c5_prism_512_slices_full_size(orc, ns):
    """BASELINE configs[4] as the combination it names: the dispersive-prism scene (demos/prism.py geometry: nested analytic CSG, two
    Sellmeier glasses, Lambert screen, importance sampling towards the prism) at 1024x1024 with 512 spectral bins rendered as 512
    one-bin slices — every slice bends through its own refractive index — 1 sample per pixel per pass: 537 M paths into a 10.7 GB
    device-resident frame. Strips of three slices against the oracle (mean; the variance of a 1-sample pass is 0), sample counts over
    the whole frame, and the dispersion itself (the far slices light different screen pixels)."""
    import time
    world, prims = scenes.build_prism(ns)
    cam, pipe = scenes.prism_camera(ns, world, (1024, 1024), 1, 512, 512)
    cam.frame_sampler = ns.RectFrameSampler2D()
    cam.render_engine = ns.HipEngine(rng="philox", seed=29)
    t0 = time.perf_counter()
    cam.observe()
    f = pipe.frame
    assert f.shape == (1024, 1024, 512)
    assert (f.samples == 1).all()
    elapsed = time.perf_counter() - t0
    mean = f.mean
    assert np.isfinite(mean).all() and (f.variance == 0).all()
    flat = world.flatten()
    slices = cam._slice_spectrum()
    assert len(slices) == 512 and all(s.bins == 1 and s.offset == k for k, s in enumerate(slices))
    lit = []
    for k in (3, 256, 508):
        keep = []
        desc = cam.render_desc(world, None, slices[k], cam.render_engine, keep, rect=(0, 600, 1024, 606))
        m, v, rays = orc.render_pinhole(flat, desc, threads=orc.max_threads())
        assert eq(mean[:, 600:606, k], m.reshape(6, 1024).T), k
        lit.append(mean[:, :, k] > 0)
    assert lit[0].sum() > 1000 and lit[2].sum() > 1000 and (lit[0] != lit[2]).sum() > 1000     # dispersion: blue and red land apart
    print("configs[4] combination: 512 slices x 1024^2 x 1 spp in %.1f s (%.3g paths/s, %d rays)" % (elapsed, 512 * 1024 * 1024 / elapsed, cam.stats["rays"]))
    f.release()


def test_c5_accumulating_passes_against_merged_oracle(orc, ns):
    """BASELINE configs[4] as WRITTEN is 256 samples per pixel: passes of 16 accumulated into the frame (observer.pyx:265-340 called again
    and again into an accumulating pipeline, power.pyx:399-437). The whole thing runs once per round from tools/c5_256spp.py (kept
    log under profiles/); here its shape in the suite's budget: the prism scene at 256 x 256, 64 one-bin spectral slices, TWO passes of
    16 samples per pixel — 134 M paths through the overlapping slice launches, the redo passes and the frame merges — and three slice
    strips rendered by the oracle with the same Philox counters, pass by pass, merged with the reference's combine_samples law:
    sample counts, means and variances EQUAL (the merge order is the reference's: pass after pass into the frame)."""
    from source_amd import distributed as D
    world, prims = scenes.build_prism(ns)
    NX = NY = 256
    cam, pipe = scenes.prism_camera(ns, world, (NX, NY), 16, 64, 64)
    cam.frame_sampler = ns.RectFrameSampler2D()
    cam.render_engine = ns.HipEngine(rng="philox", seed=29)
    cam.observe()
    cam.observe()
    f = pipe.frame
    assert f.shape == (NX, NY, 64) and (f.samples == 32).all()
    mean, var = f.mean, f.variance
    assert np.isfinite(mean).all() and np.isfinite(var).all() and (var >= 0).all() and (mean > 0).sum() > 500
    flat = world.flatten()
    slices = cam._slice_spectrum()
    rect = (0, 148, NX, 156)
    for k in (2, 31, 61):
        om = ov = on = None
        for p in range(2):
            keep = []
            desc = cam.render_desc(world, None, slices[k], cam.render_engine, keep, rect=rect, sample_offset=p * 16)
            m, v, rays = orc.render_pinhole(flat, desc, threads=orc.max_threads())
            m, v = m.reshape(8, NX).T, v.reshape(8, NX).T
            cnt = np.full(m.shape, 16, dtype=np.int32)
            if om is None:
                om, ov, on = m, np.maximum(v, 0.0), cnt
            else:
                om, ov, on = D.combine_arrays(om, ov, on, m, np.maximum(v, 0.0), cnt)
        assert (on == 32).all() and (om > 0).sum() > 0, k
        assert eq(mean[:, 148:156, k], om) and eq(var[:, 148:156, k], ov), k
    f.release()


def test_edge_semantics_on_device(orc, ns, golden):
    """Fixture F11 on the device: empty world, coincident primitives, t == max_distance, surface origins, axis-parallel grazing
    rays, zero-length batches, and 1x2 / 3x5x1 / masked (ragged task list) frames — all bit-exact against the compiled reference."""
    from tests.test_oracle_golden import EDGE_FRAMES
    g = golden("f11_edges")
    for name, (world, prims) in scenes.build_edge_worlds(ns).items():
        sc = world.build_accelerator()
        o, d, m = scenes.edge_rays(name)
        dev = sc.hit_batch(o, d, m, geometry=True)
        _check_world(dev, g[name + "_idx"], g[name + "_rec"])
        pts = np.concatenate([o, o + 0.25 * d])
        assert eq(sc.contains_batch(pts), g[name + "_contains"]), name
        empty = sc.hit_batch(np.zeros((0, 3)), np.zeros((0, 3)), np.zeros(0))            # n = 0 is a valid call
        assert len(empty["prim"]) == 0
        # the object API agrees (World.hit returns None / the right object)
        for k in range(min(4, len(o))):
            hit = world.hit(ns.Ray(ns.Point3D(*o[k]), ns.Vector3D(*d[k]), max_distance=float(m[k])))
            assert (hit is None) == (g[name + "_idx"][k] < 0)
            if hit is not None:
                assert hit.primitive is prims[int(g[name + "_idx"][k])] and hit.ray_distance == g[name + "_rec"][k, 0]
    world, mesh, box = scenes.build_c2(ns, n=24)
    for tag, pixels, spp, bins, mask in EDGE_FRAMES:
        cam, pipe = scenes.edge_camera(ns, world, pixels, spp, bins, mask)
        mean, var, n = _observe(ns, cam, pipe, 21)
        assert eq(mean, g[tag + "_mean"]) and eq(var, g[tag + "_var"]) and eq(n, g[tag + "_n"]), tag


# ------------------------------------------------------------------------------------------------- CSG on the device
def _check_world(dev, idx, rec):
    assert eq(dev["prim"], idx)
    hit = idx >= 0
    assert eq(dev["t"][hit], rec[hit, 0])
    assert eq(dev["exiting"][hit], rec[hit, 1])
    assert eq(dev["geom"][hit], rec[hit, 2:])


def test_csg_demo_world(orc, ns, golden):
    """demos/csg.py tree (Intersect(sphere, Subtract(cube, Union(Union(cyl, cyl), cyl))) x4 + lens): hits with full
    geometry, next_intersection sequences and contains() against the reference's vectors."""
    g = golden("f06_csg")
    world, prims = scenes.build_csg_demo(ns)
    flat = world.flatten()
    sc = world.build_accelerator()
    o, d, m = raysets.scene_rays(12000, 101, 9.0, 4.5)
    og, dg, mg = raysets.pinhole_grid(64, (0.0, 0.0, -4.0), 75.0)
    o, d, m = np.concatenate([o, og]), np.concatenate([d, dg]), np.concatenate([m, mg])
    dev = sc.hit_batch(o, d, m, geometry=True)
    _check_world(dev, g["world_idx"], g["world_rec"])
    assert_hits_equal(dev, orc.hit_batch(flat, o, d, m, geometry=True), geometry=True)
    from source_amd.device import HostScene                  # the single-ray host walk (csrc/rsx_hostwalk.cpp) answers CSG worlds with the device's bits
    assert_hits_equal(HostScene(flat).hit_batch(o, d, m, geometry=True), dev, geometry=True)
    for name, index in (("obj0", 0), ("lens", 4)):
        counts, t, ex = sc.roots_batch(index, o[:4000], d[:4000], None, max_roots=64)
        assert eq(counts, g[name + "_counts"]), name
        mask = np.arange(64)[None, :] < counts[:, None]
        assert eq(t[mask], g[name + "_t"]) and eq(ex[mask], g[name + "_ex"]), name
    assert eq(sc.contains_batch(raysets.points(4000, 102, 4.5)), g["contains"])


def test_mixed_world_with_csg_and_instances(orc, ns, golden):
    g = golden("f07_world")
    world, prims = scenes.build_mixed(ns)
    sc = world.build_accelerator()
    o, d, m = raysets.scene_rays(20000, 111, 6.0, 2.2)
    dev = sc.hit_batch(o, d, m, geometry=True)
    _check_world(dev, g["idx"], g["rec"])
    assert not np.isin(dev["prim"], [2, 3]).any()
    assert eq(sc.contains_batch(raysets.points(6000, 112, 2.0)), g["contains"])


def test_frames_csg_stream_parity(ns, golden):
    g = golden("f10_frames")
    world, prims = scenes.build_csg_demo(ns)
    cam, pipe = scenes.csg_camera(ns, world, (32, 32), spp=6, bins=5)
    m, v, n = _observe(ns, cam, pipe, 4)
    assert eq(m, g["csg_mean"]) and eq(v, g["csg_var"]) and eq(n, g["csg_n"])


def test_csg_with_mesh_operand_vs_oracle(orc, ns):
    """A mesh as CSG operand exercises the mesh next_intersection stream inside the merge (no golden: oracle is the checker)."""
    v, t = scenes.displaced_sphere(24, radius=0.5)
    world = ns.World()
    mesh = ns.Mesh(v, t, smoothing=False, transform=ns.translate(0.1, 0, 0))
    cut = ns.Box(ns.Point3D(-0.3, -1, -1), ns.Point3D(0.25, 1, 1))
    ns.Subtract(mesh, cut, world, ns.translate(0, 0.1, 0.2) * ns.rotate(10, 20, 30), ns.AbsorbingSurface())
    mesh2 = ns.Mesh(v, t, smoothing=False)
    ns.Intersect(ns.Sphere(0.45, transform=ns.translate(0.2, 0, 0)), mesh2, world, ns.translate(1.5, 0, 0), ns.AbsorbingSurface())
    flat = world.flatten()
    sc = world.build_accelerator()
    o, d, m = raysets.scene_rays(20000, 301, 4.0, 1.2)
    assert_hits_equal(sc.hit_batch(o, d, m, geometry=True), orc.hit_batch(flat, o, d, m, geometry=True))
    for index in (0, 1):
        dc, dt, de, dg, dtri, duvw = sc.roots_batch(index, o[:3000], d[:3000], None, max_roots=16, geometry=True)
        oc, ot, oe, og = orc.roots_batch(flat, index, o[:3000], d[:3000], None, max_roots=16, geometry=True)
        mask = np.arange(16)[None, :] < oc[:, None]
        assert eq(dc, oc) and eq(dt[mask], ot[mask]) and eq(de[mask], oe[mask]) and eq(dg[mask], og[mask])
    pts = raysets.points(3000, 302, 1.5)
    assert eq(sc.contains_batch(pts), orc.contains_batch(flat, pts))


def test_exact_division():
    """The hoisted-reciprocal division used at every KD node equals the IEEE quotient bit for bit (2^28 operand pairs)."""
    import ctypes as C
    from source_amd import _lib
    from source_amd.device import get_context
    bad = C.c_uint64(123)
    for seed in (1, 2):
        _lib.check(_lib.lib().rsx_selftest_exact_division(get_context().handle, 1 << 28, seed, C.byref(bad)))
        assert bad.value == 0


def test_frames_volume_emitters_stream_parity(orc, ns, golden):
    """NullMaterial + UniformVolumeEmitter on the device (k_render_trace_path): frames bit-identical to the reference's SerialEngine,
    incl. the accumulate pass; Philox mode equals the oracle, also with forty emitters overlapping at a point."""
    g = golden("f12_volumes")
    world, prims = scenes.build_volumes(ns)
    cam, pipe = scenes.volumes_camera(ns, world)
    m, v, n = _observe(ns, cam, pipe, 31)
    assert eq(m, g["mean"]) and eq(v, g["var"]) and eq(n, g["n"])
    m, v, n = _observe(ns, cam, pipe, 32)
    assert eq(m, g["mean2"]) and eq(v, g["var2"]) and eq(n, g["n2"])
    open_world, _ = scenes.build_volumes(ns, enclosed=False)
    cam_o, pipe_o = scenes.volumes_camera(ns, open_world)
    m, v, n = _observe(ns, cam_o, pipe_o, 33)
    assert eq(m, g["open_mean"]) and eq(v, g["open_var"]) and eq(n, g["open_n"])
    # throughput mode at a larger size against the oracle
    cam2, pipe2 = scenes.volumes_camera(ns, world, (256, 192), spp=8, bins=6)
    cam2.frame_sampler = ns.RectFrameSampler2D()
    cam2.render_engine = ns.HipEngine(rng="philox", seed=9)
    cam2.observe()
    keep = []
    desc = cam2.render_desc(world, None, cam2._slice_spectrum()[0], cam2.render_engine, keep, rect=(0, 0, 256, 192))
    om, ov, rays = orc.render_pinhole(world.flatten(), desc, threads=orc.max_threads())
    assert eq(pipe2.frame.mean, om.reshape(192, 256, 6).transpose(1, 0, 2)) and eq(pipe2.frame.variance, ov.reshape(192, 256, 6).transpose(1, 0, 2))
    # the single-ray API walks the same path on the host side of the plugin interface: a unit-scale emitting sphere of radius 0.5
    # seen along a diameter radiates its diameter (minus the two 1e-9 surface nudges), through a transparent shell and back out
    single = ns.World()
    ns.Sphere(0.5, single, ns.translate(0, 0, 2), ns.UniformVolumeEmitter(ns.ConstantSF(1.0), 1.0))
    ns.Sphere(0.8, single, ns.translate(0, 0, 2), ns.NullMaterial())
    spectrum = ns.Ray(ns.Point3D(0, 0, 0), ns.Vector3D(0, 0, 1), bins=4).trace(single)
    assert np.allclose(spectrum.samples, 1.0, rtol=0, atol=1e-8) and (spectrum.samples < 1.0).all()
    # forty nested emitting shells: ten times more emitters overlap at a point than the path kernel keeps in registers — the extra
    # terms come from re-walking world.contains() (a former RSX_EUNSUPPORTED limit); the frame equals the oracle's
    deep = ns.World()
    for k in range(40):
        ns.Sphere(0.2 + 0.02 * k, deep, ns.translate(0, 0, 3), ns.UniformVolumeEmitter(ns.ConstantSF(1.0), 0.1))
    cam3, pipe3 = scenes.volumes_camera(ns, deep, (16, 16), spp=1, bins=2)
    cam3.frame_sampler = ns.RectFrameSampler2D()
    cam3.render_engine = ns.HipEngine(rng="philox", seed=1)
    cam3.observe()
    keep = []
    desc = cam3.render_desc(deep, None, cam3._slice_spectrum()[0], cam3.render_engine, keep, rect=(0, 0, 16, 16))
    om, ov, rays = orc.render_pinhole(deep.flatten(), desc, threads=orc.max_threads())
    assert eq(pipe3.frame.mean, om.reshape(16, 16, 2).transpose(1, 0, 2)) and pipe3.frame.mean.max() > 0.1 and cam3.stats["rays"] == rays


def test_frames_lambert_against_oracle(orc, ns):
    """Lambert scattering on the device (RSX_MAT_LAMBERT, k_render_trace_path): Philox-keyed paths, term lists replayed per bin in
    the reference's order. The oracle — pinned bit for bit to the reference's SerialEngine frames by fixture F13 on the CPU side —
    renders the same Philox paths; frames must be identical: observer defaults (roulette 0.01 from depth 3, depth limit 500: long
    paths that chain arena blocks), an aggressive roulette with a shallow depth limit, and spectral slices."""
    world, prims = scenes.build_lambert(ns)
    for pixels, spp, bins, ext, rays in (((96, 80), 8, 5, (0.01, 3, 500), 1), ((64, 48), 6, 6, (0.3, 1, 4), 2), ((40, 40), 3, 4, (0.1, 2, 12), 1)):
        cam, pipe = scenes.lambert_camera(ns, world, pixels, spp, bins, ext)
        cam.spectral_rays = rays
        cam.frame_sampler = ns.RectFrameSampler2D()
        cam.render_engine = ns.HipEngine(rng="philox", seed=77)
        cam.observe()
        w, h = pixels
        ref_m, ref_v = np.zeros((w, h, bins)), np.zeros((w, h, bins))
        ref_rays = 0
        for sl in cam._slice_spectrum():
            keep = []
            desc = cam.render_desc(world, None, sl, cam.render_engine, keep, rect=(0, 0, w, h))
            om, ov, n_rays = orc.render_pinhole(world.flatten(), desc, threads=orc.max_threads())
            ref_rays += n_rays
            ref_m[:, :, sl.offset:sl.offset + sl.bins] = om.reshape(h, w, sl.bins).transpose(1, 0, 2)
            ref_v[:, :, sl.offset:sl.offset + sl.bins] = ov.reshape(h, w, sl.bins).transpose(1, 0, 2)
        assert eq(pipe.frame.mean, ref_m) and eq(pipe.frame.variance, ref_v), (pixels, ext)
        assert cam.stats["rays"] == ref_rays                      # Ray.ray_count: primary rays + every daughter spawned
        if ext[2] == 500:
            assert (pipe.frame.mean > 0).mean() > 0.5              # light reaches most pixels only through diffuse bounces
    # the reference's stream engine cannot drive scattering paths in parallel: loud error, no silent substitute
    from source_amd._lib import RsxError
    cam, pipe = scenes.lambert_camera(ns, world, (8, 8), 1, 2)
    cam.render_engine = ns.SerialEngine()
    with pytest.raises(RsxError):
        cam.observe()


def test_all_wide_worlds_take_their_answers_without_the_walk_and_ties_walk(orc, ns, monkeypatch):
    """A world of at most eight analytic primitives (no CSG, no mesh) has every primitive answered before the per-lane walk; the path
    kernel then takes a ray's nearest eligible answer and skips the walk (world_trace_wave, DScene::all_wide8) — except where the
    nearest answer is shared by two primitives, which the leaf's item order decides (kdtree.pyx:99-116). This world is made of ties:
    two spheres that are the same sphere with different materials, a slab whose top face lies in the floor's, a box that shares a
    wall's face plane. Frames equal the oracle's (whose walk is the reference's), and the same frames come out with the short cut off."""
    P = ns.Point3D
    red = ns.InterpolatedSF([300, 560, 600, 800], np.array([0.08, 0.1, 0.75, 0.8]))
    blue = ns.InterpolatedSF([300, 440, 480, 800], np.array([0.7, 0.75, 0.1, 0.08]))
    white = ns.ConstantSF(0.7)

    def build():
        world = ns.World()
        ns.Box(P(-1.0, -1.05, 0.0), P(1.0, -1.0, 2.0), world, material=ns.Lambert(white))                      # floor
        ns.Box(P(-0.5, -1.04, 0.6), P(0.1, -1.0, 1.2), world, material=ns.Lambert(red))                        # slab: its top face IS the floor's
        ns.Box(P(-1.0, -1.0, 2.0), P(1.0, 1.0, 2.05), world, material=ns.Lambert(white))                       # back wall
        ns.Box(P(0.3, -0.2, 1.7), P(0.8, 0.4, 2.0), world, material=ns.Lambert(blue))                          # its far face lies in the wall's near face
        ns.Sphere(0.3, world, ns.translate(-0.35, -0.4, 1.2), ns.Lambert(red))                                 # the same sphere twice
        ns.Sphere(0.3, world, ns.translate(-0.35, -0.4, 1.2), ns.Lambert(blue))
        ns.Box(P(-0.6, 0.98, 0.4), P(0.6, 0.999, 1.6), world, material=ns.UniformSurfaceEmitter(ns.ConstantSF(1.0), 6.0))
        ns.Cylinder(0.15, 0.5, world, ns.translate(0.5, -1.0, 0.8) * ns.rotate(0, -90, 0), ns.Lambert(white))  # stands on the floor plane
        return world

    frames = {}
    for off in (False, True):
        if off:
            monkeypatch.setenv("RSX_NO_PKT_CLUSTERS", "1")
        world = build()
        assert len(world.primitives) == 8
        cam, pipe = scenes.lambert_camera(ns, world, (72, 60), 6, 5, (0.05, 3, 40))
        cam.frame_sampler = ns.RectFrameSampler2D()
        cam.render_engine = ns.HipEngine(rng="philox", seed=123)
        cam.observe()
        frames[off] = (pipe.frame.mean.copy(), pipe.frame.variance.copy(), cam.stats["rays"])
        if not off:
            w, h = 72, 60
            keep = []
            sl = list(cam._slice_spectrum())[0]
            desc = cam.render_desc(world, None, sl, cam.render_engine, keep, rect=(0, 0, w, h))
            om, ov, n_rays = orc.render_pinhole(world.flatten(), desc, threads=orc.max_threads())
            assert eq(pipe.frame.mean, om.reshape(h, w, sl.bins).transpose(1, 0, 2)) and eq(pipe.frame.variance, ov.reshape(h, w, sl.bins).transpose(1, 0, 2))
            assert cam.stats["rays"] == n_rays
            assert (pipe.frame.mean > 0).mean() > 0.2              # (an open scene: most of the frame looks past it)
    assert eq(frames[False][0], frames[True][0]) and eq(frames[False][1], frames[True][1]) and frames[False][2] == frames[True][2]


def test_csg_worlds_answered_before_the_walk_skip_it_and_ties_walk(orc, ns, monkeypatch):
    """The same short cut in the fast forms of a CSG scene (DScene::all_answered_csg): at most four analytic primitives in the wide slots,
    at most four solids answered by the prefill round — nearest eligible answer, ties walk. Here two of the solids are the same solid
    with different materials (every ray that meets them ties) and a box shares a face plane with a solid's operand. Frames equal the
    oracle's, and the same frames come out with the short cut off."""
    P = ns.Point3D
    red = ns.InterpolatedSF([300, 560, 600, 800], np.array([0.08, 0.1, 0.75, 0.8]))
    blue = ns.InterpolatedSF([300, 440, 480, 800], np.array([0.7, 0.75, 0.1, 0.08]))
    white = ns.ConstantSF(0.7)

    def cup(material, world):
        return ns.Subtract(ns.Box(P(-0.3, -0.3, -0.3), P(0.3, 0.3, 0.3)), ns.Sphere(0.35, transform=ns.translate(0.15, 0.2, -0.2)), world,
                           ns.translate(-0.4, -0.7, 1.2) * ns.rotate(20, 0, 0), material)

    def build():
        world = ns.World()
        ns.Box(P(-1.0, -1.05, 0.0), P(1.0, -1.0, 2.0), world, material=ns.Lambert(white))                      # floor
        ns.Box(P(-1.0, -1.0, 2.0), P(1.0, 1.0, 2.05), world, material=ns.Lambert(white))                       # back wall
        ns.Box(P(-0.6, 0.98, 0.4), P(0.6, 0.999, 1.6), world, material=ns.UniformSurfaceEmitter(ns.ConstantSF(1.0), 6.0))
        cup(ns.Lambert(red), world)                                                                            # the same solid twice
        cup(ns.Lambert(blue), world)
        ns.Union(ns.Box(P(0.2, -1.0, 0.7), P(0.7, -0.6, 1.1)), ns.Cylinder(0.15, 0.7, transform=ns.translate(0.45, -1.0, 0.9) * ns.rotate(0, -90, 0)),
                 world, material=ns.Lambert(white))                                                            # its box stands in the floor's top plane
        ns.Intersect(ns.Sphere(0.3), ns.Box(P(-0.3, -0.3, 0.0), P(0.3, 0.3, 0.3)), world, ns.translate(0.5, 0.2, 1.5), ns.Lambert(blue))
        return world

    frames = {}
    for off in (False, True):
        if off:
            monkeypatch.setenv("RSX_NO_PKT_CLUSTERS", "1")
        world = build()
        assert len(world.primitives) == 7
        cam, pipe = scenes.lambert_camera(ns, world, (72, 60), 6, 5, (0.05, 3, 40))
        cam.frame_sampler = ns.RectFrameSampler2D()
        cam.render_engine = ns.HipEngine(rng="philox", seed=321)
        cam.observe()
        frames[off] = (pipe.frame.mean.copy(), pipe.frame.variance.copy(), cam.stats["rays"])
        if not off:
            w, h = 72, 60
            keep = []
            sl = list(cam._slice_spectrum())[0]
            desc = cam.render_desc(world, None, sl, cam.render_engine, keep, rect=(0, 0, w, h))
            om, ov, n_rays = orc.render_pinhole(world.flatten(), desc, threads=orc.max_threads())
            assert eq(pipe.frame.mean, om.reshape(h, w, sl.bins).transpose(1, 0, 2)) and eq(pipe.frame.variance, ov.reshape(h, w, sl.bins).transpose(1, 0, 2))
            assert cam.stats["rays"] == n_rays
            assert (pipe.frame.mean > 0).mean() > 0.2
    assert eq(frames[False][0], frames[True][0]) and eq(frames[False][1], frames[True][1]) and frames[False][2] == frames[True][2]


def test_frames_dielectric_against_oracle(orc, ns):
    """Dielectric on the device (RSX_MAT_DIELECTRIC): refraction / reflection choice, total internal reflection, transmission_only,
    per-slice Sellmeier index, Beer-Lambert attenuation. The oracle is pinned bit for bit to the reference by fixture F14. Clear and
    tinted glass alike: frames and ray statistics identical (0 ulp) — the attenuation transmission ** length goes through the
    portable pow that oracle, device and host restate operation for operation (the reference's libm pow differs from it by at most
    one unit in the last place per attenuated segment, tests/test_oracle_golden.py::test_portable_pow)."""
    for clear in (True, False):
        world, prims = scenes.build_glass(ns, unit_transmission=clear)
        cam, pipe = scenes.glass_camera(ns, world, (96, 72), 6, 6, 3, (0.01, 3, 500) if clear else (0.1, 2, 20))
        cam.frame_sampler = ns.RectFrameSampler2D()
        cam.render_engine = ns.HipEngine(rng="philox", seed=123)
        cam.observe()
        w, h, bins = 96, 72, 6
        ref_m, ref_v = np.zeros((w, h, bins)), np.zeros((w, h, bins))
        ref_rays = 0
        for sl in cam._slice_spectrum():
            keep = []
            desc = cam.render_desc(world, None, sl, cam.render_engine, keep, rect=(0, 0, w, h))
            om, ov, n_rays = orc.render_pinhole(world.flatten(), desc, threads=orc.max_threads())
            ref_rays += n_rays
            ref_m[:, :, sl.offset:sl.offset + sl.bins] = om.reshape(h, w, sl.bins).transpose(1, 0, 2)
            ref_v[:, :, sl.offset:sl.offset + sl.bins] = ov.reshape(h, w, sl.bins).transpose(1, 0, 2)
        assert cam.stats["rays"] == ref_rays
        assert eq(pipe.frame.mean, ref_m) and eq(pipe.frame.variance, ref_v), clear
        assert (pipe.frame.mean > 0).mean() > 0.3
    # the same tinted scene through the host-callback path: the Python restatement of the pow gives the device's frame
    world, prims = scenes.build_glass(ns, unit_transmission=False)
    cam2, pipe2 = scenes.glass_camera(ns, world, (96, 72), 6, 6, 3, (0.1, 2, 20))
    cam2.pixels = (24, 18)
    frames = []
    for host in (False, True):
        cam3, pipe3 = scenes.glass_camera(ns, world, (24, 18), 2, 6, 3, (0.1, 2, 20))
        cam3.frame_sampler = ns.RectFrameSampler2D()
        cam3.render_engine = ns.HipEngine(rng="philox", seed=9, host_materials=host)
        cam3.observe()
        frames.append((pipe3.frame.mean.copy(), pipe3.frame.variance.copy()))
        cam3.parent = None
    cam2.parent = None
    assert eq(frames[0][0], frames[1][0]) and eq(frames[0][1], frames[1][1])


def test_device_known_answers(orc, ns, golden):
    """The reference's golden vectors against the DEVICE functions directly (rsx_selftest_*): box slabs (F2), camera rays (F8),
    Welford states and the combine law (F9); the portable pow / sin / cos / asin of the path kernels against the oracle's."""
    import ctypes as C
    from source_amd import _lib
    from source_amd.device import get_context
    L, ctx = _lib.lib(), get_context()
    # F2: BoundingBox3D.intersect
    g = golden("f02_aabb")
    res = np.zeros((len(g["lower"]), 3))
    bad = C.c_uint64(1)
    arrs = [np.ascontiguousarray(g[k]) for k in ("lower", "upper", "origin", "direction")]
    _lib.check(L.rsx_selftest_aabb(ctx.handle, len(res), *(_lib.ptr(a) for a in arrs), _lib.ptr(res), C.byref(bad)))
    assert bad.value == 0 and eq(res, g["result"])
    # F8: pinhole rays in camera space (identity to_root, as the fixture records them) and through a camera transform (oracle)
    g = golden("f08_camera")
    world = ns.World()
    cam = ns.PinholeCamera((48, 32), fov=52.0, parent=world, transform=ns.translate(0.3, -0.2, 1.0) * ns.rotate(20, 10, 5))
    tasks = np.array([(0, 0), (47, 31), (13, 7), (24, 16), (5, 30)], dtype=np.int32)
    u = np.ascontiguousarray(g["uniforms"])
    for identity in (True, False):
        desc = _lib.RenderDesc()
        desc.camera = cam.device_camera()
        if identity:
            for i, v in enumerate(ns.AffineMatrix3D().m):
                desc.camera.to_root[i] = v
        desc.tasks, desc.n_tasks, desc.spp, desc.uniforms, desc.rng_mode = _lib.ptr(tasks), 5, 16, _lib.ptr(u), _lib.RNG_STREAM
        rays = np.zeros((80, 7))
        _lib.check(L.rsx_selftest_camera_rays(ctx.handle, C.byref(desc), _lib.ptr(rays)))
        assert eq(rays, g["rows"][:, 2:] if identity else orc.pinhole_rays(desc))
    # F9: Welford states after every prefix of the reference's sample sequences, and combine_samples
    g = golden("f09_stats")
    x, states = np.ascontiguousarray(g["x"]), g["states"]
    for k in range(1, x.shape[1] + 1):
        m, v = np.zeros(len(x)), np.zeros(len(x))
        _lib.check(L.rsx_selftest_welford(ctx.handle, len(x), k, _lib.ptr(np.ascontiguousarray(x[:, :k])), _lib.ptr(m), _lib.ptr(v)))
        assert eq(m, states[:, k - 1, 0]) and eq(v, states[:, k - 1, 1]), k
    n = len(g["ma"])
    dev = [ctx.alloc(8 * n) for _ in range(6)]
    host = [np.ascontiguousarray(g["ma"]), np.ascontiguousarray(g["va"]), g["na"].astype(np.int32), np.ascontiguousarray(g["mb"]),
            np.ascontiguousarray(g["vb"]), g["nb"].astype(np.int32)]
    for p, a in zip(dev, host):
        ctx.upload(p, a)
    _lib.check(L.rsx_frame_combine_dev(ctx.handle, n, *dev))
    out = [np.zeros(n), np.zeros(n), np.zeros(n, dtype=np.int32)]
    for p, a in zip(dev[:3], out):
        ctx.download(a, p)
    for p in dev:
        ctx.free(p)
    assert eq(np.stack([out[0], out[1], out[2].astype(float)], axis=1), g["comb"])
    # portable math: device == oracle, bit for bit
    rng = np.random.RandomState(4)
    a = np.concatenate([rng.uniform(0, 1, 20000) ** rng.choice([1, 3, 8], 20000), 2.0 ** rng.randint(-1060, 1000, 4000) * rng.uniform(1, 2, 4000)])
    b = np.concatenate([rng.uniform(0, 50, 20000) ** rng.choice([1, 2], 20000) - 5, rng.uniform(-2, 2, 4000)])
    a = np.ascontiguousarray(np.where(a > 0, a, 0.5))
    o0, o1 = np.zeros(len(a)), np.zeros(len(a))
    _lib.check(L.rsx_selftest_math(ctx.handle, 0, len(a), _lib.ptr(a), _lib.ptr(b), _lib.ptr(o0), None))
    assert eq(o0, orc.portable_pow(a, b))
    phi = np.ascontiguousarray(rng.uniform(0, 2 * np.pi, 20000))
    _lib.check(L.rsx_selftest_math(ctx.handle, 1, len(phi), _lib.ptr(phi), None, _lib.ptr(o0), _lib.ptr(o1)))
    sn, cs = orc.portable_sincos(phi)
    assert eq(o0[:len(phi)], sn) and eq(o1[:len(phi)], cs) and np.abs(sn - np.sin(phi)).max() < 4e-16
    t = np.ascontiguousarray(np.concatenate([rng.uniform(0, 1, 20000), [0.0, 0.5, 0.975, 1.0, 1e-9]]))
    _lib.check(L.rsx_selftest_math(ctx.handle, 2, len(t), _lib.ptr(t), None, _lib.ptr(o0), None))
    assert eq(o0[:len(t)], orc.portable_asin(t))


def test_frames_importance_sampling_against_oracle(orc, ns):
    """Multiple importance sampling on the device: the oracle — pinned bit for bit to the reference's frames and
    to ImportanceManager.sample()/pdf() vectors by fixture F15 — and the device draw the same Philox numbers; frames and ray counts
    must be identical for the reference's default weight, for a light-dominated mixture and with the camera's defaults."""
    world, prims = scenes.build_lambert(ns)
    prims[5].material.importance = 3.0
    ns.Sphere(0.12, world, ns.translate(-0.6, 0.5, 1.2), ns.UniformSurfaceEmitter(ns.ConstantSF(1.0), 4.0))
    for weight, ext in ((0.25, (0.01, 3, 500)), (0.9, (0.1, 2, 12)), (0.0, (0.1, 2, 12)), (1.0, (0.1, 2, 12))):
        w, h, bins = 80, 64, 5
        cam, pipe = scenes.lambert_camera(ns, world, (w, h), 6, bins, ext)
        cam.ray_importance_sampling = True
        cam.ray_important_path_weight = weight
        cam.frame_sampler = ns.RectFrameSampler2D()
        cam.render_engine = ns.HipEngine(rng="philox", seed=31)
        cam.observe()
        keep = []
        desc = cam.render_desc(world, None, cam._slice_spectrum()[0], cam.render_engine, keep, rect=(0, 0, w, h))
        assert desc.n_important == 3
        om, ov, n_rays = orc.render_pinhole(world.flatten(), desc, threads=orc.max_threads())
        assert eq(pipe.frame.mean, om.reshape(h, w, bins).transpose(1, 0, 2)) and eq(pipe.frame.variance, ov.reshape(h, w, bins).transpose(1, 0, 2)), weight
        assert cam.stats["rays"] == n_rays


def test_rgb_pipeline_on_device(orc, ns, golden):
    """RGBPipeline2D through rsx_render_pinhole_xyz (k_accumulate_xyz): with the stream engine the XYZ frame, the spectral frame
    rendered next to it from the same rays, the accumulate pass over the adaptive sampler's tasks and the sampler's task lists are
    bit-identical to the reference's (fixture F16); on a path-traced scene in Philox mode the per-task XYZ results equal the oracle's."""
    import random as pyrandom
    from source_amd.core import random as rsrandom
    g = golden("f16_rgb")
    world, mesh, box = scenes.build_c2(ns, n=48, smoothing=True, with_normals=True)
    rgb, spectral = ns.RGBPipeline2D(), ns.SpectralPowerPipeline2D()
    cam = ns.PinholeCamera((20, 16), fov=45, sensitivity=2.5, parent=world, pipelines=[rgb, spectral], frame_sampler=ns.FullFrameSampler2D(),
                           transform=ns.translate(0, 0.16, -0.4) * ns.rotate(0, -12, 0))
    cam.pixel_samples, cam.spectral_bins, cam.spectral_rays, cam.quiet = 3, 9, 3, True
    cam.min_wavelength, cam.max_wavelength = 400.0, 700.0
    cam.render_engine = ns.SerialEngine()
    pyrandom.seed(71); rsrandom.seed(71); cam.observe()
    f = rgb.xyz_frame
    assert eq(f.mean, g["xyz_mean"]) and eq(f.variance, g["xyz_var"]) and eq(f.samples, g["xyz_n"])
    assert eq(spectral.frame.mean, g["spec_mean"])
    sampler = ns.RGBAdaptiveSampler2D(rgb, ratio=2, fraction=0.3, min_samples=5, cutoff=0.05)
    pyrandom.seed(72)
    assert np.array_equal(np.array(sampler.generate_tasks((20, 16))), g["tasks1"])
    cam.frame_sampler = sampler
    pyrandom.seed(73); rsrandom.seed(73); cam.observe()
    assert eq(f.mean, g["xyz_mean2"]) and eq(f.variance, g["xyz_var2"]) and eq(f.samples, g["xyz_n2"])
    pyrandom.seed(74)
    assert np.array_equal(np.array(sampler.generate_tasks((20, 16))), g["tasks2"])
    assert rgb.rgb_frame.shape == (20, 16, 3) and (rgb.rgb_frame >= 0).all() and (rgb.rgb_frame <= 1).all()
    # path-traced scene, Philox: the XYZ kernel replays the term lists per bin like the spectral one
    world2, prims = scenes.build_lambert(ns)
    rgb2 = ns.RGBPipeline2D()
    cam2, _ = scenes.lambert_camera(ns, world2, (64, 48), 4, 6, (0.1, 2, 12))
    cam2.pipelines = [rgb2]
    cam2.spectral_rays = 2
    cam2.frame_sampler = ns.RectFrameSampler2D()
    cam2.render_engine = ns.HipEngine(rng="philox", seed=9)
    cam2.observe()
    want_m, want_v = np.zeros((64, 48, 3)), np.zeros((64, 48, 3))
    for slice_id, sl in enumerate(cam2._slice_spectrum()):
        keep = []
        desc = cam2.render_desc(world2, None, sl, cam2.render_engine, keep, rect=(0, 0, 64, 48))
        om, ov, _ = orc.render_pinhole_xyz(world2.flatten(), desc, rgb2._resampled[slice_id], rgb2._deltas[slice_id], threads=orc.max_threads())
        want_m += om.reshape(48, 64, 3).transpose(1, 0, 2)
        want_v += ov.reshape(48, 64, 3).transpose(1, 0, 2)
    assert eq(rgb2.xyz_frame.mean, want_m) and eq(rgb2.xyz_frame.variance, want_v) and (rgb2.xyz_frame.samples == 4).all()
    assert (rgb2.xyz_frame.mean[:, :, 1] > 0).mean() > 0.5


def test_lambert_furnace_full_size(ns):
    """Size-independent property at 1024 x 1024: inside a closed furnace — every surface either a unit-reflectivity Lambert wall or
    an emitter of radiance L — with roulette off (probability 0: normalisation exactly 1) every path ends on an emitter and carries
    L * prod(pdf * (1 / pdf)), so every pixel's mean is L within a few ulp per bounce and its variance vanishes, however long the path
    (paths here average tens of bounces and chain arena blocks)."""
    world = ns.World()
    P = ns.Point3D
    ns.Box(P(-1, -1, -1), P(1, 1, 1), world, material=ns.Lambert(ns.ConstantSF(1.0)))
    ns.Sphere(0.35, world, ns.translate(0.2, -0.3, 0.4), ns.UniformSurfaceEmitter(ns.ConstantSF(1.0), 3.0))
    ns.Sphere(0.25, world, ns.translate(-0.5, 0.4, 0.1), ns.Lambert(ns.ConstantSF(1.0)))
    cam, pipe = scenes.lambert_camera(ns, world, (1024, 1024), 2, 3, (0.0, 1, 30000))
    cam.transform = ns.translate(0, 0, -0.9)
    cam.frame_sampler = ns.RectFrameSampler2D()
    cam.render_engine = ns.HipEngine(rng="philox", seed=5)
    cam.observe()
    m, v = pipe.frame.mean, pipe.frame.variance
    # radiance pipeline: sample = L * projection weight (cos of the pixel's ray to the optical axis), which varies by < 1e-6 inside a pixel
    assert np.isfinite(m).all() and (m > 0).all()
    desc_cam = cam.device_camera()
    ix, iy = np.meshgrid(np.arange(1024), np.arange(1024), indexing="ij")
    x = desc_cam.image_start_x - desc_cam.image_delta * (ix + 0.5)
    y = desc_cam.image_start_y - desc_cam.image_delta * (iy + 0.5)
    w_centre = 1.0 / np.sqrt(x * x + y * y + 1.0)
    assert np.abs(m / (3.0 * w_centre[:, :, None]) - 1.0).max() < 2e-3          # jitter moves the weight inside the pixel
    assert np.abs(m[:, :, 0] - m[:, :, 2]).max() <= 1e-12 * 3.0               # the bins of a pixel see the same paths
    assert v.max() < 1e-5


def test_cornell_box_end_to_end(orc, ns):
    """BASELINE configs[0]'s scene the way demos/cornell_box.py drives it: Lambert walls, glass, an important light (multiple importance
    sampling), an RGB pipeline next to a spectral one, and RGBAdaptiveSampler2D deciding what each further pass renders. Pass 1 must
    equal the oracle (same Philox paths) in both pipelines; the adaptive passes must only add samples where the sampler asked."""
    world, prims = scenes.build_cornell(ns)
    rgb, spectral = ns.RGBPipeline2D(), ns.SpectralRadiancePipeline2D()
    cam, _ = scenes.cornell_camera(ns, world, (48, 48), 4, 6, pipelines=[rgb, spectral])
    cam.frame_sampler = ns.RGBAdaptiveSampler2D(rgb, ratio=4, fraction=0.3, min_samples=8, cutoff=0.01)
    cam.render_engine = ns.HipEngine(rng="philox", seed=3)
    cam.observe()
    # oracle for pass 1: the sampler's first task list is the full frame (x outer, y inner, shuffled) — order does not matter in Philox mode
    keep = []
    full = [(x, y) for x in range(48) for y in range(48)]
    desc = cam.render_desc(world, full, cam._slice_spectrum()[0], cam.render_engine, keep)
    om, ov, _ = orc.render_pinhole(world.flatten(), desc, threads=orc.max_threads())
    xm, xv, _ = orc.render_pinhole_xyz(world.flatten(), desc, rgb._resampled[0], rgb._deltas[0], threads=orc.max_threads())
    assert eq(spectral.frame.mean, om.reshape(48, 48, 6)) and eq(spectral.frame.variance, ov.reshape(48, 48, 6))
    assert eq(rgb.xyz_frame.mean, xm.reshape(48, 48, 3)) and eq(rgb.xyz_frame.variance, xv.reshape(48, 48, 3))
    assert (rgb.xyz_frame.samples == 4).all() and (rgb.xyz_frame.mean[:, :, 1] > 0).mean() > 0.4
    before = rgb.xyz_frame.samples.copy()
    for step in range(3):                                          # adaptive passes
        engine_offset = (step + 1) * 4
        cam.render_engine.sample_offset = engine_offset
        cam.observe()
    after = rgb.xyz_frame.samples
    assert (after >= before).all() and after.max() == 16 and (after % 4 == 0).all()
    assert after.min() >= 8                                        # min_samples reached everywhere after the second pass
    assert np.isfinite(rgb.xyz_frame.mean).all() and np.isfinite(rgb.rgb_frame).all()
    # render_complete (observer.pyx:265-309): False after a pass that rendered something, True once the sampler has nothing left to ask for
    assert cam.render_complete is False
    sampler = cam.frame_sampler
    sampler.cutoff, sampler.min_samples, sampler.ratio = 1.0, 1, 1000.0
    passes = 0
    while not cam.render_complete and passes < 50:
        cam.render_engine.sample_offset = (4 + passes) * 4
        cam.observe()
        passes += 1
    assert cam.render_complete is True and passes < 50


def test_prism_scene_against_oracle(orc, ns):
    """BASELINE configs[4]'s scene (dispersive prism: nested analytic CSG through the state-free evaluator and the two-pass path
    kernel, two Sellmeier glasses, importance sampling towards the prism, one-bin spectral slices): device frame = oracle frame
    (the oracle reproduces the reference's frame of this scene bit for bit, fixture F17), ray statistics included."""
    world, prims = scenes.build_prism(ns)
    w, h, bins = 96, 64, 8
    cam, pipe = scenes.prism_camera(ns, world, (w, h), 4, bins, bins)
    cam.frame_sampler = ns.RectFrameSampler2D()
    cam.render_engine = ns.HipEngine(rng="philox", seed=17)
    cam.observe()
    ref_m, ref_v, ref_rays = np.zeros((w, h, bins)), np.zeros((w, h, bins)), 0
    for sl in cam._slice_spectrum():
        keep = []
        desc = cam.render_desc(world, None, sl, cam.render_engine, keep, rect=(0, 0, w, h))
        om, ov, n_rays = orc.render_pinhole(world.flatten(), desc, threads=orc.max_threads())
        ref_rays += n_rays
        ref_m[:, :, sl.offset:sl.offset + sl.bins] = om.reshape(h, w, sl.bins).transpose(1, 0, 2)
        ref_v[:, :, sl.offset:sl.offset + sl.bins] = ov.reshape(h, w, sl.bins).transpose(1, 0, 2)
    assert eq(pipe.frame.mean, ref_m) and eq(pipe.frame.variance, ref_v)
    assert cam.stats["rays"] == ref_rays
    assert (pipe.frame.mean > 0).mean() > 0.1


def test_path_arena_grows_on_demand(ns):
    """Long paths: a closed furnace with a small emitter and roulette off makes paths hundreds of bounces long, far more terms than the
    arena a pass starts with (two blocks per ray). The pass is traced again with a larger arena (deterministic Philox paths) until it
    fits; the furnace property (every pixel = the emitter's radiance) shows that nothing was lost or merged twice on the way."""
    world = ns.World()
    P = ns.Point3D
    ns.Box(P(-1, -1, -1), P(1, 1, 1), world, material=ns.Lambert(ns.ConstantSF(1.0)))
    ns.Sphere(0.12, world, ns.translate(0.3, -0.2, 0.4), ns.UniformSurfaceEmitter(ns.ConstantSF(1.0), 2.0))
    cam, pipe = scenes.lambert_camera(ns, world, (256, 256), 4, 2, (0.0, 1, 30000))
    cam.transform = ns.translate(0, 0, -0.9)
    cam.frame_sampler = ns.RectFrameSampler2D()
    cam.render_engine = ns.HipEngine(rng="philox", seed=11)
    cam.observe()
    m = pipe.frame.mean
    assert cam.stats["rays"] > 50 * 256 * 256 * 4                 # paths really are long (mean > 50 segments)
    desc_cam = cam.device_camera()
    ix, iy = np.meshgrid(np.arange(256), np.arange(256), indexing="ij")
    x = desc_cam.image_start_x - desc_cam.image_delta * (ix + 0.5)
    y = desc_cam.image_start_y - desc_cam.image_delta * (iy + 0.5)
    w_centre = 1.0 / np.sqrt(x * x + y * y + 1.0)
    assert np.isfinite(m).all() and np.abs(m / (2.0 * w_centre[:, :, None]) - 1.0).max() < 5e-3
    assert (pipe.frame.samples == 4).all()


def test_pipelining_does_not_change_frames():
    """Render-pass pipelining (private lanes, longest-first unit order, XCD work lists) only changes which wave renders which unit:
    120 accumulating passes give the same frame digest with 1, 3 and 4 lanes (separate processes: the depth is read at rsx_init)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = []
    for depth in ("1", "3", "4"):
        env = dict(os.environ, RSX_PIPELINE=depth, KB_WARM="20", RSX_AUTO_BATCH="0")      # (every pass its own launch: that is what is under test)
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "kbench.py"), "100", "c2"], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        digests.append(json.loads(out.stdout.strip().splitlines()[-1])["digest"])
    assert len(set(digests)) == 1, digests



def test_host_callback_path_equals_device_path(ns):
    """The host-callback render path (source_amd/optical/hybrid.py: rays traced on the device wave by wave, every material's
    evaluate_surface / evaluate_volume called on the host) against the on-device path tracer on the same scenes and Philox counters:
    frames and ray statistics bit-identical. Diffuse room (Lambert, roulette, CSG solid, smooth mesh, volume emitter, null shell)
    and the Cornell box (multiple importance sampling, clear dielectrics)."""
    for build, camera in ((lambda: scenes.build_lambert(ns), lambda w: scenes.lambert_camera(ns, w, (40, 32), spp=3, bins=4, extinction=(0.2, 2, 9))),
                          (lambda: scenes.build_cornell(ns), lambda w: scenes.cornell_camera(ns, w, (28, 24), 2, 5))):
        frames = []
        for host in (False, True):
            world, prims = build()
            cam, pipe = camera(world)
            cam.frame_sampler = ns.RectFrameSampler2D()
            cam.render_engine = ns.HipEngine(rng="philox", seed=31, host_materials=host)
            cam.observe()
            cam.observe()                                                       # two accumulating passes
            frames.append((pipe.frame.mean.copy(), pipe.frame.variance.copy(), pipe.frame.samples.copy(), cam.stats["rays"]))
        assert eq(frames[0][0], frames[1][0]) and eq(frames[0][1], frames[1][1]) and eq(frames[0][2], frames[1][2])
        assert frames[0][3] == frames[1][3] and (frames[0][0] > 0).mean() > 0.2


def test_python_materials_in_forked_workers_with_the_device_in_the_loop():
    """A scene with a user-written material splits its primary rays over forked worker processes (hybrid.run_block) while the parent —
    HIP initialised, scene resident — answers their ray waves on the device: the frame equals the all-device render and the
    one-process host render bit for bit, per_node_materials=True (every material through the plugin API) included; the parent's
    device stays usable afterwards (a second device render of another scene). The case (tests/forked_workers_case.py) runs in a process
    of its own: forked children of THIS process inherit whatever the suite has mapped by then — after the full-size frames of the tests
    above, every fork and every copy-on-write fault of 4 busy children cost so much that the case took 510 s of the suite's 690
    (round 5) against seconds alone; what is tested is the scheme, not fork() of a 20 GB test runner."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "forked_workers_case.py")], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "forked workers OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_user_written_material_through_observe(orc, ns):
    """A Material subclass the library has never seen renders through observe() (SURVEY.md §8b plug-point #3: the material plugin API
    stays untouched). (1) A user re-implementation of Lambert's shading goes through the host-callback path and reproduces the
    device-lowered Lambert's frame bit for bit (and so, by test_frames_lambert_against_oracle, the oracle's). (2) A material that
    traces two daughters per hit and adds a volume term renders, finite and non-trivial. (3) The single-ray API: Ray.trace()."""
    from source_amd.optical.material import Material, has_device_lowering, hemisphere_cosine_pdf

    class MyLambert(ns.Lambert):
        def evaluate_shading(self, world, ray, s_in, s_out, w_refl, w_trans, back_face, w2s, s2w, intersection):
            pdf = hemisphere_cosine_pdf(s_out)
            if pdf == 0.0:
                return ray.new_spectrum()
            spectrum = ray.spawn_daughter(w_refl, s_out.transform(s2w)).trace(world)
            spectrum.mul_array(self.reflectivity.sample(spectrum.min_wavelength, spectrum.max_wavelength, spectrum.bins))
            spectrum.mul_scalar(pdf)
            return spectrum

    class Splitter(Material):
        def evaluate_surface(self, world, ray, primitive, hit_point, exiting, inside_point, outside_point, normal, w2p, p2w, intersection):
            n = normal.transform_with_inverse(w2p).normalise()
            d = ray.direction
            k = 2 * (d.x * n.x + d.y * n.y + d.z * n.z)
            a = ray.spawn_daughter((inside_point if exiting else outside_point).transform(p2w), ns.Vector3D(d.x - k * n.x, d.y - k * n.y, d.z - k * n.z)).trace(world)
            b = ray.spawn_daughter((outside_point if exiting else inside_point).transform(p2w), d).trace(world)
            a.mul_scalar(0.5)
            a.mad_scalar(0.5, b.samples)
            return a

        def evaluate_volume(self, spectrum, world, ray, primitive, start_point, end_point, w2p, p2w):
            spectrum.samples[:] = spectrum.samples + 0.05 * start_point.vector_to(end_point).length
            return spectrum

    assert not has_device_lowering(MyLambert()) and has_device_lowering(ns.Lambert())
    frames = []
    for lambert in (ns.Lambert, MyLambert):
        world, prims = scenes.build_lambert(ns)
        for p in prims:
            if isinstance(p.material, ns.Lambert):
                p.material = lambert(p.material.reflectivity)
        cam, pipe = scenes.lambert_camera(ns, world, (36, 28), spp=3, bins=4, extinction=(0.2, 2, 9))
        cam.frame_sampler = ns.RectFrameSampler2D()
        cam.render_engine = ns.HipEngine(rng="philox", seed=77)
        cam.observe()
        frames.append((pipe.frame.mean.copy(), pipe.frame.variance.copy(), cam.stats["rays"]))
    assert eq(frames[0][0], frames[1][0]) and eq(frames[0][1], frames[1][1]) and frames[0][2] == frames[1][2]
    # the lowered frame is the oracle's frame (same check as test_frames_lambert_against_oracle, restated for this camera)
    keep = []
    desc = cam.render_desc(world, None, cam._slice_spectrum()[0], cam.render_engine, keep, rect=(0, 0, 36, 28))
    for p, lam in zip(world.primitives, [type(q.material) for q in world.primitives]):
        if lam is MyLambert:
            p.material = ns.Lambert(p.material.reflectivity)
    desc = cam.render_desc(world, None, cam._slice_spectrum()[0], cam.render_engine, keep, rect=(0, 0, 36, 28))
    om, ov, _ = orc.render_pinhole(world.flatten(), desc, threads=orc.max_threads())
    assert eq(frames[1][0], om.reshape(28, 36, 4).transpose(1, 0, 2)) and eq(frames[1][1], ov.reshape(28, 36, 4).transpose(1, 0, 2))
    # (2) two daughters per hit + a volume term
    world, prims = scenes.build_lambert(ns, with_volume=False, csg=False)
    ns.Sphere(0.25, world, ns.translate(0.35, 0.1, 0.9), Splitter())
    cam, pipe = scenes.lambert_camera(ns, world, (32, 24), spp=4, bins=3, extinction=(0.2, 2, 8))
    cam.render_engine = ns.HipEngine(rng="philox", seed=5)
    cam.observe()
    assert np.isfinite(pipe.frame.mean).all() and (pipe.frame.samples == 4).all() and (pipe.frame.mean > 0).mean() > 0.3
    assert cam.stats["rays"] > 32 * 24 * 4 * 2
    # (3) single rays through the same plugin API (host MT stream for the stochastic choices)
    from source_amd.core import random as rsrandom
    rsrandom.seed(3)
    ray = ns.Ray(ns.Point3D(0, 0, -1.9), ns.Vector3D(0.05, -0.1, 1).normalise(), min_wavelength=400, max_wavelength=700, bins=6,
                 extinction_prob=0.2, extinction_min_depth=2, max_depth=12)
    total = ray.sample(world, 24)
    assert total.samples.shape == (6,) and np.isfinite(total.samples).all() and total.samples.max() > 0 and ray.ray_count > 1


def test_fused_welford_form_gives_the_same_frames(ns):
    """The opt-in fused form (RSX_FUSE=1: per-pixel Welford and frame merge inside the trace kernel, sample records in per-wave
    rings instead of one HBM buffer) against the two-kernel form: frames bit-identical, for whole-pixel units at 64, 16 and 1 samples
    per pixel, a frame whose size is not a multiple of the 8x8 unit tiles, the power pipeline, and two accumulating passes."""
    import os
    import subprocess
    import sys
    code = """
import hashlib, sys
import numpy as np
sys.path.insert(0, %r)
from source_amd import api as ns, scenes
world = scenes.build_c3(ns, n=24)[0]
out = []
for (nx, ny, spp, power) in ((72, 44, 64, False), (100, 61, 16, True), (64, 64, 1, False)):
    pipe = ns.SpectralPowerPipeline2D() if power else ns.SpectralRadiancePipeline2D()
    cam, _ = scenes.c3_camera(ns, world, (nx, ny), spp=spp, bins=7)
    cam.pipelines = [pipe]
    cam.sensitivity = 2.5
    cam.frame_sampler = ns.RectFrameSampler2D()
    cam.render_engine = ns.HipEngine(rng="philox", seed=8)
    cam.observe(); cam.observe()
    h = hashlib.sha256()
    for a in (pipe.frame.mean, pipe.frame.variance, pipe.frame.samples):
        h.update(np.ascontiguousarray(a).tobytes())
    assert (pipe.frame.samples == 2 * spp).all() and pipe.frame.mean.max() > 0
    out.append(h.hexdigest())
print(" ".join(out))
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = []
    for fuse in ("0", "1"):
        env = dict(os.environ, RSX_FUSE=fuse, RSX_PIPELINE="1")       # un-pipelined: the fused form only serves passes that run alone
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        digests.append(r.stdout.strip().splitlines()[-1])
    assert digests[0] == digests[1] and len(digests[0].split()) == 3


def test_lifted_limits_many_tables_and_nested_volumes(orc, ns):
    """Two former RSX_EUNSUPPORTED limits now take a slower correct path. (1) A 512-bin single-slice render with 20 materials: 82 KB of
    spectral tables, more than the accumulate kernel's LDS — the tables are read from global memory; frame = oracle. (2) Six nested
    volume emitters around the camera: more overlapping volumes than the path kernel keeps in registers; the extra terms come from
    re-walking world.contains(); frame = oracle."""
    world = ns.World()
    rng = np.random.RandomState(5)
    for k in range(20):
        sf = ns.InterpolatedSF([300, 400 + 15 * k, 800], np.array([0.1 + 0.04 * k, 1.0, 0.3]))
        ns.Sphere(0.12, world, ns.translate(-1.2 + 0.125 * k, 0.3 * np.sin(k), 2.0 + 0.05 * k), ns.UniformSurfaceEmitter(sf, 1.0 + 0.1 * k))
    pipe = ns.SpectralRadiancePipeline2D()
    cam = ns.PinholeCamera((96, 48), fov=60, parent=world, pipelines=[pipe], frame_sampler=ns.RectFrameSampler2D())
    cam.pixel_samples, cam.spectral_bins, cam.spectral_rays, cam.quiet = 8, 512, 1, True
    cam.render_engine = ns.HipEngine(rng="philox", seed=2)
    cam.observe()
    keep = []
    desc = cam.render_desc(world, None, cam._slice_spectrum()[0], cam.render_engine, keep, rect=(0, 0, 96, 48))
    assert desc.n_tables == 20 and desc.n_tables * 512 * 8 > 60 * 1024
    om, ov, _ = orc.render_pinhole(world.flatten(), desc, threads=orc.max_threads())
    assert eq(pipe.frame.mean, om.reshape(48, 96, 512).transpose(1, 0, 2)) and eq(pipe.frame.variance, ov.reshape(48, 96, 512).transpose(1, 0, 2))
    assert (pipe.frame.mean.max(axis=2) > 0).mean() > 0.05
    # (2) nested volumes
    world = ns.World()
    for k in range(6):
        ns.Sphere(1.0 + 0.3 * k, world, ns.translate(0.02 * k, 0, 0), ns.UniformVolumeEmitter(ns.ConstantSF(0.2 + 0.1 * k), 0.5 + 0.25 * k))
    ns.Sphere(4.0, world, material=ns.AbsorbingSurface())
    pipe = ns.SpectralRadiancePipeline2D()
    cam = ns.PinholeCamera((48, 32), fov=70, parent=world, pipelines=[pipe], frame_sampler=ns.RectFrameSampler2D())
    cam.pixel_samples, cam.spectral_bins, cam.quiet = 3, 4, True
    cam.render_engine = ns.HipEngine(rng="philox", seed=4)
    cam.observe()
    keep = []
    desc = cam.render_desc(world, None, cam._slice_spectrum()[0], cam.render_engine, keep, rect=(0, 0, 48, 32))
    om, ov, rays = orc.render_pinhole(world.flatten(), desc, threads=orc.max_threads())
    assert eq(pipe.frame.mean, om.reshape(32, 48, 4).transpose(1, 0, 2)) and eq(pipe.frame.variance, ov.reshape(32, 48, 4).transpose(1, 0, 2))
    assert cam.stats["rays"] == rays and pipe.frame.mean.min() > 0


def test_random_analytic_worlds_wide_slots_and_cull(orc, ns):
    """The world level of the traversal kernels on worlds made to stress it (tools/stress_world.py, reduced): 3 .. 28 grid-snapped
    spheres / boxes / cylinders — coincident faces, a quarter of them huge so that they sit in most world leaves (wide primitives),
    sometimes a mesh (subtrees that cannot be culled) — hit by random, grid-aligned and axis-parallel rays (two wide slots) and
    path traced with scattering, refracting and emitting materials, half the frames with importance sampling (eight wide slots,
    cull by the nearest wide answer): ids, distances, geometry and frames equal the oracle bit for bit."""
    rng = np.random.RandomState(97531)
    grid = [-1.0, -0.5, -0.25, 0.0, 0.25, 0.5, 1.0]
    P = ns.Point3D

    def snap(scale=1.0):
        return scale * (float(rng.choice(grid)) if rng.rand() < 0.6 else float(rng.uniform(-1.0, 1.0)))

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
            if huge and rng.rand() < 0.5:
                ax = rng.randint(3)
                lo[ax], hi[ax] = -0.05, 0.0
            return ns.Box(P(*lo), P(*hi), world, t)
        return ns.Cylinder(s * float(rng.choice([0.25, 0.5])), s * float(rng.choice([0.5, 1.0])), world, t)

    # (with the enclosing emitter of the path-traced half: worlds of at most eight analytic primitives take every primitive into a
    # wide slot — single-leaf ones too — and find the boxes' roots per lane; larger or mixed worlds keep the >= 2-leaf rule)
    for wi, n_prims in enumerate([3, 5, 7, 6, 7, 8, 9, 12, 20, 28]):
        world = ns.World()
        for _ in range(n_prims):
            primitive(world, huge=rng.rand() < 0.25)
        if wi % 3 == 2:
            v, t = scenes.displaced_sphere(4, radius=0.4)
            ns.Mesh(v, t, parent=world, transform=ns.translate(snap(), snap(), snap()))
        scene = world.build_accelerator()
        n = 40000
        o = rng.uniform(-3, 3, size=(n, 3))
        d = rng.normal(size=(n, 3))
        k8 = n // 8
        o[:k8] = rng.choice(grid + [2.0, -2.0, 3.0], size=(k8, 3))
        d[k8:2 * k8] = 0.0
        d[np.arange(k8, 2 * k8), rng.randint(3, size=k8)] = rng.choice([-1.0, 1.0], size=k8)
        d /= np.linalg.norm(d, axis=1)[:, None]
        m = np.where(rng.rand(n) < 0.2, rng.uniform(0.1, 4.0, size=n), np.inf)
        assert_hits_equal(scene.hit_batch(o, d, m, geometry=True), orc.hit_batch(world.flatten(), o, d, m, geometry=True))
        mats = [ns.Lambert(ns.ConstantSF(0.8)), ns.Dielectric(ns.ConstantSF(1.5), ns.ConstantSF(1.0)), ns.UniformVolumeEmitter(ns.ConstantSF(1.0), 0.5),
                ns.Lambert(ns.ConstantSF(0.5)), ns.NullMaterial(), ns.UniformSurfaceEmitter(ns.ConstantSF(1.0), 0.7)]
        for prim in list(world._primitives):
            prim.material = mats[rng.randint(len(mats))]
            if rng.rand() < 0.2:
                prim.material.importance = float(rng.choice([1.0, 4.0]))
        ns.Box(P(-6, -6, -6), P(6, 6, 6), world, material=ns.UniformSurfaceEmitter(ns.ConstantSF(1.0), 1.0))
        pipe = ns.SpectralRadiancePipeline2D()
        cam = ns.PinholeCamera((64, 64), fov=60, parent=world, pipelines=[pipe], frame_sampler=ns.RectFrameSampler2D(), transform=ns.translate(0.2, 0.1, -4.5))
        cam.pixel_samples, cam.spectral_bins, cam.spectral_rays, cam.quiet = 4, 3, 1, True
        cam.ray_extinction_prob, cam.ray_extinction_min_depth, cam.ray_max_depth = 0.05, 2, 60
        cam.ray_importance_sampling = bool(wi % 2)
        cam.render_engine = ns.HipEngine(rng="philox", seed=wi)
        cam.observe()
        keep = []
        desc = cam.render_desc(world, None, cam._slice_spectrum()[0], cam.render_engine, keep, rect=(0, 0, 64, 64))
        om, ov, rays = orc.render_pinhole(world.flatten(), desc, threads=orc.max_threads())
        assert cam.stats["rays"] == rays
        assert eq(np.array(pipe.frame.mean), om.reshape(64, 64, 3).transpose(1, 0, 2))


def test_deferred_slice_checks_give_the_same_frames(ns, monkeypatch):
    """A HipEngine render of several spectral slices lets the slices' path passes overlap on the device (rsx_defer_path_checks) and
    collects their end-of-pass checks at the end of observe(). Frames and ray counts equal the slice-by-slice render bit for bit —
    (1) on the glass scene with three slices, (2) on a furnace whose paths outgrow the term arena, so that every deferred pass
    fails its check, leaves the frame untouched and is issued again through the ordinary retry path."""
    from source_amd.optical import observer as obs

    def render(defer, build):
        world, cam, pipe = build()
        cam.frame_sampler = ns.RectFrameSampler2D()
        cam.render_engine = ns.HipEngine(rng="philox", seed=3)
        if not defer:
            monkeypatch.setattr(obs._ObserverBase, "_begin_deferred_slices", lambda self, n: False)
        else:
            monkeypatch.undo()
        for _ in range(2):
            cam.observe()
        return np.array(pipe.frame.mean), np.array(pipe.frame.variance), np.array(pipe.frame.samples), cam.stats["rays"]

    def glass():
        world = scenes.build_glass(ns)[0]
        cam, pipe = scenes.glass_camera(ns, world, (96, 64), 4, 6, 3, (0.01, 3, 200))
        return world, cam, pipe

    def furnace():
        world = ns.World()
        P = ns.Point3D
        ns.Box(P(-1, -1, -1), P(1, 1, 1), world, material=ns.Lambert(ns.ConstantSF(1.0)))
        ns.Sphere(0.12, world, ns.translate(0.3, -0.2, 0.4), ns.UniformSurfaceEmitter(ns.ConstantSF(1.0), 2.0))
        cam, pipe = scenes.lambert_camera(ns, world, (64, 64), 4, 4, (0.0, 1, 30000))
        cam.spectral_rays = 2
        cam.transform = ns.translate(0, 0, -0.9)
        return world, cam, pipe

    for build in (glass, furnace):
        a = render(True, build)
        b = render(False, build)
        assert a[3] == b[3] and a[3] > 0
        assert eq(a[0], b[0]) and eq(a[1], b[1]) and eq(a[2], b[2])


@pytest.mark.parametrize("workload,sharding", [("c2", "tile"), ("c2", "sample"), ("c5s", "slice")])
def test_bench_distributed_paths_start_on_rccl(workload, sharding):
    """bench.py under torch.distributed.run with ONE rank, for each sharding: dlopen(librccl), ncclCommInitRank, the untimed first
    exchange and the timed one (rsx_allgather_frame / rsx_allreduce_frame / rsx_allgather_bins) all run from librsx on this box —
    what the driver's 2 / 4 / 8-GPU runs start from. (More than one rank needs more than one GPU: the CPU gloo tests cover the shard
    logic, bench.py checks the exchanged frame against a one-GPU render in the run itself.)"""
    import json
    import subprocess
    import sys
    port = 29600 + (os.getpid() + hash(sharding)) % 300
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-pmc", "--no-cpu-baseline", "--workload", workload,
           "--sharding", sharding]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    cfg = d["config"]
    assert cfg["sharding"] == sharding and cfg["collective"].startswith("RCCL from librsx"), cfg["collective"]
    assert cfg["rccl_ranks"] == 1 and d["n_gpus"] == 1 and d["value"] > 0
    assert d["scaling"] == ("weak" if sharding == "sample" else "strong")
    if sharding == "slice":
        assert cfg["slice_bounds"] == [0, 16]


def test_bench_self_launch_two_ranks_one_gpu():
    """`python bench.py --gpus 2 --collective host` with no launcher around it: bench.py starts its two ranks itself (torch.distributed.run
    on 127.0.0.1 at a free port), both render their tile on this one GPU, the frame is exchanged over the host route and checked against
    a one-GPU render inside the run; rank 0's line comes back with n_gpus = 2."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-pmc", "--no-cpu-baseline",
                        "--workload", "c2", "--collective", "host"], cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["sharding"] == "tile"
    assert d["config"].get("frame_digest_equals_single_gpu") is True


def test_csg_trees_of_any_depth_and_size_vs_oracle(orc, ns):
    """CSG operand trees beyond the former limits (5 nested levels, 16 nodes; csg.pyx:132-234 recurses without a bound): a 12-level
    Union chain, a 14-level mixed chain, and random trees of 33 - 70 nodes — first hits with full geometry, next_intersection
    sequences, contains(), and a path-traced frame, all against the oracle's recursion, bit for bit. (The stream merge runs as one
    loop over an explicit frame stack; trees of more than 16 nodes keep their node states in the scene's arena.)"""
    rng = np.random.RandomState(77)
    P = ns.Point3D

    def leaf(k):
        t = ns.translate(0.18 * np.cos(0.9 * k), 0.18 * np.sin(1.3 * k), 0.07 * ((k % 5) - 2)) * ns.rotate(17.0 * k, 11.0 * k, 5.0 * k)
        kind = k % 3
        if kind == 0:
            return ns.Sphere(0.25 + 0.02 * (k % 4), transform=t)
        if kind == 1:
            return ns.Box(P(-0.2, -0.25, -0.15), P(0.25, 0.2, 0.3), transform=t)
        return ns.Cylinder(0.2, 0.5, transform=t)

    def chain(n, ops):
        node = leaf(0)
        for k in range(1, n + 1):
            node = ops[k % len(ops)](node, leaf(k), transform=ns.translate(0.01 * k, -0.005 * k, 0.0) * ns.rotate(3.0 * k, 0, 2.0 * k))
        return node

    def random_tree(depth):
        if depth == 0 or (depth < 5 and rng.rand() < 0.2):
            return leaf(int(rng.randint(1000)))
        op = [ns.Union, ns.Union, ns.Intersect, ns.Subtract][rng.randint(4)]
        return op(random_tree(depth - 1), random_tree(depth - 1), transform=ns.translate(*(0.05 * rng.randn(3))))

    def nodes(p):
        return 1 + nodes(p.primitive_a) + nodes(p.primitive_b) if hasattr(p, "primitive_a") else 1

    world = ns.World()
    trees = [chain(12, [ns.Union]), chain(14, [ns.Union, ns.Subtract, ns.Union, ns.Intersect, ns.Union])]
    while len(trees) < 5:
        t = random_tree(6)
        if 33 <= nodes(t) <= 70:
            trees.append(t)
    assert nodes(trees[0]) == 25 and max(nodes(t) for t in trees) >= 33
    for k, t in enumerate(trees):
        t.parent = world
        t.transform = ns.translate(1.6 * (k - 2), 0.3 * (k % 2), 0.0) * (t.transform or ns.translate(0, 0, 0))
        t.material = ns.AbsorbingSurface()
    flat = world.flatten()
    sc = world.build_accelerator()
    n = 60000
    o = rng.uniform(-4.5, 4.5, (n, 3)) * [1.0, 0.5, 1.0]
    tgt = np.stack([1.6 * (rng.randint(5, size=n) - 2) + 0.3 * rng.randn(n), 0.3 * rng.randn(n), 0.3 * rng.randn(n)], axis=1)
    d = tgt - o
    d /= np.linalg.norm(d, axis=1)[:, None]
    m = np.where(rng.rand(n) < 0.2, rng.uniform(0.5, 6.0, n), np.inf)
    dev = sc.hit_batch(o, d, m, geometry=True)
    ref = orc.hit_batch(flat, o, d, m, geometry=True, threads=orc.max_threads())
    assert (ref["prim"] >= 0).sum() > n // 4
    assert_hits_equal(dev, ref, geometry=True)
    for index in range(len(trees)):
        counts, t, ex = sc.roots_batch(index, o[:6000], d[:6000], None, max_roots=96)
        rc, rt, rex = orc.roots_batch(flat, index, o[:6000], d[:6000], None, max_roots=96)
        assert eq(counts, rc) and counts.max() >= 2, index
        mask = np.arange(96)[None, :] < counts[:, None]
        assert eq(t[mask], rt[mask]) and eq(ex[mask], rex[mask]), index
    pts = np.concatenate([tgt[:20000] + 0.2 * rng.randn(20000, 3), rng.uniform(-4, 4, (5000, 3))])
    cd, cr = sc.contains_batch(pts), orc.contains_batch(flat, pts)
    assert eq(cd, cr) and cr.sum() > 2000
    # path traced: the same solids as glass, diffuse and emitting bodies inside an emitting shell
    mats = [ns.Lambert(ns.ConstantSF(0.8)), ns.Dielectric(ns.ConstantSF(1.5), ns.ConstantSF(1.0)), ns.UniformVolumeEmitter(ns.ConstantSF(1.0), 0.5),
            ns.Lambert(ns.ConstantSF(0.5)), ns.Dielectric(ns.ConstantSF(1.3), ns.ConstantSF(1.0), transmission_only=True)]
    for k, prim in enumerate(list(world._primitives)):
        prim.material = mats[k % len(mats)]
    ns.Box(P(-6, -6, -6), P(6, 6, 6), world, material=ns.UniformSurfaceEmitter(ns.ConstantSF(1.0), 1.0))
    pipe = ns.SpectralRadiancePipeline2D()
    cam = ns.PinholeCamera((80, 48), fov=70, parent=world, pipelines=[pipe], frame_sampler=ns.RectFrameSampler2D(), transform=ns.translate(0.1, 0.2, -4.5))
    cam.pixel_samples, cam.spectral_bins, cam.spectral_rays, cam.quiet = 3, 3, 1, True
    cam.ray_extinction_prob, cam.ray_extinction_min_depth, cam.ray_max_depth = 0.05, 2, 40
    cam.render_engine = ns.HipEngine(rng="philox", seed=4)
    cam.observe()
    keep = []
    desc = cam.render_desc(world, None, cam._slice_spectrum()[0], cam.render_engine, keep, rect=(0, 0, 80, 48))
    om, ov, rays = orc.render_pinhole(world.flatten(), desc, threads=orc.max_threads())
    assert eq(np.array(pipe.frame.mean), om.reshape(48, 80, 3).transpose(1, 0, 2)) and eq(np.array(pipe.frame.variance), ov.reshape(48, 80, 3).transpose(1, 0, 2))
    assert cam.stats["rays"] == rays


def test_arena_csg_scene_passes_do_not_share_node_states(orc, ns):
    """A CSG operand tree of more than 16 nodes keeps its stream-merge node states in ONE arena per scene, indexed by the launch's own
    blockIdx (dev_csg.hpp: csg_arena_slots). Passes of such a scene must not overlap on the private lanes — the slices of one
    observe() used to, and wrote each other's states in the middle of a merge: every one-bin slice of a 12-slice observe() against
    the oracle, bit for bit (csg.pyx:132-234), and three more renders of the same frame."""
    rng = np.random.RandomState(5)
    P = ns.Point3D

    def leaf(k):
        t = ns.translate(0.22 * np.cos(0.9 * k), 0.22 * np.sin(1.3 * k), 0.08 * ((k % 5) - 2)) * ns.rotate(17.0 * k, 11.0 * k, 5.0 * k)
        kind = k % 3
        if kind == 0:
            return ns.Sphere(0.25 + 0.02 * (k % 4), transform=t)
        if kind == 1:
            return ns.Box(P(-0.2, -0.25, -0.15), P(0.25, 0.2, 0.3), transform=t)
        return ns.Cylinder(0.2, 0.5, transform=t)

    def tree(depth):
        if depth == 0:
            return leaf(int(rng.randint(1000)))
        op = [ns.Union, ns.Union, ns.Subtract, ns.Intersect][rng.randint(4)] if depth < 4 else ns.Union
        return op(tree(depth - 1), tree(depth - 1), transform=ns.translate(*(0.08 * rng.randn(3))))

    world = ns.World()
    sf = ns.InterpolatedSF([300, 500, 800], np.array([0.2, 1.0, 0.5]))
    for k in range(2):
        t = tree(4)                                                              # 31 nodes, 16 leaves: the arena, no state-free form
        t.parent, t.transform, t.material = world, ns.translate(1.3 * (k - 0.5), 0.0, 0.0), ns.UniformSurfaceEmitter(sf, 1.0 + k)
    ns.Box(P(-4, -4, -4), P(4, 4, 4), world, material=ns.UniformSurfaceEmitter(ns.ConstantSF(0.1), 1.0))
    pipe = ns.SpectralRadiancePipeline2D()
    cam = ns.PinholeCamera((160, 96), fov=60, parent=world, pipelines=[pipe], frame_sampler=ns.RectFrameSampler2D(), transform=ns.translate(0.0, 0.1, -3.2))
    cam.pixel_samples, cam.spectral_bins, cam.spectral_rays, cam.quiet = 1, 12, 12, True
    cam.render_engine = ns.HipEngine(rng="philox", seed=12)
    cam.observe()
    first = np.array(pipe.frame.mean)
    flat = world.flatten()
    slices = cam._slice_spectrum()
    assert len(slices) == 12
    for k, sl in enumerate(slices):
        keep = []
        desc = cam.render_desc(world, None, sl, cam.render_engine, keep, rect=(0, 0, 160, 96))
        m, v, rays = orc.render_pinhole(flat, desc, threads=orc.max_threads())
        assert eq(first[:, :, k], m.reshape(96, 160).T), k
    assert (first[:, :, 5] > 0.11).sum() > 2000                                  # the solids are in view
    # the same again on fresh pipelines: passes in flight next to each other may not disturb one another from call to call
    for _ in range(3):
        pipe2 = ns.SpectralRadiancePipeline2D()
        cam.pipelines = [pipe2]
        cam.render_engine = ns.HipEngine(rng="philox", seed=12)
        cam.observe()
        assert eq(np.array(pipe2.frame.mean), first)




@pytest.mark.parametrize("world_size", [2, 3, 8, -3])
def test_multi_rank_exchange_through_transport_stub(ns, tmp_path, world_size):
    """librsx's multi-rank framebuffer exchange (csrc/rsx_comm.hpp; the reference's counterpart: the result queue of
    workflow.py:201-251 folded by power.pyx:424-437) EXECUTED with W = 2, 3 and 8 ranks on the one GPU of the box: W processes share
    device 0 and librsx talks to tests/stub_rccl/librccl_stub.so (RSX_RCCL_LIB) — the eleven nccl* symbols over files, group
    semantics kept — so gather_runs' (me +- k) % W schedule, rsx_frame_segment's offsets on the device, the [W][mine] fold of
    rsx_allreduce_frame, k_pack_bins and the chunked workspace of rsx_allgather_bins all run. Tile and slice sharding: every rank's
    exchanged frame equals the one-process render bit for bit; sample sharding: every rank holds the same frame (bit for bit) and it
    equals the one-process render of the same W x passes x spp samples within the merge's tolerance (the fold associates
    differently from a sequential accumulation — 1e-12 relative on the mean, 16 eps (mean^2 + var) per merged pass on the variance),
    sample counts exact.
    world_size -3 = three ranks with the stub's ASYNCHRONOUS completion (RSX_STUB_ASYNC=1): ncclGroupEnd returns at once and the
    bytes land 20 ms after the stream's earlier work has finished, moved by another thread over another stream while the caller's
    stream is parked — what a consumer on a different stream, a host read without a synchronisation or a send buffer recycled too
    early would get wrong."""
    import shutil
    import subprocess
    import sys
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    stub = str(tmp_path / "librccl_stub.so")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-O2", os.path.join(ROOT, "tests", "stub_rccl", "rccl_stub.cpp"), "-o", stub, "-lpthread"])
    modes = "tile,tile_balanced,sample,slice,slice_chunked"
    env = dict(os.environ, RSX_RCCL_LIB=stub, RSX_STUB_DIR=str(tmp_path), RSX_DEVICE="0", RSX_STUB_TIMEOUT_S="240")
    if world_size < 0:
        world_size = -world_size
        env.update(RSX_STUB_ASYNC="1", RSX_STUB_DELAY_MS="20")
    procs = []
    for r in range(world_size):
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "stub_rccl", "worker.py"), str(r), str(world_size), str(tmp_path), modes],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=900)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-1500:] for o in outs)
    # the one-process renders
    NX, NY, SPP, BINS, PASSES = 72, 40, 3, 6, 2

    def single(slices, passes, spp_offsets):
        world, mesh, box = scenes.build_c2(ns, n=24)
        cam, pipe = scenes.c2_camera(ns, world, (NX, NY), spp=SPP, bins=BINS)
        cam.spectral_rays = slices
        cam.render_engine = ns.HipEngine(rng="philox", seed=77)
        cam.frame_sampler = ns.RectFrameSampler2D()
        for off in spp_offsets:
            cam.render_engine.sample_offset = off
            cam.observe()
        f = pipe.frame
        return np.array(f.mean), np.array(f.variance), np.array(f.samples)

    ref_tile = single(1, PASSES, [p * SPP for p in range(PASSES)])
    ref_slice = single(BINS, PASSES, [p * SPP for p in range(PASSES)])
    ref_sample = single(1, PASSES * world_size, [k * SPP for k in range(PASSES * world_size)])       # the same counters, all on one rank
    load = lambda mode, r: np.load(str(tmp_path / ("%s_rank%d.npz" % (mode, r))))
    for r in range(world_size):
        for mode, ref in (("tile", ref_tile), ("tile_balanced", ref_tile), ("slice", ref_slice), ("slice_chunked", ref_slice)):
            got = load(mode, r)
            assert eq(got["mean"], ref[0]) and eq(got["variance"], ref[1]) and eq(got["samples"], ref[2]), (mode, r)
        got = load("sample", r)
        first = load("sample", 0)
        assert eq(got["mean"], first["mean"]) and eq(got["variance"], first["variance"]) and eq(got["samples"], first["samples"]), r
    got = load("sample", 0)
    assert eq(got["samples"], ref_sample[2]) and (got["samples"] == PASSES * world_size * SPP).all()
    assert np.allclose(got["mean"], ref_sample[0], rtol=1e-12, atol=0.0)
    eps = np.finfo(np.float64).eps
    assert (np.abs(got["variance"] - ref_sample[1]) <= 16 * eps * (ref_sample[0] ** 2 + ref_sample[1]) * (PASSES * world_size)).all()
    assert ref_tile[0].max() > 0 and ref_slice[0].max() > 0


def test_auto_batched_passes_equal_separate_passes(orc, ns):
    """HipEngine.auto_batch (on by default): consecutive small observe() calls are held back and submitted as one library call. The
    frames must be those of the separate passes bit for bit — whatever ends a batch: a full unit (64 / spp passes), a read of the
    frame, ctx.synchronize(), a scenegraph change, a changed camera or engine setting — and one batch is also checked directly against
    the oracle (observer.pyx:265-309 called K times, power.pyx:424-437 merging each pass)."""
    from source_amd.device import get_context

    def run(auto, spp, script):
        world, mesh, box = scenes.build_c2(ns, n=24)
        cam, pipe = scenes.c2_camera(ns, world, (96, 64), spp=spp, bins=5)
        cam.frame_sampler = ns.RectFrameSampler2D()
        cam.render_engine = ns.HipEngine(rng="philox", seed=3, auto_batch=auto)
        seen = []
        for step in script:
            if step == "observe":
                cam.observe()
                assert cam.stats["rays"] == 96 * 64 * spp
            elif step == "read":
                seen.append(np.array(pipe.frame.mean).sum())
            elif step == "sync":
                get_context().synchronize()
            elif step == "move":
                mesh.transform = ns.translate(0.01, 0.0, 0.0) * (mesh.transform or ns.translate(0, 0, 0))
            elif step == "camera":
                cam.fov = 40
            elif step == "offset":
                cam.render_engine.sample_offset = 1000
            elif step == "shuffled":
                cam.frame_sampler = ns.FullFrameSampler2D()
            elif step == "brighter":                       # a material parameter changed in place: no scenegraph notification
                box.material.scale = box.material.scale * 1.5
        f = pipe.frame
        return np.array(f.mean), np.array(f.variance), np.array(f.samples), seen, world, cam, pipe

    scripts = {
        1: ["observe"] * 70 + ["read"] + ["observe"] * 3 + ["sync"] + ["observe"] * 5 + ["move"] + ["observe"] * 4 + ["camera"] + ["observe"] * 3
           + ["offset"] + ["observe"] * 6 + ["shuffled"] + ["observe"] * 5 + ["brighter"] + ["observe"] * 3,
        2: ["observe"] * 33 + ["move"] + ["observe"] * 2 + ["read"] + ["observe"] * 31,
        16: ["observe"] * 9,
    }
    for spp, script in scripts.items():
        a = run(True, spp, script)
        b = run(False, spp, script)
        assert eq(a[0], b[0]) and eq(a[1], b[1]) and eq(a[2], b[2]) and a[3] == b[3], spp
        assert (a[2] == spp * script.count("observe")).all()
    # one batch against the oracle: 4 passes of 2 spp = the pixel's samples 0 .. 7, merged pass by pass
    m, v, n, _, world, cam, pipe = run(True, 2, ["observe"] * 4)
    flat = world.flatten()
    from source_amd import distributed as D
    om = ov = on = None
    for p in range(4):
        keep = []
        desc = cam.render_desc(world, None, cam._slice_spectrum()[0], cam.render_engine, keep, rect=(0, 0, 96, 64), sample_offset=2 * p)
        pm, pv, rays = orc.render_pinhole(flat, desc, threads=orc.max_threads())
        pm, pv = pm.reshape(64, 96, 5).transpose(1, 0, 2), pv.reshape(64, 96, 5).transpose(1, 0, 2)
        if om is None:
            om, ov, on = pm, pv, np.full(pm.shape, 2, dtype=np.int32)
        else:
            om, ov, on = D.combine_arrays(om, ov, on, pm, np.maximum(pv, 0.0), np.full(pm.shape, 2, dtype=np.int32))
    assert eq(n, on) and eq(m, om) and eq(v, ov)              # (combine_arrays is the reference's merge law element by element: the same bits)


def test_several_path_passes_per_call_equal_separate_passes(orc, ns):
    """HipEngine(passes_per_call=K) on path-traced scenes: the path kernel traces the K passes' paths in one launch (Philox counters of K
    consecutive passes), k_accumulate replays every pixel's term lists pass by pass — per pass the recurrence from its first sample and
    the frame merge. The frame must be that of K separate observe() calls bit for bit: Cornell box (importance sampling, glass; 4 spp per
    pass: the chunked replay; 1 spp: the plain one), the diffuse room with its CSG solid, volume emitter and null shell (two-pass CSG
    path kernels, volume terms), several spectral slices (deferred end-of-pass checks) — and one call is checked against the oracle
    merged pass by pass with combine_samples (power.pyx:424-437)."""
    from source_amd import distributed as D
    cases = ((lambda: scenes.build_cornell(ns), lambda w, spp: scenes.cornell_camera(ns, w, (56, 40), spp, 5), 4, 3, 1),
             (lambda: scenes.build_cornell(ns), lambda w, spp: scenes.cornell_camera(ns, w, (56, 40), spp, 5), 1, 5, 1),
             (lambda: scenes.build_lambert(ns), lambda w, spp: scenes.lambert_camera(ns, w, (40, 32), spp=spp, bins=4, extinction=(0.2, 2, 9)), 2, 4, 1),
             (lambda: scenes.build_cornell(ns), lambda w, spp: scenes.cornell_camera(ns, w, (40, 32), spp, 6), 4, 2, 3))
    for build, camera, spp, K, slices in cases:
        frames = []
        for per_call in (K, 1):
            world, prims = build()
            cam, pipe = camera(world, spp)
            cam.spectral_rays = slices
            cam.frame_sampler = ns.RectFrameSampler2D()
            cam.render_engine = ns.HipEngine(rng="philox", seed=23, passes_per_call=per_call, auto_batch=False)
            rays = 0
            for _ in range(2 * K // per_call):                                   # two calls of K passes / 2 K separate passes
                cam.observe()
                rays += cam.stats["rays"]
            f = pipe.frame
            frames.append((np.array(f.mean), np.array(f.variance), np.array(f.samples), rays, world, cam))
        a, b = frames
        assert eq(a[0], b[0]) and eq(a[1], b[1]) and eq(a[2], b[2]) and a[3] == b[3], (spp, K, slices)
        assert (a[2] == 2 * K * spp).all() and (a[0] > 0).mean() > 0.2
    # the first case's first call against the oracle: K = 3 passes of 4 spp, merged pass by pass
    build, camera, spp, K, _ = cases[0]
    world, prims = build()
    cam, pipe = camera(world, spp)
    cam.frame_sampler = ns.RectFrameSampler2D()
    cam.render_engine = ns.HipEngine(rng="philox", seed=23, passes_per_call=K, auto_batch=False)
    cam.observe()
    nx, ny = cam.pixels
    flat = world.flatten()
    om = ov = on = None
    for p in range(K):
        keep = []
        desc = cam.render_desc(world, None, cam._slice_spectrum()[0], cam.render_engine, keep, rect=(0, 0, nx, ny), sample_offset=spp * p)
        pm, pv, _ = orc.render_pinhole(flat, desc, threads=orc.max_threads())
        pm, pv = pm.reshape(ny, nx, 5).transpose(1, 0, 2), pv.reshape(ny, nx, 5).transpose(1, 0, 2)
        if om is None:
            om, ov, on = pm, pv, np.full(pm.shape, spp, dtype=np.int32)
        else:
            om, ov, on = D.combine_arrays(om, ov, on, pm, np.maximum(pv, 0.0), np.full(pm.shape, spp, dtype=np.int32))
    f = pipe.frame
    assert eq(np.array(f.samples), on) and eq(np.array(f.mean), om) and eq(np.array(f.variance), ov)


def test_toolchain_divergent_loop_exit_workaround(tmp_path):
    """hipcc 7.2 miscompiles per-lane loops that the lanes of a wave leave at different turns through a `return` / `continue` in the
    middle of the body (tests/toolchain/divergent_loop_exit.hip holds both forms of one loop and says so in its output). librsx writes
    such loops — CSG contains(), the stream merge, the volume enumeration of the path kernel — with ONE exit test; this test builds
    the reproducer with the box's own hipcc and fails if that form stops being right, i.e. if the next toolchain moves the fault."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "divergent_loop_exit")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-unused-value", "-Wno-unused-result",
                           os.path.join(ROOT, "tests", "toolchain", "divergent_loop_exit.hip"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(r.stdout.strip())
    assert r.returncode == 0 and "single_exit wrong: 0 of" in r.stdout, r.stdout + r.stderr


def test_packet_walk_equals_per_lane_walk_at_any_sample_count():
    """The packet walk (dev_packet.hpp: the 64 rays of a unit traverse the trees together) serves passes of 16 samples per pixel
    and more by default; forced on for passes of 2, 3, 5 and 12 samples per pixel — units of up to 32 different pixels, odd splits of
    pixels over units, frames that are not a multiple of the 8 x 8 unit tiles — it must give the frames of the per-lane walk bit for
    bit (which other tests tie to the oracle), fused Welford or not."""
    import subprocess
    import sys
    code = """
import hashlib, sys
import numpy as np
sys.path.insert(0, %r)
from source_amd import api as ns, scenes
world = scenes.build_c3(ns, n=24)[0]
solids = scenes.build_csg_demo(ns)[0]                    # (CSG scenes: the packet kernel is the fast pass, the redo pass walks per lane)
out = []
for (nx, ny, spp) in ((72, 44, 2), (50, 61, 3), (64, 40, 5), (33, 47, 12), (40, 24, 16), (24, 24, 64), (-56, 40, 4), (-40, 33, 16), (-24, 24, 64)):
    pipe = ns.SpectralRadiancePipeline2D()
    if nx < 0:
        cam, _ = scenes.csg_camera(ns, solids, (-nx, ny), spp=spp, bins=5)
    else:
        cam, _ = scenes.c3_camera(ns, world, (nx, ny), spp=spp, bins=5)
    cam.pipelines = [pipe]
    cam.frame_sampler = ns.RectFrameSampler2D()
    cam.render_engine = ns.HipEngine(rng="philox", seed=3)
    cam.observe(); cam.observe()
    h = hashlib.sha256()
    for a in (pipe.frame.mean, pipe.frame.variance, pipe.frame.samples):
        h.update(np.ascontiguousarray(a).tobytes())
    assert (pipe.frame.samples == 2 * spp).all() and pipe.frame.mean.max() > 0
    out.append(h.hexdigest())
print(" ".join(out))
""" % ROOT
    digests = {}
    for name, env in (("per-lane", dict(RSX_PACKET_MIN_SPP="0")), ("packet", dict(RSX_PACKET_MIN_SPP="2", RSX_FUSE="0")),
                      ("packet fused", dict(RSX_PACKET_MIN_SPP="2", RSX_FUSE="1", RSX_PIPELINE="1")),
                      ("packet on vertex records", dict(RSX_PACKET_MIN_SPP="2", RSX_CAMERA_RELATIVE="0")), ("default", {})):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, name + ": " + r.stderr[-2000:]
        digests[name] = r.stdout.strip().splitlines()[-1]
    assert len(set(digests.values())) == 1, digests


def test_passes_per_call_equals_separate_passes(ns):
    """rsx_render_desc.passes / HipEngine(passes_per_call=K): K passes submitted as one library call leave the frames that K observe()
    calls leave (observer.pyx:265-309 called K times: K Welford chains per pixel and bin, K merges) — mean, variance and sample counts
    bit for bit: mesh scenes, a CSG scene, a task-list pass with a mask, several spectral slices, power and radiance pipelines."""
    def frames(make, K, calls, together, uneven=False):
        world, cam, pipes = make()
        if uneven:                                          # pixels with different histories in one wave: the merge's per-lane form
            sampler = cam.frame_sampler
            mask = np.zeros(cam.pixels, dtype=bool)
            mask[5:-3:2, 1:-2] = True
            cam.frame_sampler = ns.FullFrameSampler2D(mask)
            cam.render_engine = ns.HipEngine(rng="philox", seed=5)
            cam.observe()
            cam.frame_sampler = sampler
        cam.render_engine = ns.HipEngine(rng="philox", seed=11, passes_per_call=K if together else 1)
        for _ in range(calls * (1 if together else K)):
            cam.observe()
        return [np.concatenate([p.frame.mean.ravel(), p.frame.variance.ravel(), p.frame.samples.ravel().astype(np.float64)]) for p in pipes]

    def c2(spp, pixels=(72, 50), slices=1):
        def make():
            world = scenes.build_c2(ns, n=24)[0]
            pipes = [ns.SpectralRadiancePipeline2D(), ns.SpectralPowerPipeline2D()]
            cam, _ = scenes.c2_camera(ns, world, pixels, spp=spp, bins=6)
            cam.spectral_rays = slices
            cam.pipelines = pipes
            cam.frame_sampler = ns.RectFrameSampler2D()
            return world, cam, pipes
        return make

    def c3_masked():
        world = scenes.build_c3(ns, n=24)[0]
        pipes = [ns.SpectralRadiancePipeline2D()]
        cam, _ = scenes.c3_camera(ns, world, (40, 36), spp=3, bins=5)
        cam.pipelines = pipes
        mask = np.zeros((40, 36), dtype=bool)
        mask[3:31, 5:29] = True
        mask[10, 10] = False
        cam.frame_sampler = ns.FullFrameSampler2D(mask)
        return world, cam, pipes

    def csg():
        world = scenes.build_csg_demo(ns)[0]
        pipes = [ns.SpectralRadiancePipeline2D()]
        cam, _ = scenes.csg_camera(ns, world, (48, 40), spp=2, bins=4)
        cam.pipelines = pipes
        cam.frame_sampler = ns.RectFrameSampler2D()
        return world, cam, pipes

    for name, make, K, calls in (("c2 1 spp x 16", c2(1), 16, 2), ("c2 1 spp x 3", c2(1), 3, 1), ("c2 4 spp x 5, 3 slices", c2(4, slices=3), 5, 2),
                                 ("c2 20 spp x 4", c2(20, (33, 21)), 4, 1), ("c2 1 spp x 100 (served as 64 + 32 + 4)", c2(1, (40, 24)), 100, 1), ("c3 masked 3 spp x 7", c3_masked, 7, 2), ("csg 2 spp x 8", csg, 8, 1)):
        uneven = name == "c2 1 spp x 3"
        one, many = frames(make, K, calls, True, uneven), frames(make, K, calls, False, uneven)
        for a, b in zip(one, many):
            assert a.max() > 0 and np.array_equal(a.view(np.uint64), b.view(np.uint64)), name

    # what the option does not cover fails loudly
    world, cam, pipes = c2(1)()
    cam.pipelines = [ns.SpectralRadiancePipeline2D(accumulate=False)]
    cam.render_engine = ns.HipEngine(passes_per_call=4)
    with pytest.raises(ValueError):
        cam.observe()
    world = scenes.build_lambert(ns)[0]                      # (path-traced scenes are covered: test_several_path_passes_per_call_equal_separate_passes)
    cam, _ = scenes.lambert_camera(ns, world)
    cam.pipelines = [ns.RGBPipeline2D()]
    cam.render_engine = ns.HipEngine(passes_per_call=4)
    with pytest.raises(Exception, match="passes"):
        cam.observe()


def test_handed_on_paths_give_the_same_frames():
    """Path passes: waves that run out of new rays hand their last paths to a second, small launch (k_render_trace_path, PathState) —
    by default only in passes that overlap others (the spectral slices of one observe()). Forced on for every path pass
    (RSX_PATH_DONATE=2), forced off (0) and by default the frames and ray counts must be the same bit for bit: glass with
    trapped paths over three slices, a diffuse room with CSG and volumes (the two-pass CSG form), the Cornell box."""
    import subprocess
    import sys
    code = """
import hashlib, sys
import numpy as np
sys.path.insert(0, %r)
from source_amd import api as ns, scenes
out = []
for build, camera, kw in ((scenes.build_glass, scenes.glass_camera, dict(pixels=(96, 80), spp=4)),
                          (scenes.build_lambert, scenes.lambert_camera, dict(pixels=(64, 48), spp=4)),
                          (scenes.build_cornell, scenes.cornell_camera, dict(pixels=(96, 96), spp=4)),
                          (scenes.build_prism, scenes.prism_camera, dict(pixels=(96, 64), spp=2, bins=8, spectral_rays=8))):
    world = build(ns)[0]
    cam, pipe = camera(ns, world, **kw)
    cam.frame_sampler = ns.RectFrameSampler2D()
    cam.render_engine = ns.HipEngine(rng="philox", seed=5)
    cam.observe(); cam.observe()
    h = hashlib.sha256()
    for a in (pipe.frame.mean, pipe.frame.variance, pipe.frame.samples):
        h.update(np.ascontiguousarray(a).tobytes())
    assert pipe.frame.mean.max() > 0
    out.append(h.hexdigest() + ":%%d" %% cam.stats["rays"])
print(" ".join(out))
""" % ROOT
    digests = {}
    for mode in ("0", "1", "2"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, RSX_PATH_DONATE=mode), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, mode + ": " + r.stderr[-2000:]
        digests[mode] = r.stdout.strip().splitlines()[-1]
    # the same four scenes with the >= 2-leaf rule for the wide slots (RSX_NO_WIDE_ALL=1: single-leaf primitives are met in their leaf
    # instead of answered before the walk — rsx_scene_create): when a primitive is asked never shows in a frame
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, RSX_NO_WIDE_ALL="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, "RSX_NO_WIDE_ALL: " + r.stderr[-2000:]
    digests["no_wide_all"] = r.stdout.strip().splitlines()[-1]
    assert len(set(digests.values())) == 1, digests


def test_task_lists_render_in_coherent_order(ns):
    """Philox frames do not depend on the order of a task list, so observe() renders a shuffled full-frame list (FullFrameSampler2D, as
    sampler2d.pyx:42-102 builds it) as a rectangle and any other list in 8 x 8 tile order (`_coherent_tasks`): the frames must equal the
    rectangle sampler's bit for bit — on the pixels a mask selects, and nowhere else."""
    world = scenes.build_c3(ns, n=24)[0]

    def frame(sampler, spp=5):
        pipe = ns.SpectralRadiancePipeline2D()
        cam, _ = scenes.c3_camera(ns, world, (56, 44), spp=spp, bins=4)
        cam.pipelines = [pipe]
        cam.frame_sampler = sampler
        cam.render_engine = ns.HipEngine(rng="philox", seed=9)
        cam.observe()
        cam.observe()
        return pipe.frame.mean.copy(), pipe.frame.variance.copy(), pipe.frame.samples.copy(), cam

    rect = frame(ns.RectFrameSampler2D())
    full = frame(ns.FullFrameSampler2D())
    assert isinstance(full[3]._coherent_cache[1], RectTasks)
    for a, b in zip(rect[:3], full[:3]):
        assert np.array_equal(a, b) and a.max() > 0
    mask = np.zeros((56, 44), dtype=bool)
    mask[3:50:2, 1:40] = True
    mask[7, 7] = True
    part = frame(ns.FullFrameSampler2D(mask))
    assert not isinstance(part[3]._coherent_cache[1], RectTasks)
    for a, b in zip(rect[:3], part[:3]):
        assert np.array_equal(a[mask], b[mask]) and not b[~mask].any()


def test_staged_path_passes_equal_the_one_kernel_form(orc, ns):
    """Path-traced passes level by level (dev_wavefront.hpp: one launch per path segment over lists of live paths filed by material arm,
    the last paths handed to the drain launch or walked level by level) against the one-kernel form on the same calls: frames and Ray.ray_count
    identical — Lambert with a volume emitter and a CSG solid (two-pass), clear and tinted glass (attenuation terms), the Cornell box
    with importance sampling, the prism scene (state-free CSG evaluator + redo pass) — and two of them against the oracle directly.
    One spectral slice per observe(): several slices overlap on private lanes and keep the one-kernel form."""
    from source_amd.device import get_context
    ctx = get_context()

    def cases():
        world, _ = scenes.build_lambert(ns)
        yield "lambert", world, scenes.lambert_camera(ns, world, (96, 80), 8, 5, (0.01, 3, 500)), True
        world, _ = scenes.build_lambert(ns, with_volume=False, csg=False)
        yield "lambert_plain", world, scenes.lambert_camera(ns, world, (64, 48), 6, 6, (0.3, 1, 4)), False
        for clear in (True, False):
            world, _ = scenes.build_glass(ns, unit_transmission=clear)
            yield "glass_%s" % clear, world, scenes.glass_camera(ns, world, (96, 72), 6, 6, 1, (0.01, 3, 500) if clear else (0.1, 2, 20)), not clear
        world, _ = scenes.build_cornell(ns)
        yield "cornell", world, scenes.cornell_camera(ns, world, (128, 96), 4, 6), False
        world, _ = scenes.build_prism(ns)
        yield "prism", world, scenes.prism_camera(ns, world, (96, 64), 4, 4, 1), False

    try:
        for name, world, (cam, pipe), against_oracle in cases():
            cam.frame_sampler = ns.RectFrameSampler2D()
            pipe.accumulate = False                                # (every observe() starts a fresh frame)
            frames = []
            for mode in (1, 0, 3, 2):                              # (2 / 3: the forms with the mesh walk, which none of these scenes needs)
                ctx.set_path_stages(mode, 0 if mode & 1 else -1)
                cam.render_engine = ns.HipEngine(rng="philox", seed=41)
                cam.observe()
                frames.append((pipe.frame.mean.copy(), pipe.frame.variance.copy(), pipe.frame.samples.copy(), cam.stats["rays"]))
                cam.render_engine.sample_offset = 0
            for other in frames[1:]:
                assert eq(frames[0][0], other[0]) and eq(frames[0][1], other[1]) and eq(frames[0][2], other[2]), name
                assert frames[0][3] == other[3], (name, frames[0][3], other[3])
            assert (frames[0][0] > 0).mean() > 0.05, name
            if against_oracle:
                w, h = cam.pixels
                sl = cam._slice_spectrum()[0]
                keep = []
                desc = cam.render_desc(world, None, sl, cam.render_engine, keep, rect=(0, 0, w, h))
                om, ov, n_rays = orc.render_pinhole(world.flatten(), desc, threads=orc.max_threads())
                assert eq(frames[0][0], om.reshape(h, w, sl.bins).transpose(1, 0, 2)) and eq(frames[0][1], ov.reshape(h, w, sl.bins).transpose(1, 0, 2)), name
                assert frames[0][3] == n_rays, name
            cam.parent = None
    finally:
        ctx.set_path_stages(-1, -1)


@pytest.mark.gpu
def test_three_worlds_in_one_process_render_without_stalls(ns):
    """Round 5's open fault: after a second world was built in a process, two or three calls of the next few hundred milliseconds took 60 - 90 ms
    (2 ms otherwise) — the OpenMP team of the host KD build, 256 spinning threads on a 16-core cgroup quota, got the whole process throttled
    (tools/r6_world_stalls.py; csrc/rsx_host.cpp: host_team_size). Three worlds, sixty synchronised calls each: after a world's first call
    no call may take more than 25 ms (the stalls were >= 45 ms; a steady call takes about 2)."""
    import time
    from source_amd.device import get_context
    worst = []
    for w in range(3):
        world = scenes.build_cornell(ns)[0]
        cam, pipe = scenes.cornell_camera(ns, world, (256, 256), spp=4, bins=15)
        cam.frame_sampler = ns.RectFrameSampler2D()
        cam.render_engine = ns.HipEngine(rng="philox", seed=5, auto_batch=False)
        world.build_accelerator()
        times = []
        for k in range(60):
            t0 = time.perf_counter()
            cam.observe()
            get_context().synchronize()
            times.append((time.perf_counter() - t0) * 1e3)
        worst.append(max(times[1:]))
    assert max(worst) < 25.0, worst


@pytest.mark.gpu
def test_packet_walk_takes_the_compiled_transitions_where_the_quotient_needs_its_range_test(orc, ns):
    """The hand-written descent (dev_packet.hpp: packet_descend) serves ray spaces in which the hoisted-reciprocal quotient needs no
    per-node range test (PacketSpace::fast == 7); any other space takes the compiled form of the same transitions. Such a space, made
    on purpose: a camera whose origin has an x coordinate of exactly 0 in a world whose tree holds splits closer to zero than 2^-240
    (a sphere of radius 1e-80 at the origin: numerators `split - origin` then leave the shortcut's operand range). A 64-spp packet pass
    over an instanced mesh world must equal the oracle bit for bit on that path too."""
    world = scenes.build_c3(ns, n=24)[0]
    ns.Sphere(1e-80, world, material=ns.AbsorbingSurface())
    pipe = ns.SpectralRadiancePipeline2D()
    cam = ns.PinholeCamera((40, 24), fov=60, parent=world, pipelines=[pipe], frame_sampler=ns.RectFrameSampler2D(),
                           transform=ns.translate(0, 0.35, -0.9))
    cam.pixel_samples, cam.spectral_bins, cam.spectral_rays, cam.quiet = 64, 5, 1, True
    cam.render_engine = ns.HipEngine(rng="philox", seed=23)
    cam.observe()
    mean, var = pipe.frame.mean.copy(), pipe.frame.variance.copy()
    assert (pipe.frame.samples == 64).all() and mean.max() > 0
    keep = []
    desc = cam.render_desc(world, None, cam._slice_spectrum()[0], cam.render_engine, keep, rect=(0, 0, 40, 24))
    m, v, rays = orc.render_pinhole(world.flatten(), desc, threads=orc.max_threads())
    assert rays == 40 * 24 * 64
    assert eq(mean, m.reshape(24, 40, 5).transpose(1, 0, 2)) and eq(var, v.reshape(24, 40, 5).transpose(1, 0, 2))
