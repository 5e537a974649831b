"""Host-side product logic (no GPU): math vs the reference's matrices, the C-ABI surface, MT stream, scenegraph behaviour."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from source_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def eq(a, b):
    return np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


def test_library_exports_every_declared_symbol():
    """Every function include/rsx.h declares must be exported by the built librsx.so (and bound in _lib.SYMBOLS)."""
    header = open(os.path.join(os.path.dirname(_lib._HERE), "include", "rsx.h")).read()
    declared = set(re.findall(r"\b(rsx_[a-z0-9_]+)\s*\(", header))
    bound = {name for name, _, _ in _lib.SYMBOLS}
    assert declared == bound, declared ^ bound
    handle = C.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(handle, name), name
    assert b"gfx950" in _lib.lib().rsx_version()


def test_struct_layouts_match_header():
    assert C.sizeof(_lib.KDNode) == 16
    assert C.sizeof(_lib.Primitive) == 24 + 8 * (6 + 16 + 16 + 6)
    assert C.sizeof(_lib.MT) == 313 * 8
    assert C.sizeof(_lib.Material) == 8 + 8 + 24


def test_no_device_means_loud_failure():
    """On a box without a GPU rsx_init must fail with an error, never fall back."""
    import subprocess, sys
    code = ("import ctypes as C; from source_amd import _lib; h=C.c_void_p(); rc=_lib.lib().rsx_init(0, C.byref(h)); "
            "print(rc, _lib.lib().rsx_last_error())")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=os.path.dirname(_lib._HERE),
                         env={**os.environ, "HIP_VISIBLE_DEVICES": "-1", "ROCR_VISIBLE_DEVICES": "-1"})
    assert out.returncode == 0, out.stderr
    assert out.stdout.split()[0] in ("-2", "-3"), out.stdout


def test_affine_math_matches_reference(ns, golden):
    g = golden("f00_math")
    p = g["params"]
    for k in range(len(p)):
        m = ns.translate(*p[k, 0:3]) * ns.rotate(*p[k, 3:6])
        assert eq(np.array(m.m).reshape(4, 4), g["tr"][k])
        assert eq(np.array(m.inverse().m).reshape(4, 4), g["inv"][k])
        c = m * ns.rotate_x(p[k, 6] * 30) * ns.translate(*p[k, 6:9]) * ns.rotate_z(p[k, 7] * 50) * ns.rotate_y(p[k, 8] * 70)
        assert eq(np.array(c.m).reshape(4, 4), g["chain"][k])
        assert eq(np.array(c.inverse().m).reshape(4, 4), g["chain_inv"][k])
        assert eq(np.array(ns.rotate_vector(p[k, 3], ns.Vector3D(*p[k, 0:3])).m).reshape(4, 4), g["rvec"][k])
        q = ns.Point3D(*p[k, 6:9]).transform(c.inverse())
        v = ns.Vector3D(*p[k, 0:3]).transform(c)
        vn = ns.Vector3D(*p[k, 0:3]).normalise()
        assert eq([q.x, q.y, q.z], g["pts"][k])
        assert eq([v.x, v.y, v.z, vn.x, vn.y, vn.z], g["vecs"][k])


def test_mt_stream_product(golden):
    from source_amd.core import random as rsrandom
    g = golden("f01_mt")
    for s, ref in zip(g["seeds"], g["uniforms"]):
        rsrandom.seed(int(s))
        assert eq(rsrandom.uniform_block(600), ref[:600])
        assert eq([rsrandom.uniform() for _ in range(400)], ref[600:])


def test_scenegraph_bookkeeping(ns):
    world = ns.World()
    s = ns.Sphere(1.0, world, ns.translate(1, 2, 3))
    assert world.primitives == [s] and world._rebuild_accelerator
    node = ns.Node(parent=world, transform=ns.translate(0, 0, 1))
    b = ns.Box(parent=node)
    assert world.primitives == [s, b]
    assert b.to_root().m[11] == 1.0 and b.to_local().m[11] == -1.0
    world._rebuild_accelerator = False
    node.transform = ns.translate(0, 0, 2)                    # GEOMETRY change propagates to the world (world.pyx:220-238)
    assert world._rebuild_accelerator and b.to_root().m[11] == 2.0
    b.parent = None
    assert world.primitives == [s]
    with pytest.raises(ValueError):
        node.parent = node
    csg = ns.Union(ns.Sphere(0.5), ns.Box(), world)
    assert world.primitives == [s, csg]                       # operands live under the private CSGRoot, not the world
    flat = world.flatten()
    assert flat.n_world == 2 and len(flat.records) == 4
    with pytest.raises(ValueError):
        ns.Sphere(-1)
    with pytest.raises(ValueError):
        ns.Box(ns.Point3D(1, 0, 0), ns.Point3D(0, 1, 1))


def test_observer_defaults_and_slicing(ns):
    world = ns.World()
    cam = ns.PinholeCamera((64, 32), parent=world)
    assert (cam.spectral_bins, cam.spectral_rays, cam.min_wavelength, cam.max_wavelength) == (15, 1, 375.0, 740.0)
    assert cam.pixel_samples == 100 and cam.ray_max_depth == 500
    # pinhole.pyx:76-83: neither pipelines nor sampler given -> an RGB pipeline sampled adaptively on it; one given -> the other's plain default
    assert isinstance(cam.pipelines[0], ns.RGBPipeline2D) and isinstance(cam.frame_sampler, ns.RGBAdaptiveSampler2D)
    assert cam.frame_sampler.pipeline is cam.pipelines[0]
    spectral = ns.PinholeCamera((8, 8), parent=world, pipelines=[ns.SpectralRadiancePipeline2D()])
    assert isinstance(spectral.frame_sampler, ns.FullFrameSampler2D)
    sampled = ns.PinholeCamera((8, 8), parent=world, frame_sampler=ns.FullFrameSampler2D())
    assert isinstance(sampled.pipelines[0], ns.RGBPipeline2D)
    cam.spectral_bins, cam.spectral_rays = 7, 3
    assert [(s.offset, s.bins) for s in cam._slice_spectrum()] == [(0, 2), (2, 3), (5, 2)]
    with pytest.raises(ValueError):
        cam.fov = 180
    tasks = ns.FullFrameSampler2D().generate_tasks((3, 2))
    assert sorted(tasks) == [(x, y) for x in range(3) for y in range(2)]
    # the reference's engine names are accepted where a render engine is expected
    serial, multi = ns.SerialEngine(), ns.MulticoreEngine(processes=8)
    assert isinstance(serial, ns.RenderEngine) and serial.rng == "stream" and serial.worker_count() == 1
    assert multi.rng == "philox" and multi.processes == 8 and ns.MulticoreEngine(seed=5).seed == 5
    cam.render_engine = serial


def test_rsm_reader_and_obj_io(ns, golden, tmp_path):
    """Mesh I/O: RSM files written by the compiled reference load into the same mesh — KD-tree taken from
    the file node for node, re-saved bytes identical — and OBJ export/import round-trips geometry."""
    import io
    from source_amd import scenes
    from source_amd.primitive import MeshData
    g = golden("f03_kd")
    builders = {"cube": lambda: scenes.cube_mesh(), "fan500": lambda: scenes.fan_mesh()}
    for name in ("cube", "sphere8", "blob24", "fan500"):
        blob = g[name].tobytes()
        data = MeshData.from_file(io.BytesIO(blob))
        assert np.array_equal(data.face_normals, g[name + "_face_normals"])          # recomputed on load, mesh.pyx:1012-1013
        out = io.BytesIO()
        data.save(out)
        assert out.getvalue() == blob, name
        if name in builders:                                                          # same tree as building from the arrays
            v, t = builders[name]()
            built = ns.Mesh(v, t, smoothing=False)
            assert np.array_equal(built.data.kd.nodes, data.kd.nodes) and np.array_equal(built.data.kd.items, data.kd.items)
            assert np.array_equal(built.data.vertices, data.vertices) and np.array_equal(built.data._triangles, data._triangles)
    mesh = ns.Mesh.from_file(io.BytesIO(g["cube"].tobytes()), transform=ns.translate(0, 0, 1), name="cube")
    assert mesh.name == "cube" and mesh.data.kd.max_depth >= 0
    with pytest.raises(ValueError):
        MeshData.from_file(io.BytesIO(b"XYZ" + g["cube"].tobytes()[3:]))
    # OBJ: export then import reproduces vertices (to %e precision), triangles and normals
    v, t = scenes.displaced_sphere(8)
    normals = scenes.vertex_normals(v, t)
    src = ns.Mesh(v, np.hstack([t, t]), normals, smoothing=True, name="blob")
    path = str(tmp_path / "blob.obj")
    ns.export_obj(src, path)
    back = ns.import_obj(path, smoothing=True)
    assert np.array_equal(back.data.triangles, src.data.triangles)
    assert np.allclose(back.data.vertices, src.data.vertices, rtol=1e-6, atol=0) and back.data.vertex_normals.shape == normals.shape
    assert np.allclose(back.data.vertex_normals, src.data.vertex_normals, rtol=2e-6, atol=1e-7)
    plain = ns.Mesh(v, t, smoothing=False)
    ns.export_obj(plain, path)
    scaled = ns.import_obj(path, scaling=2.0, smoothing=False)
    assert scaled.data.vertex_normals is None and np.allclose(scaled.data.vertices, 2.0 * plain.data.vertices, rtol=1e-6)
    with open(path, "a") as f:
        f.write("f 1 2 3 4\n")
    with pytest.raises(ValueError):
        ns.import_obj(path)


def test_combine_arrays_matches_scalar_law():
    """The vectorised combine_samples used by RGBPipeline2D.finalise() performs the scalar restatement's operations element by
    element: identical bits for every branch of the law (empty, single, many samples on either side)."""
    import numpy as np
    from source_amd.device import combine_arrays, combine_scalar
    rng = np.random.RandomState(3)
    n = 4000
    mx, my = rng.normal(size=n), rng.normal(size=n)
    vx, vy = rng.uniform(0, 2, n), rng.uniform(0, 2, n)
    nx, ny = rng.randint(0, 7, n), rng.randint(0, 7, n)
    nx[:200], ny[:200] = rng.randint(100, 100000, 200), rng.randint(1, 5, 200)
    vx, vy = np.where(nx > 1, vx, 0.0), np.where(ny > 1, vy, 0.0)
    m, v, c = combine_arrays(mx, vx, nx, my, vy, ny)
    for i in range(n):
        want = combine_scalar(mx[i], vx[i], int(nx[i]), my[i], vy[i], int(ny[i]))
        assert (want[0], want[1], want[2]) == (m[i], v[i], c[i]), i


def test_material_primitive_lists_and_world_accelerator(ns):
    """primitive.pyx:62-96: a material knows the primitives it coats (the setter moves the primitive between the lists);
    world.pyx:59-70: World.accelerator exists, defaults to the device accelerator and accepts any Accelerator-shaped object."""
    from source_amd.core.scenegraph import Accelerator, HipAccelerator
    world = ns.World()
    a, b = ns.AbsorbingSurface(), ns.AbsorbingSurface()
    s = ns.Sphere(0.5, world, material=a)
    assert a.primitives == [s] and b.primitives == []
    s.material = b
    assert a.primitives == [] and b.primitives == [s]
    with pytest.raises(TypeError):
        s.material = 3
    assert isinstance(world.accelerator, HipAccelerator)
    with pytest.raises(TypeError):
        world.accelerator = object()

    class Recorder(Accelerator):
        def __init__(self):
            self.built = None

        def build(self, primitives):
            self.built = list(primitives)

        def hit(self, ray):
            return "hit"

        def contains(self, point):
            return ["contains"]
    world.accelerator = Recorder()
    assert world._rebuild_accelerator is True
    # geometry edits of a primitive without a World bump its private version (device.scene_for_primitive keys its cache on it)
    free = ns.Sphere(0.5)
    v0 = free._geometry_version
    free.radius = 0.75
    assert free._geometry_version == v0 + 1


def test_philox_counters_advance_between_passes(ns):
    """Consecutive observe() passes of one observer must not reuse Philox (pixel, sample) counters (the reference's engines draw
    fresh numbers every pass); assigning engine.sample_offset restarts the count from the assigned value."""
    from source_amd import scenes
    world = scenes.build_c2(ns, n=8)[0]
    cam, pipe = scenes.c2_camera(ns, world, (8, 8), spp=5, bins=3)
    eng = ns.HipEngine(rng="philox", seed=1)
    seen = []
    cam._render_slice_device = lambda *a, **k: seen.append(cam._pass_offset)     # no GPU here: record what a pass would draw
    cam.render_engine = eng
    world.build_accelerator = lambda force=False: None
    cam.observe(); cam.observe()
    cam.pixel_samples = 7
    cam.observe(); cam.observe()
    assert seen == [0, 5, 10, 17]
    eng.sample_offset = 100                                                      # explicit placement (multi-GPU shards): restart there
    cam.observe(); cam.observe()
    assert seen[4:] == [100, 107]
    eng.sample_offset = 100
    cam.observe()
    assert seen[6] == 114                                                        # same value re-assigned: the count carries on
    assert ns.HipEngine().timing is False


def test_bench_roofline_helpers():
    """bench.py's ceiling arithmetic on a recorded counter set (profiles/: rocprofv3 --pmc summary of a bench run): every ceiling has the
    shape {achieved, peak, unit, frac}, the binding one is instruction issue, every fraction is below 1 (the contract's
    algorithmic-HBM line is not among the ceilings), and the dominant instantiation of a multi-kernel workload is the one that did
    the vector work."""
    import glob
    import json
    import bench
    path = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_c3.json")))[-1]
    table = json.load(open(path))["kernels"]
    name, c = bench.kernel_counters(table, "k_render_trace")
    assert name.startswith("k_render_trace<false, 0, 1")
    ceil, binding = bench.ceilings(c, 27.3, 17000.0)
    assert binding == "valu_issue" and 0.5 < ceil["valu_issue"]["frac"] < 1.0
    assert all({"achieved", "peak", "unit", "frac"} <= set(v) for v in ceil.values())
    assert all(v["frac"] < 1.0 for v in ceil.values()) and ceil["hbm_measured"]["frac"] < 0.5 and 0.5 < ceil["l2"]["hit_rate"] < 1.0
    assert 0.5 < ceil["valu_issue"]["lane_utilisation"] < 1.0
    if "salu_issue" in ceil:
        assert 0.1 < ceil["salu_issue"]["frac"] < ceil["valu_issue"]["frac"]
    assert "l1_vector_cache" not in bench.ceilings(c, 27.3, None)[0]      # (path-traced workloads have no algorithmic-bytes line)
    c4 = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_c4.json")))
    if c4:
        name, _ = bench.kernel_counters(json.load(open(c4[-1]))["kernels"], "k_render_trace")
        assert name.startswith("k_render_trace<true, 1")          # the fast CSG pass, not the (idle) redo pass
    b, per = bench.ray_bytes(dict(nodes=27, items=8, tris=3, prims=5), 1)
    assert b == 56 + 16 * 27 + 4 * 8 + 48 * 3 + 216 * 5 + 24


def test_frame_segments_of_the_reduce_scatter():
    """rsx_frame_segment: the segment arithmetic of rsx_allreduce_frame (rsx_comm.hpp) — the segments of W ranks tile [0, n) in rank
    order without gaps or overlap whatever n % W is, including frames shorter than the number of ranks."""
    from source_amd import distributed as D
    for n in (0, 1, 7, 8, 9, 10, 63, 64, 65, 1000003, 2048 * 2048 * 15):
        for w in (1, 2, 3, 4, 7, 8):
            segs = [D.frame_segment(n, w, r) for r in range(w)]
            at = 0
            for off, length in segs:
                assert off == at and length >= 0
                at += length
            assert at == n and D.frame_segment(n, w, w) == (n, 0)
            width = -(-n // w)
            assert all(length in (width, max(0, n - off)) for off, length in segs)       # ceil(n / W) each, the tail cut at n
            assert max(l for _, l in segs) - min(l for _, l in segs if l or n == 0) <= width
    assert D.frame_segment(10, 4, 3) == (9, 1) and D.frame_segment(7, 8, 7) == (7, 0)


def test_missing_rccl_is_an_error_code_not_a_crash(tmp_path):
    """librccl that cannot be loaded: rsx_comm_unique_id / rsx_comm_create must RETURN RSX_EUNSUPPORTED with the loader's message
    (bench.py falls back to another exchange on it) — round 2's form read dlerror() twice and dereferenced the second, null, answer."""
    import subprocess
    import sys
    code = """
import ctypes as C, sys
sys.path.insert(0, %r)
from source_amd import _lib
L = _lib.lib()
buf = (C.c_char * 128)()
rc = L.rsx_comm_unique_id(buf)
msg = L.rsx_last_error().decode()
h = C.c_void_p()
rc2 = L.rsx_comm_create(None, 2, 0, buf, C.byref(h))
print(rc, rc2, msg)
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RSX_RCCL_LIB=str(tmp_path / "no_such_librccl.so"))      # ($RSX_RCCL_LIB names THE library: no fallback to another copy)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-1500:]
    rc, rc2, msg = r.stdout.strip().split(" ", 2)
    assert int(rc) == -5 and int(rc2) in (-5, -1), (rc, rc2, msg)      # RSX_EUNSUPPORTED (the null ctx of the second call may be refused first)
    assert "librccl could not be loaded" in msg and "no_such_librccl.so" in msg


def test_spectrum_resampling_follows_the_reference(ns):
    """Spectrum.integrate / average / sample (spectrum.pyx:202-300): the samples are values at the bin centres, integrated as the
    piecewise-linear curve through them and extended by its end values; nothing is cached across in-place mutation."""
    s = ns.Spectrum(1.0, 4.0, 3)
    s.samples[:] = [1.0, 2.0, 3.0]
    assert s.integrate(1.0, 2.0) == 1.125                   # 0.5 * 1 (flat to the first centre) + 0.5 * (1 + 1.5) * 0.5
    assert s.average(1.0, 2.0) == 1.125
    assert np.allclose(s.sample(1.0, 4.0, 3), [1.125, 2.0, 2.875])
    first = s.sample(1.5, 3.5, 2).copy()
    s.mul_scalar(2.0)
    assert np.allclose(s.sample(1.5, 3.5, 2), 2.0 * first)  # (a cached resampling would have gone stale here)
    assert s.integrate(0.25, 0.75) == 1.0                   # below the first centre: its value, nearest-neighbour
    with pytest.raises(ValueError):
        s.integrate(2.0, 2.0)
    from source_amd.optical.spectral import SpectralFunction
    assert ns.Spectrum.evaluate is SpectralFunction.evaluate   # (the reference's Spectrum defines no evaluate() of its own)


def test_stale_library_is_refused_at_load_time(tmp_path):
    """A binary that lacks a declared entry point must stop the load with the symbol's name ($RSX_LIB or not) — only the explicit list
    of newer, optional entry points may be absent from an A/B build."""
    import subprocess
    import sys
    src = tmp_path / "stub.c"
    src.write_text("int rsx_version(void) { return 0; }\n")
    so = tmp_path / "librsx_stub.so"
    subprocess.check_call(["gcc", "-shared", "-fPIC", str(src), "-o", str(so)])
    code = "import sys; sys.path.insert(0, %r)\nfrom source_amd import _lib\ntry:\n    _lib.lib()\n    print('loaded')\nexcept _lib.RsxError as e:\n    print('refused', e)\n" % \
        os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, RSX_LIB=str(so)), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.startswith("refused") and "does not export" in r.stdout, r.stdout + r.stderr[-800:]


def test_passes_per_call_bookkeeping(ns):
    """HipEngine(passes_per_call=K): K passes per observe() — the Philox counters an observe() hands out and the rays one library
    call may carry both scale with K (the frames themselves: tests/test_gpu_parity.py::test_passes_per_call_equals_separate_passes)."""
    from source_amd import scenes
    with pytest.raises(ValueError):
        ns.HipEngine(passes_per_call=0)
    world = scenes.build_c2(ns, n=8)[0]
    cam, pipe = scenes.c2_camera(ns, world, (64, 32), spp=3, bins=4)
    cam.frame_sampler = ns.RectFrameSampler2D()
    cam.MAX_RAYS_PER_CALL = 64 * 32 * 3 * 2
    cam.render_engine = ns.HipEngine(passes_per_call=1)
    assert len(cam._pieces(cam._generate_tasks(), world)) == 1
    cam.render_engine = ns.HipEngine(passes_per_call=4)
    pieces = cam._pieces(cam._generate_tasks(), world)
    assert len(pieces) == 2 and pieces[0]["rect"] == (0, 0, 32, 32)    # (4 passes x 3 spp per pixel: half the frame per call)
    # sample_stride interleaves the ranks' single passes; K passes per call would run into the next rank's counters: refused, loudly
    cam.render_engine = ns.HipEngine(passes_per_call=4, sample_stride=2, sample_offset=3)
    with pytest.raises(ValueError, match="sample_stride"):
        cam._check_counter_layout(cam.render_engine)
    cam._check_counter_layout(ns.HipEngine(passes_per_call=4))
    cam._check_counter_layout(ns.HipEngine(sample_stride=2, sample_offset=3))
    # ... the documented way: every call placed by hand, blocks of K x spp counters per rank
    from source_amd import distributed as D
    K, spp, N = 4, 3, 2
    blocks = sorted((D.rank_sample_offset(call, rank, N, K * spp), K * spp) for call in range(3) for rank in range(N))
    assert all(blocks[i][0] + blocks[i][1] == blocks[i + 1][0] for i in range(len(blocks) - 1)) and blocks[0][0] == 0


def test_auto_batch_eligibility(ns):
    """HipEngine.auto_batch: which observe() calls may be held back and batched (PinholeCamera._lazy_signature) — small passes of a
    rectangle of pixels into accumulating spectral pipelines, closed-form materials, the engine in its default Philox form — and
    what ends a batch (a different signature)."""
    from source_amd import scenes
    world = scenes.build_c2(ns, n=8)[0]
    cam, pipe = scenes.c2_camera(ns, world, (64, 32), spp=2, bins=4)
    cam.frame_sampler = ns.RectFrameSampler2D()
    cam._slices = cam._slice_spectrum()
    cam._initialise_pipelines(cam.min_wavelength, cam.max_wavelength, cam.spectral_bins, cam._slices, True)
    tasks = cam._generate_tasks()
    sig = cam._lazy_signature(tasks, ns.HipEngine(), world)
    assert sig is not None
    eng = ns.HipEngine()
    assert cam._lazy_signature(tasks, eng, world) == cam._lazy_signature(tasks, eng, world)
    for kw in (dict(auto_batch=False), dict(rng="stream"), dict(fused=False), dict(timing=True), dict(host_materials=True), dict(passes_per_call=2),
               dict(sample_stride=2)):
        assert cam._lazy_signature(tasks, ns.HipEngine(**kw), world) is None, kw
    shuffled = [(ix, iy) for iy in range(32) for ix in range(64)]
    import random
    random.Random(1).shuffle(shuffled)
    assert cam._lazy_signature(shuffled, eng, world) == cam._lazy_signature(tasks, eng, world)       # a full rectangle, whatever its order
    assert cam._lazy_signature(shuffled[:-5], eng, world) is None                                      # a picked list: rendered now
    cam.pixel_samples = 64
    assert cam._lazy_signature(tasks, eng, world) is None                                              # a pass that fills its units by itself
    cam.pixel_samples = 3
    assert cam._lazy_signature(tasks, eng, world) is None
    cam.pixel_samples = 2
    a = cam._lazy_signature(tasks, eng, world)
    cam.transform = ns.translate(0, 0, -1)
    assert cam._lazy_signature(tasks, eng, world) != a                                                  # a moved camera starts a new batch
    b = cam._lazy_signature(tasks, eng, world)
    cam.pipelines = [ns.SpectralRadiancePipeline2D(accumulate=False)]
    cam._initialise_pipelines(cam.min_wavelength, cam.max_wavelength, cam.spectral_bins, cam._slices, True)
    assert cam._lazy_signature(tasks, eng, world) is None                                              # nothing to accumulate into
    cam.pipelines = [ns.RGBPipeline2D()]
    cam._initialise_pipelines(cam.min_wavelength, cam.max_wavelength, cam.spectral_bins, cam._slices, True)
    assert cam._lazy_signature(tasks, eng, world) is None
    cam.pipelines = [pipe]
    assert cam._lazy_signature(tasks, eng, world) == b
    world._primitives[0].material = ns.Lambert(ns.ConstantSF(0.5))
    assert cam._lazy_signature(tasks, eng, world) is None                                              # path passes render by themselves
    import os
    os.environ["RSX_AUTO_BATCH"] = "0"
    try:
        assert ns.HipEngine().auto_batch is False and ns.HipEngine(auto_batch=True).auto_batch is True
    finally:
        del os.environ["RSX_AUTO_BATCH"]


def test_balance_tiles_converges_on_measured_times(ns):
    """distributed.balance_tiles (what the reference's task queue does dynamically, workflow.py:201-251): every rank times its own tile,
    the times are shared, the cuts move to equal measured cost. Driven here by stand-in observers whose observe() takes time in
    proportion to a known cost density over its tile — two 'ranks' stepped in lock-step: the cuts end where the density says, stay
    multiples of 8 columns, and the engine's sample offset and the frame sampler are left as documented."""
    import time
    from source_amd import distributed as D
    from source_amd.optical.observer import RectFrameSampler2D
    nx, ny, W = 256, 8, 2
    density = [1.0 if x < 128 else 3.0 for x in range(nx)]            # the right half costs three times the left
    clock = [0.0]

    class Engine:
        sample_offset = 7

    class Cam:
        def __init__(self):
            self.pixels, self.render_engine, self.pipelines = (nx, ny), Engine(), []
            self.frame_sampler = RectFrameSampler2D()
            self.cost = 0.0

        def observe(self):
            x0, _, x1, _ = self.frame_sampler.rect
            self.cost = sum(density[x0:x1]) * 1e-4
            time.sleep(self.cost)

    cams = [Cam() for _ in range(W)]
    shared = {}

    # the ranks run one after the other; allgather hands every rank the times of a full round: rank 0's round k is computed first, so the
    # exchange is emulated by recording per-(round, rank) times from the known density instead of the wall clock of the other process
    def make_allgather(rank):
        calls = [0]

        def allgather(value):
            k = calls[0]
            calls[0] += 1
            out = []
            for r in range(W):
                b = shared.setdefault(("bounds", k), None) or [(nx * q) // W for q in range(W)] + [nx]
                out.append(sum(density[b[r]:b[r + 1]]) * 1e-4)
            return out
        return allgather

    bounds = None
    for rank in range(W):
        # every rank computes the same cuts from the same shared times: emulate by iterating the rule itself
        b = [(nx * q) // W for q in range(W)] + [nx]
        for k in range(4):
            shared[("bounds", k)] = list(b)
            times = [sum(density[b[r]:b[r + 1]]) * 1e-4 for r in range(W)]
            b = [int(v) for v in D.rebalance_bounds(b, times, nx)]
        got = D.balance_tiles(cams[rank], rank, W, make_allgather(rank), lambda: None, rounds=4, min_seconds=0.0)
        assert got == b, (got, b)
        bounds = got
        assert cams[rank].render_engine.sample_offset == 7
        assert cams[rank].frame_sampler.rect == D.tile_rect(rank, W, nx, ny, bounds)
    assert bounds[0] == 0 and bounds[-1] == nx and all(v % 8 == 0 for v in bounds)
    cost = [sum(density[bounds[r]:bounds[r + 1]]) for r in range(W)]
    assert max(cost) / (sum(cost) / W) < 1.06, (bounds, cost)          # equal cost within a quantum of 8 columns (total 512: 256 each)
    assert D.balance_tiles(cams[0], 0, 1, None, None) == [0, nx]


def test_coherent_task_order(ns):
    """PinholeCamera._coherent_tasks (what a Philox pass does with a task list): a shuffled list of every pixel of a rectangle becomes
    that rectangle; any other list keeps its pixels, each once, in 8 x 8 tile order; the conversion is cached per list."""
    import random
    from source_amd.optical.observer import RectTasks
    cam = ns.PinholeCamera((40, 24))
    full = [(ix, iy) for iy in range(24) for ix in range(40)]
    random.Random(3).shuffle(full)
    out = cam._coherent_tasks(full)
    assert isinstance(out, RectTasks) and out.rect == (0, 0, 40, 24) and cam._coherent_tasks(full) is out
    window = [(ix, iy) for iy in range(5, 17) for ix in range(8, 31)]
    random.Random(4).shuffle(window)
    assert cam._coherent_tasks(window).rect == (8, 5, 31, 17)
    some = [t for t in full if (t[0] * 7 + t[1] * 3) % 5]
    out = cam._coherent_tasks(some)
    assert not isinstance(out, RectTasks) and sorted(map(tuple, out.tolist())) == sorted(some)
    tiles = [(int(x) >> 3, int(y) >> 3) for x, y in out]
    assert all(tiles[i] == tiles[i - 1] or tiles[i] not in tiles[:i] for i in range(1, len(tiles)))     # a tile's pixels are contiguous
    twice = full + [(3, 3)]
    assert not isinstance(cam._coherent_tasks(twice), RectTasks)     # (a pixel listed twice: not a rectangle)


def test_host_walk_single_rays_against_golden_vectors(orc, ns, golden, m70k):
    """rsx_hit_host / rsx_contains_host (csrc/rsx_hostwalk.cpp: the host side of World.hit / World.contains for single rays, SURVEY.md 8b
    "n = 1 -> CPU lib") against the reference's own vectors, no GPU involved: the mesh ray classes of F04 (ids, t, u, v, w, exiting,
    geometry; vertex / edge rays, origins on the surface, max_distance short / exact / long), the analytic primitives of F05 (first
    root, full geometry, contains), the edge worlds of F11 (empty world, coincident primitives, t == max_distance, axis-parallel
    grazing rays), the CSG worlds of F06 / F07 — and against the pinned oracle on a mixed world of instanced meshes and analytic
    primitives, on CSG solids with mesh operands and on the prism scene."""
    import raysets
    from source_amd import scenes, _lib
    from source_amd._flatten import FlatScene
    from source_amd.device import HostScene
    import test_oracle_golden as T
    # F04: the 69 432-triangle mesh as a one-primitive world
    g = golden("f04_mesh")
    mesh, v, t = m70k
    host = HostScene(FlatScene([mesh]))
    sets = {"grid": raysets.pinhole_grid(96), "outside": raysets.random_outside(6000, 41), "outside_raw": raysets.random_outside(2000, 42, unit=False),
            "interior": raysets.random_interior(4000, 43), "vertices": raysets.through_vertices(v, 4000, 44), "edges": raysets.along_edges(v, t, 3000, 45),
            "axis": raysets.axis_aligned(3000, 46, 0.1, v)}
    for name, (o, d, m) in sets.items():
        r = host.hit_batch(o, d, m, geometry=True)
        hit = T._check_mesh(r, g, name)
        if name + "_extra" in g:
            assert eq(r["geom"][hit], g[name + "_extra"][hit]), name
    T._check_mesh(host.hit_batch(g["surf_o"], g["surf_d"]), g, "surf")
    o, d, _ = sets["outside"]
    T._check_mesh(host.hit_batch(o[g["maxd_idx"]], d[g["maxd_idx"]], g["maxd_m"]), g, "maxd")
    assert eq(host.contains_batch(raysets.points(4000, 49, 0.1))[:, 0], g["contains"])
    # F05: sphere / box / cylinder, transformed and not: World.hit of a one-primitive world = the primitive's first root
    g = golden("f05_primitives")
    for k, (name, prim) in enumerate(T._prims(ns).items()):
        host = HostScene(FlatScene([prim]))
        o, d, m = raysets.primitive_rays(3000, 70 + k)
        r = host.hit_batch(o, d, m, geometry=True)
        ref = g[name][:, 0, :]                               # Primitive.hit's first root: t, exiting, hit, inside, outside, normal
        hit = r["prim"] >= 0
        # World.hit puts the BoundPrimitive gate and the tree's bounds in front of Primitive.hit (boundprimitive.pyx:42-51): some rays the bare
        # cylinder answers (along its axis, from its caps) never reach it in a world. Where the world answers, it is the primitive's root ...
        assert not np.isnan(ref[hit, 0]).any() and hit.sum() > 0.5 * (~np.isnan(ref[:, 0])).sum(), name
        assert eq(r["t"][hit], ref[hit, 0]) and eq(r["exiting"][hit], ref[hit, 1]) and eq(r["geom"][hit], ref[hit, 2:]), name
        # ... and which rays it answers is what the pinned oracle's World.hit says
        w = orc.hit_batch(FlatScene([prim]), o, d, m, geometry=True)
        assert eq(r["prim"], w["prim"]) and eq(r["t"][hit], w["t"][hit]) and eq(r["geom"][hit], w["geom"][hit]), name
        assert eq(host.contains_batch(raysets.points(2000, 90 + k, 1.2))[:, 0], g[name + "_contains"]), name
    # F11: the edge semantics of SURVEY.md Appendix B
    g = golden("f11_edges")
    for name, (world, prims) in scenes.build_edge_worlds(ns).items():
        host = HostScene(world.flatten())
        o, d, m = scenes.edge_rays(name)
        T._check_world(host.hit_batch(o, d, m, geometry=True), g[name + "_idx"], g[name + "_rec"])
        assert eq(host.contains_batch(np.concatenate([o, o + 0.25 * d])), g[name + "_contains"]), name
    # instanced meshes + analytic primitives in one world, against the pinned oracle (smoothing normals, transforms, the world tree)
    world = ns.World()
    vs, ts = scenes.displaced_sphere(24, radius=0.5)
    ts6 = np.concatenate([ts, ts], axis=1)                     # (vertex-normal indices = vertex indices)
    base = ns.Mesh(vs, ts6, normals=scenes.vertex_normals(vs, ts), smoothing=True, closed=True, parent=world, transform=ns.translate(-0.7, 0.1, 0.2) * ns.rotate(20, 30, 40))
    base.instance(parent=world, transform=ns.translate(0.8, -0.2, 0.1) * ns.rotate(-50, 10, 5))
    ns.Sphere(0.4, world, ns.translate(0.0, 0.9, -0.3))
    ns.Box(ns.Point3D(-2, -2, -1.6), ns.Point3D(2, 2, -1.5), world)
    ns.Cylinder(0.25, 1.1, world, ns.translate(0.1, -0.9, -0.4) * ns.rotate(0, 70, 0))
    flat = world.flatten()
    host = HostScene(flat)
    o, d, m = raysets.scene_rays(20000, 321, 6.0, 2.2)
    a, b = host.hit_batch(o, d, m, geometry=True), orc.hit_batch(flat, o, d, m, geometry=True)
    assert eq(a["prim"], b["prim"]) and (a["prim"] >= 0).mean() > 0.2
    hit = b["prim"] >= 0
    for key in ("t", "exiting", "tri", "uvw", "geom"):
        assert eq(a[key][hit], b[key][hit]), key
    pts = raysets.points(6000, 322, 1.5)
    assert eq(host.contains_batch(pts), orc.contains_batch(flat, pts))
    # single-ray form through preallocated buffers = the batch form
    one = host.hit_one(*o[0], *d[0], float(m[0]))
    assert (one is None) == (a["prim"][0] < 0) and (one is None or (one[0] == a["prim"][0] and one[1] == a["t"][0] and eq(one[5], a["geom"][0])))
    # CSG worlds: the host walk runs the reference's stream merge (csg.pyx:132-234) — the demos/csg.py tree (F06) and the mixed world with CSG
    # solids and instances (F07) against the reference's own vectors, ids / t / exiting / geometry and contains()
    g = golden("f06_csg")
    world, prims = scenes.build_csg_demo(ns)
    host = HostScene(world.flatten())
    o, d, m = raysets.scene_rays(12000, 101, 9.0, 4.5)
    og, dg, mg = raysets.pinhole_grid(64, (0.0, 0.0, -4.0), 75.0)
    o, d, m = np.concatenate([o, og]), np.concatenate([d, dg]), np.concatenate([m, mg])
    T._check_world(host.hit_batch(o, d, m, geometry=True), g["world_idx"], g["world_rec"])
    assert eq(host.contains_batch(raysets.points(4000, 102, 4.5)), g["contains"])
    g = golden("f07_world")
    world, prims = scenes.build_mixed(ns)
    host = HostScene(world.flatten())
    o, d, m = raysets.scene_rays(20000, 111, 6.0, 2.2)
    r = host.hit_batch(o, d, m, geometry=True)
    T._check_world(r, g["idx"], g["rec"])
    assert eq(host.contains_batch(raysets.points(6000, 112, 2.0)), g["contains"])
    # ... the prism scene (configs[4]: nested Intersect / Subtract of boxes) and mesh operands (the mesh's next_intersection stream inside
    # the merge, MeshIntersection extras handed up) against the pinned oracle
    v24, t24 = scenes.displaced_sphere(24, radius=0.5)
    world = ns.World()
    ns.Subtract(ns.Mesh(v24, t24, smoothing=False, transform=ns.translate(0.1, 0, 0)), ns.Box(ns.Point3D(-0.3, -1, -1), ns.Point3D(0.25, 1, 1)), world,
                ns.translate(0, 0.1, 0.2) * ns.rotate(10, 20, 30), ns.AbsorbingSurface())
    ns.Intersect(ns.Sphere(0.45, transform=ns.translate(0.2, 0, 0)), ns.Mesh(v24, t24, smoothing=False), world, ns.translate(1.5, 0, 0), ns.AbsorbingSurface())
    for w, rays, pts in ((world, raysets.scene_rays(20000, 301, 4.0, 1.2), raysets.points(3000, 302, 1.5)),
                         (scenes.build_prism(ns)[0], raysets.scene_rays(20000, 303, 3.0, 1.0), raysets.points(3000, 304, 1.0))):
        flat = w.flatten()
        host = HostScene(flat)
        a, b = host.hit_batch(*rays, geometry=True), orc.hit_batch(flat, *rays, geometry=True)
        assert eq(a["prim"], b["prim"]) and (a["prim"] >= 0).mean() > 0.05
        hit = b["prim"] >= 0
        for key in ("t", "exiting", "tri", "uvw", "geom"):
            assert eq(a[key][hit], b[key][hit]), key
        assert eq(host.contains_batch(pts), orc.contains_batch(flat, pts))


def test_binding_under_stock_raysect_fixture(ns, golden, m70k):
    """F19: tests/golden/bind_reference.py ran integration/raysect_hip.py — `HipAccelerator(raysect.core.acceleration.Accelerator)` over
    rsx_host_scene_create / rsx_hit_host_one / rsx_contains_host — under the COMPILED reference in the build container: every World.hit /
    World.contains of the F04 / F05 / F06 / F07 / F11 ray sets asked through stock `World` objects, once with Raysect's KDTree accelerator and
    once with librsx behind `world.accelerator`, all Intersection fields equal bit for bit (0 differences recorded). The fixture keeps a
    SHA-256 of what both accelerators answered; here the same rays go through this repository's own API mirror and host walk and must
    reproduce it — sign of zero included."""
    import hashlib
    import raysets
    from source_amd import scenes
    from source_amd._flatten import FlatScene
    from source_amd.device import HostScene
    import test_oracle_golden as T
    g = golden("f19_binding")

    def check(name, flat, o, d, m, pts):
        host = HostScene(flat)
        r = host.hit_batch(o, d, m, geometry=True)
        inside = host.contains_batch(pts)
        table = np.zeros((len(pts), max(1, flat.n_world)), dtype=np.uint8)
        table[:, :flat.n_world] = inside
        hit = r["prim"] >= 0
        rec = np.column_stack([r["t"], r["exiting"].astype(np.float64), r["geom"]])
        h = hashlib.sha256()
        h.update(r["prim"].astype(np.int32).tobytes())
        h.update(np.ascontiguousarray(rec[hit]).tobytes())
        h.update(table.tobytes())
        n_rays, n_hits, hit_diff, cont_diff, n_pts = (int(x) for x in g[name])
        assert (hit_diff, cont_diff) == (0, 0), name                               # what the container run recorded
        assert (n_rays, n_hits, n_pts) == (len(o), int(hit.sum()), len(pts)), name
        assert h.digest() == g[name + "_sha"].tobytes(), name

    mesh, v, t = m70k
    sets = [raysets.random_outside(3000, 41), raysets.through_vertices(v, 1500, 44), raysets.along_edges(v, t, 1000, 45), raysets.axis_aligned(1000, 46, 0.1, v)]
    check("f04_mesh_world", FlatScene([mesh]), *(np.concatenate([s[k] for s in sets]) for k in range(3)), raysets.points(1500, 49, 0.1))
    for k, (name, prim) in enumerate(T._prims(ns).items()):
        world = ns.World()
        prim.parent = world                                                        # (in a world, as the container run had them: to_local() is then the transform's inverse)
        check("f05_" + name, world.flatten(), *raysets.primitive_rays(3000, 70 + k), raysets.points(1500, 90 + k, 1.2))
    o, d, m = raysets.scene_rays(6000, 101, 9.0, 4.5)
    og, dg, mg = raysets.pinhole_grid(48, (0.0, 0.0, -4.0), 75.0)
    check("f06_csg_demo", scenes.build_csg_demo(ns)[0].flatten(), np.concatenate([o, og]), np.concatenate([d, dg]), np.concatenate([m, mg]), raysets.points(3000, 102, 4.5))
    check("f07_mixed_world", scenes.build_mixed(ns)[0].flatten(), *raysets.scene_rays(12000, 111, 6.0, 2.2), raysets.points(4000, 112, 2.0))
    for name, (world, prims) in scenes.build_edge_worlds(ns).items():
        o, d, m = scenes.edge_rays(name)
        check("f11_" + name, world.flatten(), o, d, m, np.concatenate([o, o + 0.25 * d]))


def test_host_builders_team_is_sized_to_what_the_process_may_use(tmp_path):
    """rsx_host_team_size (csrc/rsx_host.cpp): the OpenMP team of the host KD / mesh builders is the smaller of the CPU affinity and the cgroup
    CPU quota — not every hardware thread the container sees (the cause of round 5's "80 ms stalls after a second world": 256 spinning
    workers on a 16-core quota got the process throttled) — and RSX_HOST_THREADS overrides it. Checked in fresh processes (the value is
    read once)."""
    import subprocess
    import sys
    code = "import sys; sys.path.insert(0, %r); from source_amd import _lib; print(_lib.lib().rsx_host_team_size())" % os.path.dirname(_lib._HERE)
    plain = int(subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120).stdout.strip().splitlines()[-1])
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else max(1, int(q) // int(period))
    except OSError:
        pass
    assert 1 <= plain <= len(os.sched_getaffinity(0))
    if quota is not None:
        assert plain <= quota
    forced = int(subprocess.run([sys.executable, "-c", code], env=dict(os.environ, RSX_HOST_THREADS="3"), capture_output=True, text=True,
                                timeout=120).stdout.strip().splitlines()[-1])
    assert forced == 3


def test_concurrent_ranks_build_the_library_once(tmp_path):
    """`bench.py --gpus N` brings every rank through __graft_entry__.build_librsx(): with a stale or missing library they must not write
    the same file at once. One builds (under a lock, into a temporary name), the others wait and find it fresh."""
    import shutil
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fake = tmp_path / "root"
    (fake / "source_amd" / "csrc").mkdir(parents=True)
    (fake / "include").mkdir()
    for f in os.listdir(os.path.join(root, "source_amd", "csrc")):
        shutil.copy(os.path.join(root, "source_amd", "csrc", f), fake / "source_amd" / "csrc" / f)
    shutil.copy(os.path.join(root, "include", "rsx.h"), fake / "include" / "rsx.h")
    cc = tmp_path / "fakecc"
    cc.write_text('#!/bin/bash\nout="${@: -1}"\necho start >> %s/log\nsleep 1\necho data > "$out"\n' % tmp_path)
    cc.chmod(0o755)
    code = "import __graft_entry__ as g; g.ROOT = %r; print(g.build_librsx())" % str(fake)
    env = dict(os.environ, HIPCC=str(cc), PYTHONPATH=root)
    procs = [subprocess.Popen([sys.executable, "-c", code], env=env, cwd=root, stdout=subprocess.PIPE) for _ in range(4)]
    outs = [p.communicate(timeout=60)[0].decode().strip() for p in procs]
    assert all(p.returncode == 0 for p in procs)
    assert all(o.endswith("librsx.so") for o in outs)
    assert (tmp_path / "log").read_text().count("start") == 1
    assert sorted(f for f in os.listdir(fake / "source_amd" / "lib") if not f.endswith(".lock")) == ["librsx.so"]
