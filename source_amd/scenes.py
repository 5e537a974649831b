"""
Deterministic synthetic geometry and scene builders for the benchmark configs.

The Stanford bunny assets are not shipped with the reference snapshot
(/root/reference/.MISSING_LARGE_BLOBS) and there is no network, so every config
of BASELINE.json is rendered on a seeded stand-in mesh generated here (numpy
only, no reference code involved).

The scene builders take a *namespace* argument ``ns`` exposing the Raysect class
names (World, Mesh, Box, Sphere, ..., translate, rotate, Point3D, ...).  Passing
the real ``raysect`` modules builds the scene in the reference (golden fixture
generation, tests/golden/make_golden.py); passing ``source_amd.api`` builds the
identical scene in this framework.  That is the parity harness: one scene
description, two frameworks.
"""
import numpy as np


def displaced_sphere(n=132, radius=0.08, amplitude=0.25, seed=20250905, modes=6):
    """
    Closed UV-sphere with (n+1) latitude rows x 2n longitude columns whose radius
    is displaced by a seeded low-frequency field (stand-in for the ~70k-triangle
    Stanford bunny: n=132 -> 35 112 vertices, 69 696 triangles of which the pole
    fans are degenerate and are removed by the Mesh 'tolerant' filter).

    Returns (vertices float32[nv,3], triangles int32[nt,3]). Face winding gives
    outward normals (right-hand rule), as Mesh(closed=True) requires.
    """
    rng = np.random.RandomState(seed)
    a = rng.uniform(-1.0, 1.0, modes)
    p = rng.randint(1, 5, modes)
    q = rng.randint(0, 5, modes)
    ph = rng.uniform(0, 2 * np.pi, modes)
    ps = rng.uniform(0, 2 * np.pi, modes)

    rows, cols = n + 1, 2 * n
    theta = np.linspace(0.0, np.pi, rows)[:, None]            # polar angle
    phi = (np.arange(cols) * (2.0 * np.pi / cols))[None, :]   # azimuth
    field = np.zeros((rows, cols))
    for k in range(modes):
        # sin(theta) weighting makes the field azimuth-independent at the poles
        az = np.cos(q[k] * phi + ps[k]) if q[k] else np.ones_like(phi)
        w = np.sin(theta) if q[k] else 1.0
        field += a[k] * np.sin(p[k] * theta + ph[k]) * az * w
    field /= modes
    r = radius * (1.0 + amplitude * field)

    x = r * np.sin(theta) * np.cos(phi)
    y = r * np.cos(theta) * np.ones_like(phi)
    z = r * np.sin(theta) * np.sin(phi)
    vertices = np.stack([x, y, z], axis=-1).reshape(-1, 3).astype(np.float32)

    i = np.arange(rows - 1)[:, None]
    j = np.arange(cols)[None, :]
    j1 = (j + 1) % cols
    v00 = (i * cols + j).ravel()
    v01 = (i * cols + j1).ravel()
    v10 = ((i + 1) * cols + j).ravel()
    v11 = ((i + 1) * cols + j1).ravel()
    t1 = np.stack([v00, v01, v11], axis=-1)
    t2 = np.stack([v00, v11, v10], axis=-1)
    triangles = np.empty((2 * t1.shape[0], 3), dtype=np.int32)
    triangles[0::2] = t1
    triangles[1::2] = t2
    return vertices, triangles


def vertex_normals(vertices, triangles):
    """Area-weighted per-vertex normals (float32), for the smoothing path."""
    v = vertices.astype(np.float64)
    n = np.zeros_like(v)
    e1 = v[triangles[:, 1]] - v[triangles[:, 0]]
    e2 = v[triangles[:, 2]] - v[triangles[:, 0]]
    fn = np.cross(e1, e2)
    for k in range(3):
        np.add.at(n, triangles[:, k], fn)
    ln = np.linalg.norm(n, axis=1)
    ln[ln == 0] = 1.0
    return (n / ln[:, None]).astype(np.float32)


def cube_mesh(h=0.5):
    """12-triangle cube, outward winding."""
    v = np.array([[-h, -h, -h], [h, -h, -h], [h, h, -h], [-h, h, -h],
                  [-h, -h, h], [h, -h, h], [h, h, h], [-h, h, h]], dtype=np.float32)
    t = np.array([[0, 2, 1], [0, 3, 2], [4, 5, 6], [4, 6, 7],
                  [0, 1, 5], [0, 5, 4], [2, 3, 7], [2, 7, 6],
                  [1, 2, 6], [1, 6, 5], [0, 4, 7], [0, 7, 3]], dtype=np.int32)
    return v, t


def fan_mesh(n=500, seed=3):
    """Open fan of n triangles sharing one apex: a single high-valence vertex, so
    many triangles land in the same KD leaves (stress for leaf-order tie-breaks)."""
    rng = np.random.RandomState(seed)
    ang = np.linspace(0, 2 * np.pi, n + 1)
    rim = np.stack([np.cos(ang), np.sin(ang), 0.2 * rng.uniform(-1, 1, n + 1)], axis=-1)
    v = np.concatenate([[[0.0, 0.0, 0.5]], rim]).astype(np.float32)
    t = np.stack([np.zeros(n, dtype=np.int32), np.arange(1, n + 1), np.arange(2, n + 2)], axis=-1).astype(np.int32)
    return v, t


# ----------------------------------------------------------------------------------------------
# scene builders shared by reference (golden generation) and this framework
# ----------------------------------------------------------------------------------------------

def build_c2(ns, n=132, smoothing=False, with_normals=False):
    """
    BASELINE.json configs[1]: one ~70k-triangle mesh, pinhole camera, primary rays only.
    Materials are closed-form (no daughter rays): debug Light on the mesh, a uniform
    emitter on the enclosing box. Camera placement follows demos/materials/bunny.py:30,74.
    Returns (world, mesh, box).
    """
    verts, tris = displaced_sphere(n)
    world = ns.World()
    normals = None
    if with_normals:
        normals = vertex_normals(verts, tris)
        tris = np.concatenate([tris, tris], axis=1)
    mesh = ns.Mesh(verts, tris, normals, smoothing=smoothing, closed=True, parent=world,
                   transform=ns.translate(0, 0.08, 0) * ns.rotate(165, 0, 0),
                   material=ns.Light(ns.Vector3D(-1, -1, 1), 1.0, ns.ConstantSF(1.0)))
    box = ns.Box(ns.Point3D(-2, -2, -2), ns.Point3D(2, 2, 2), world,
                 material=ns.UniformSurfaceEmitter(ns.ConstantSF(1.0), 0.1))
    return world, mesh, box


def build_flat(ns, n=512):
    """HBM-stress variant of configs[1] (SURVEY.md §8d "M1M-flat"): ONE mesh of 1 048 576 triangles (n = 512; 0.45 GB of nodes and
    leaf records on the device — far more than the 32 MB of L2), same materials and camera geometry as build_c2."""
    return build_c2(ns, n=n)


def c2_camera(ns, world, pixels=(1024, 1024), spp=1, bins=15):
    pipe = ns.SpectralRadiancePipeline2D()
    cam = ns.PinholeCamera(pixels, fov=45, parent=world, pipelines=[pipe],
                           frame_sampler=ns.FullFrameSampler2D(),
                           transform=ns.translate(0, 0.16, -0.4) * ns.rotate(0, -12, 0))
    cam.pixel_samples = spp
    cam.spectral_bins = bins
    cam.spectral_rays = 1
    cam.quiet = True
    return cam, pipe


def build_c3(ns, n=132, grid=(5, 3)):
    """configs[2]: instanced ~1M-triangle scene: grid of Mesh.instance() copies + floor box."""
    verts, tris = displaced_sphere(n)
    world = ns.World()
    base = None
    meshes = []
    k = 0
    for gx in range(grid[0]):
        for gy in range(grid[1]):
            tr = ns.translate(0.22 * (gx - (grid[0] - 1) / 2), 0.09 + 0.2 * gy, 0.25 * gy) * ns.rotate(23.0 * k, 0, 0)
            mat = ns.Light(ns.Vector3D(-1, -1, 1), 1.0, ns.ConstantSF(1.0))
            if base is None:
                base = ns.Mesh(verts, tris, smoothing=False, closed=True, parent=world, transform=tr, material=mat)
                meshes.append(base)
            else:
                meshes.append(base.instance(parent=world, transform=tr, material=mat))
            k += 1
    floor = ns.Box(ns.Point3D(-3, -0.1, -3), ns.Point3D(3, 0.0, 3), world,
                   material=ns.UniformSurfaceEmitter(ns.ConstantSF(1.0), 0.3))
    sky = ns.Box(ns.Point3D(-10, -10, -10), ns.Point3D(10, 10, 10), world,
                 material=ns.UniformSurfaceEmitter(ns.ConstantSF(1.0), 0.05))
    return world, meshes, floor, sky


def c3_camera(ns, world, pixels=(2048, 2048), spp=64, bins=15):
    pipe = ns.SpectralRadiancePipeline2D()
    cam = ns.PinholeCamera(pixels, fov=60, parent=world, pipelines=[pipe],
                           frame_sampler=ns.FullFrameSampler2D(),
                           transform=ns.translate(0, 0.35, -0.9) * ns.rotate(0, -8, 0))
    cam.pixel_samples = spp
    cam.spectral_bins = bins
    cam.spectral_rays = 1
    cam.quiet = True
    return cam, pipe


def build_csg_demo(ns, emitter=True):
    """configs[3]: the CSG tree of demos/csg.py:24-40 with closed-form materials
    (primary-ray parity; the demo's Dielectric glass is a 'next' row)."""
    world = ns.World()

    def obj():
        cyl_x = ns.Cylinder(1, 4.2, transform=ns.rotate(90, 0, 0) * ns.translate(0, 0, -2.1))
        cyl_y = ns.Cylinder(1, 4.2, transform=ns.rotate(0, 90, 0) * ns.translate(0, 0, -2.1))
        cyl_z = ns.Cylinder(1, 4.2, transform=ns.rotate(0, 0, 0) * ns.translate(0, 0, -2.1))
        cube = ns.Box(ns.Point3D(-1.5, -1.5, -1.5), ns.Point3D(1.5, 1.5, 1.5))
        sphere = ns.Sphere(2.0)
        return sphere, ns.Subtract(cube, ns.Union(ns.Union(cyl_x, cyl_y), cyl_z))

    def light():
        return ns.Light(ns.Vector3D(-1, -1, 1), 1.0, ns.ConstantSF(1.0))

    prims = []
    for tx, ty, yaw, pitch in [(-2.1, 2.1, 30, -20), (2.1, 2.1, -30, -20), (2.1, -2.1, -30, 20), (-2.1, -2.1, 30, 20)]:
        s, sub = obj()
        prims.append(ns.Intersect(s, sub, world, ns.translate(tx, ty, 2.5) * ns.rotate(yaw, pitch, 0), light()))
    s1 = ns.Sphere(1.0, transform=ns.translate(0, 0, 1.0 - 0.01))
    s2 = ns.Sphere(0.5, transform=ns.translate(0, 0, -0.5 + 0.01))
    prims.append(ns.Intersect(s1, s2, world, ns.translate(0, 0, -3.6) * ns.rotate(50, 50, 0), light()))
    prims.append(ns.Box(ns.Point3D(-50, -50, 50), ns.Point3D(50, 50, 50.1), world,
                        material=ns.UniformSurfaceEmitter(ns.ConstantSF(1.0), 0.6)))
    prims.append(ns.Box(ns.Point3D(-100, -100, -100), ns.Point3D(100, 100, 100), world,
                        material=ns.UniformSurfaceEmitter(ns.ConstantSF(1.0), 0.1)))
    return world, prims


def csg_camera(ns, world, pixels=(1024, 1024), spp=16, bins=15):
    pipe = ns.SpectralRadiancePipeline2D()
    cam = ns.PinholeCamera(pixels, fov=75, parent=world, pipelines=[pipe],
                           frame_sampler=ns.FullFrameSampler2D(),
                           transform=ns.translate(0, 0, -4) * ns.rotate(0, 0, 0))
    cam.pixel_samples = spp
    cam.spectral_bins = bins
    cam.spectral_rays = 1
    cam.quiet = True
    return cam, pipe


def build_mixed(ns):
    """World-level stress scene (fixture F7): instanced meshes + analytic + CSG with
    overlapping AABBs and three coincident spheres (last-in-leaf tie rule)."""
    verts, tris = displaced_sphere(24, radius=0.5)
    world = ns.World()
    prims = []
    absorb = ns.AbsorbingSurface
    m0 = ns.Mesh(verts, tris, smoothing=False, parent=world, transform=ns.translate(-1.0, 0.2, 0.3) * ns.rotate(20, 30, 10), material=absorb())
    prims.append(m0)
    prims.append(m0.instance(parent=world, transform=ns.translate(0.4, -0.3, 0.9) * ns.rotate(-40, 10, 70), material=absorb()))
    prims.append(ns.Sphere(0.45, world, ns.translate(0.5, 0.5, -0.2), absorb()))
    prims.append(ns.Sphere(0.45, world, ns.translate(0.5, 0.5, -0.2), absorb()))
    prims.append(ns.Sphere(0.45, world, ns.translate(0.5, 0.5, -0.2), absorb()))
    prims.append(ns.Box(ns.Point3D(-0.3, -0.4, -0.5), ns.Point3D(0.3, 0.4, 0.5), world, ns.translate(-0.2, -0.9, 0.1) * ns.rotate(15, 25, 35), absorb()))
    prims.append(ns.Cylinder(0.3, 1.1, world, ns.translate(1.2, -0.5, 0.0) * ns.rotate(70, 20, 0), absorb()))
    a = ns.Sphere(0.5, transform=ns.translate(0.0, 0.0, 0.2))
    b = ns.Box(ns.Point3D(-0.3, -0.3, -0.3), ns.Point3D(0.3, 0.3, 0.3))
    prims.append(ns.Subtract(a, b, world, ns.translate(-0.5, 1.0, -0.6) * ns.rotate(10, 20, 30), absorb()))
    c = ns.Cylinder(0.25, 0.9, transform=ns.translate(0, 0, -0.45))
    d = ns.Sphere(0.4)
    prims.append(ns.Union(c, d, world, ns.translate(1.3, 0.9, 0.7) * ns.rotate(-30, 45, 0), absorb()))
    prims.append(ns.Box(ns.Point3D(-4, -4, -4), ns.Point3D(4, 4, 4), world, material=absorb()))
    return world, prims


def build_edge_worlds(ns):
    """Small worlds that pin the edge semantics of SURVEY.md Appendix B (fixture F11): an empty world (B.16), coincident
    primitives (B.19: last registered wins), t == max_distance on analytic and mesh surfaces (B.18), origins on / inside
    surfaces, axis-parallel rays along box faces / edges / corners and along a cylinder axis."""
    worlds = {}
    worlds["empty"] = (ns.World(), [])
    w = ns.World()
    prims = [ns.Sphere(1.0, w, ns.translate(0, 0, 3), ns.AbsorbingSurface()) for _ in range(3)]
    prims.append(ns.Box(ns.Point3D(-1, -1, 2), ns.Point3D(1, 1, 4), w, material=ns.AbsorbingSurface()))     # touches the spheres at z = 2 and 4
    worlds["coincident"] = (w, prims)
    w = ns.World()
    v, t = cube_mesh(0.5)
    prims = [ns.Sphere(0.5, w, ns.translate(0, 0, 3), ns.AbsorbingSurface()),
             ns.Mesh(v, t, smoothing=False, parent=w, transform=ns.translate(2, 0, 4.5), material=ns.AbsorbingSurface()),
             ns.Box(ns.Point3D(-1, -1, -1), ns.Point3D(1, 1, 1), w, ns.translate(-4, 0, 3), ns.AbsorbingSurface()),
             ns.Cylinder(0.5, 2.0, w, ns.translate(4, 0, 2), ns.AbsorbingSurface())]
    worlds["limits"] = (w, prims)
    return worlds


def edge_rays(name):
    """(origin, direction, max_distance) arrays for build_edge_worlds()[name]: hand-placed rays, all exactly representable."""
    inf = float("inf")
    up = lambda x: float(np.nextafter(x, inf))
    dn = lambda x: float(np.nextafter(x, -inf))
    z, x, y = (0.0, 0.0, 1.0), (1.0, 0.0, 0.0), (0.0, 1.0, 0.0)
    if name == "empty":
        rays = [((0, 0, 0), z, inf), ((1, 2, 3), x, 5.0), ((0, 0, 0), (0.6, 0.0, 0.8), inf)]
    elif name == "coincident":
        rays = [((0, 0, 0), z, inf), ((0, 0, 0), z, 2.0), ((0, 0, 0), z, dn(2.0)), ((0, 0, 3), z, inf), ((0, 0, 3), (0.0, 0.0, -1.0), inf),
                ((0, 0, 2), z, inf), ((0, 0, 4), z, inf), ((0.5, 0.5, 0), z, inf), ((1, 0, 0), z, inf), ((1, 1, 0), z, inf),
                ((0, 0, 10), (0.0, 0.0, -1.0), inf), ((-5, 0, 3), x, inf), ((0, -5, 3), y, inf), ((0, 0, 3), (0.6, 0.0, 0.8), inf)]
    else:
        rays = [((0, 0, 0), z, 2.5), ((0, 0, 0), z, dn(2.5)), ((0, 0, 0), z, up(2.5)), ((0, 0, 0), z, 0.0), ((0, 0, 0), z, 1e-300),
                ((0, 0, 0), z, 3.5), ((0, 0, 0), z, dn(3.5)),                                   # sphere: exit root at 3.5
                ((2, 0, 0), z, 4.0), ((2, 0, 0), z, up(4.0)), ((2, 0, 0), z, dn(4.0)), ((2, 0, 0), z, inf),   # mesh face at t = 4
                ((2, 0, 4.5), z, inf), ((2, 0, 4.0), z, inf), ((2, 0, 5.0), z, inf), ((2.5, 0, 0), z, inf), ((2.5, 0.5, 0), z, inf),
                ((-4, 0, 0), z, 2.0), ((-4, 0, 0), z, dn(2.0)), ((-3, 0, 0), z, inf), ((-3, 1, 0), z, inf), ((-3, 1, 2), z, inf),
                ((-4, 0, 3), z, inf), ((-4, 0, 3), x, inf), ((-4, 0, 3), (0.0, -1.0, 0.0), inf), ((-5, -1, 3), y, inf), ((-5, 0, 2), x, inf),
                ((4, 0, 0), z, inf), ((4, 0, 0), z, 2.0), ((4, 0, 0), z, dn(2.0)), ((4.5, 0, 0), z, inf), ((4.25, 0, 3), z, inf),
                ((4, 0, 3), z, inf), ((4, 0, 3), x, inf), ((3, 0, 3), x, inf), ((3, 0, 2), x, inf), ((3, 0, 4), x, inf), ((4, -3, 3), y, 2.5),
                ((4, -3, 3), y, dn(2.5)), ((0, 0, 3), z, inf), ((0, 0, 2.5), z, inf), ((0, 0, 2.5), (0.0, 0.0, -1.0), inf), ((0, 0, 3.5), z, inf)]
    o = np.array([r[0] for r in rays], dtype=np.float64)
    d = np.array([r[1] for r in rays], dtype=np.float64)
    m = np.array([r[2] for r in rays], dtype=np.float64)
    return o, d, m


def edge_camera(ns, world, pixels, spp, bins, mask=None):
    pipe = ns.SpectralRadiancePipeline2D()
    sampler = ns.FullFrameSampler2D(mask) if mask is not None else ns.FullFrameSampler2D()
    cam = ns.PinholeCamera(pixels, fov=50, parent=world, pipelines=[pipe], frame_sampler=sampler, transform=ns.translate(0, 0.16, -0.4) * ns.rotate(0, -12, 0))
    cam.pixel_samples, cam.spectral_bins, cam.spectral_rays, cam.quiet = spp, bins, 1, True
    return cam, pipe


def build_volumes(ns, enclosed=True):
    """Volume-emission scene (fixture F12): overlapping emitting sphere / box / cylinder / CSG lens with transparent boundaries,
    a NullMaterial shell, an opaque emitter behind them and an emitting box that contains the camera itself."""
    world = ns.World()
    sf = ns.InterpolatedSF([300, 490, 510, 590, 610, 800], np.array([0.0, 0.1, 1.0, 0.7, 0.2, 0.4]))
    prims = [
        ns.Sphere(0.5, world, ns.translate(0.0, 0.0, 2.0), ns.UniformVolumeEmitter(ns.ConstantSF(1.0), 0.7)),
        ns.Box(ns.Point3D(-0.3, -0.3, -0.3), ns.Point3D(0.3, 0.3, 0.3), world, ns.translate(0.45, 0.1, 2.1) * ns.rotate(20, 30, 0),
               ns.UniformVolumeEmitter(sf, 1.3)),
        ns.Cylinder(0.2, 1.4, world, ns.translate(-0.6, -0.5, 1.6) * ns.rotate(0, 70, 0), ns.UniformVolumeEmitter(ns.ConstantSF(0.5), 2.0)),
        ns.Sphere(0.9, world, ns.translate(0.0, 0.0, 2.0), ns.NullMaterial()),
        ns.Intersect(ns.Sphere(0.5, transform=ns.translate(0, 0, 0.3)), ns.Sphere(0.5, transform=ns.translate(0, 0, -0.3)), world,
                     ns.translate(-0.2, 0.55, 1.5), ns.UniformVolumeEmitter(sf, 0.9)),
        ns.Box(ns.Point3D(-1.0, -1.0, 3.2), ns.Point3D(1.0, 0.2, 3.4), world, material=ns.UniformSurfaceEmitter(ns.ConstantSF(1.0), 0.25)),
        ns.Box(ns.Point3D(-4, -4, -4), ns.Point3D(4, 4, 4), world, material=ns.UniformVolumeEmitter(ns.ConstantSF(1.0), 0.01)),
    ]
    if enclosed:                                            # without the shell most paths end in a miss: zero spectrum for the last
        prims.append(ns.Box(ns.Point3D(-6, -6, -6), ns.Point3D(6, 6, 6), world, material=ns.AbsorbingSurface()))   # segment, volumes of the earlier ones kept
    return world, prims


def volumes_camera(ns, world, pixels=(48, 40), spp=3, bins=6):
    pipe = ns.SpectralRadiancePipeline2D()
    cam = ns.PinholeCamera(pixels, fov=55, parent=world, pipelines=[pipe], frame_sampler=ns.FullFrameSampler2D(), transform=ns.translate(0.05, 0.1, 0.0))
    cam.pixel_samples, cam.spectral_bins, cam.spectral_rays, cam.quiet = spp, bins, 1, True
    return cam, pipe


def build_lambert(ns, with_volume=True, csg=True):
    """Diffuse inter-reflection scene (fixture F13): an open-fronted room of Lambert walls with spectrally different reflectivities,
    an emitting ceiling panel, a Lambert sphere, a smooth-shaded Lambert mesh, a Lambert CSG solid, and — so that scattering,
    transparent boundaries and volume emission meet on one path — a glowing volume and a NullMaterial shell."""
    world = ns.World()
    red = ns.InterpolatedSF([300, 560, 600, 800], np.array([0.08, 0.1, 0.75, 0.8]))
    green = ns.InterpolatedSF([300, 480, 520, 580, 620, 800], np.array([0.1, 0.12, 0.7, 0.65, 0.1, 0.08]))
    white = ns.ConstantSF(0.7)
    P = ns.Point3D
    prims = [
        ns.Box(P(-1.0, -1.05, 0.0), P(1.0, -1.0, 2.0), world, material=ns.Lambert(white)),        # floor
        ns.Box(P(-1.0, 1.0, 0.0), P(1.0, 1.05, 2.0), world, material=ns.Lambert(white)),          # ceiling
        ns.Box(P(-1.0, -1.0, 2.0), P(1.0, 1.0, 2.05), world, material=ns.Lambert(white)),         # back
        ns.Box(P(-1.05, -1.0, 0.0), P(-1.0, 1.0, 2.0), world, material=ns.Lambert(red)),          # left
        ns.Box(P(1.0, -1.0, 0.0), P(1.05, 1.0, 2.0), world, material=ns.Lambert(green)),          # right
        ns.Box(P(-0.4, 0.98, 0.6), P(0.4, 0.999, 1.4), world, material=ns.UniformSurfaceEmitter(ns.ConstantSF(1.0), 8.0)),   # light
        ns.Sphere(0.3, world, ns.translate(-0.45, -0.7, 1.3), ns.Lambert(ns.ConstantSF(0.9))),
    ]
    if csg:
        prims.append(ns.Subtract(ns.Box(P(-0.25, -0.25, -0.25), P(0.25, 0.25, 0.25)), ns.Sphere(0.3, transform=ns.translate(0.2, 0.2, -0.2)), world,
                                 ns.translate(0.45, -0.75, 0.9) * ns.rotate(25, 0, 0), ns.Lambert(green)))
    else:
        prims.append(ns.Box(P(-0.25, -0.25, -0.25), P(0.25, 0.25, 0.25), world, ns.translate(0.45, -0.75, 0.9) * ns.rotate(25, 0, 0), ns.Lambert(green)))
    v, t = displaced_sphere(10, radius=0.25)
    nv = vertex_normals(v, t)
    prims.append(ns.Mesh(v, np.concatenate([t, t], axis=1), normals=nv, smoothing=True, parent=world, transform=ns.translate(0.1, -0.2, 1.5), material=ns.Lambert(red)))
    if with_volume:
        prims.append(ns.Sphere(0.2, world, ns.translate(0.3, 0.3, 1.0), ns.UniformVolumeEmitter(ns.ConstantSF(1.0), 0.6)))
        prims.append(ns.Sphere(0.35, world, ns.translate(-0.3, 0.2, 0.8), ns.NullMaterial()))
    return world, prims


def lambert_camera(ns, world, pixels=(24, 20), spp=4, bins=5, extinction=(0.1, 2, 12)):
    pipe = ns.SpectralRadiancePipeline2D()
    cam = ns.PinholeCamera(pixels, fov=50, parent=world, pipelines=[pipe], frame_sampler=ns.FullFrameSampler2D(), transform=ns.translate(0.0, 0.0, -1.9))
    cam.pixel_samples, cam.spectral_bins, cam.spectral_rays, cam.quiet = spp, bins, 1, True
    cam.ray_extinction_prob, cam.ray_extinction_min_depth, cam.ray_max_depth = extinction
    cam.ray_importance_sampling = False                     # multiple importance sampling is the next scope row
    return cam, pipe


def build_glass(ns, unit_transmission=False):
    """Refraction scene (fixture F14): dispersive glass (Sellmeier index, absorbing transmission) as a sphere, a rotated block, a CSG
    lens and a small smooth mesh, over a Lambert floor, lit by an emitting panel and a striped back wall so that refraction is visible.
    ``unit_transmission`` makes every glass perfectly clear: no pow() in the volume pass, so device frames can be compared bit for bit."""
    world = ns.World()
    P = ns.Point3D
    bk7 = ns.Sellmeier(1.03961212, 0.231792344, 1.01046945, 6.00069867e-3, 2.00179144e-2, 1.03560653e2)
    tint = ns.ConstantSF(1.0) if unit_transmission else ns.InterpolatedSF([300, 450, 550, 650, 800], np.array([0.35, 0.6, 0.9, 0.7, 0.4]))
    clear = ns.ConstantSF(1.0) if unit_transmission else ns.ConstantSF(0.8)
    stripes = ns.InterpolatedSF([300, 480, 500, 600, 620, 800], np.array([0.1, 0.15, 1.0, 0.9, 0.2, 0.1]))
    prims = [
        ns.Box(P(-2.0, -1.05, -0.5), P(2.0, -1.0, 3.0), world, material=ns.Lambert(ns.ConstantSF(0.6))),               # floor
        ns.Box(P(-2.0, -1.0, 3.0), P(2.0, 2.0, 3.05), world, material=ns.UniformSurfaceEmitter(stripes, 0.8)),          # back wall, emitting
        ns.Box(P(-0.6, 1.6, 0.6), P(0.6, 1.62, 1.8), world, material=ns.UniformSurfaceEmitter(ns.ConstantSF(1.0), 6.0)),   # panel
        ns.Sphere(0.4, world, ns.translate(-0.7, -0.6, 1.4), ns.Dielectric(bk7, tint)),
        ns.Box(P(-0.3, -0.3, -0.3), P(0.3, 0.3, 0.3), world, ns.translate(0.75, -0.7, 1.6) * ns.rotate(30, 10, 0), ns.Dielectric(ns.ConstantSF(1.5), clear)),
        ns.Intersect(ns.Sphere(0.6, transform=ns.translate(0, 0, 0.45)), ns.Sphere(0.6, transform=ns.translate(0, 0, -0.45)), world,
                     ns.translate(0.0, 0.1, 1.0), ns.Dielectric(bk7, clear, transmission_only=True)),
        ns.Sphere(0.25, world, ns.translate(0.1, -0.75, 0.7), ns.Dielectric(ns.ConstantSF(1.33), tint, external_index=ns.ConstantSF(1.0))),
    ]
    v, t = displaced_sphere(8, radius=0.22)
    prims.append(ns.Mesh(v, np.concatenate([t, t], axis=1), normals=vertex_normals(v, t), smoothing=True, parent=world,
                         transform=ns.translate(-0.2, -0.75, 2.1), material=ns.Dielectric(ns.ConstantSF(1.7), clear)))
    return world, prims


def glass_camera(ns, world, pixels=(24, 20), spp=4, bins=6, spectral_rays=3, extinction=(0.1, 2, 20)):
    pipe = ns.SpectralRadiancePipeline2D()
    cam = ns.PinholeCamera(pixels, fov=50, parent=world, pipelines=[pipe], frame_sampler=ns.FullFrameSampler2D(), transform=ns.translate(0.0, -0.2, -1.6))
    cam.pixel_samples, cam.spectral_bins, cam.spectral_rays, cam.quiet = spp, bins, spectral_rays, True
    cam.ray_extinction_prob, cam.ray_extinction_min_depth, cam.ray_max_depth = extinction
    cam.ray_importance_sampling = False
    return cam, pipe


def build_cornell(ns):
    """A Cornell-box variant after the reference's demos/cornell_box.py (BASELINE.json configs[0]): five Lambert walls (white, red,
    green; smooth stand-ins for the measured reflectances), a ceiling light, a glass block and a glass sphere (Sellmeier N-BK7
    coefficients instead of the reference's library entry). The emitter carries importance 1, so with ray_importance_sampling on
    (the observer default) Lambert surfaces use the important-path / BSDF mixture."""
    world = ns.World()
    P = ns.Point3D
    wl = [400, 450, 500, 550, 600, 650, 700]
    white = ns.InterpolatedSF(wl, np.array([0.45, 0.74, 0.75, 0.74, 0.75, 0.72, 0.74]))
    green = ns.InterpolatedSF(wl, np.array([0.09, 0.10, 0.29, 0.37, 0.16, 0.12, 0.16]))
    red = ns.InterpolatedSF(wl, np.array([0.04, 0.06, 0.06, 0.06, 0.29, 0.61, 0.64]))
    light = ns.InterpolatedSF([400, 500, 600, 700], np.array([0.0, 8.0, 15.6, 18.4]))
    bk7 = ns.Sellmeier(1.03961212, 0.231792344, 1.01046945, 6.00069867e-3, 2.00179144e-2, 1.03560653e2)
    prims = [
        ns.Box(P(-1, -1, 0), P(1, 1, 0.02), world, ns.translate(0, 0, 1), ns.Lambert(white)),                           # back
        ns.Box(P(-1, -1, 0), P(1, 1, 0.02), world, ns.translate(0, -1, 0) * ns.rotate(0, -90, 0), ns.Lambert(white)),   # bottom
        ns.Box(P(-1, -1, 0), P(1, 1, 0.02), world, ns.translate(0, 1, 0) * ns.rotate(0, 90, 0), ns.Lambert(white)),     # top
        ns.Box(P(-1, -1, 0), P(1, 1, 0.02), world, ns.translate(1, 0, 0) * ns.rotate(-90, 0, 0), ns.Lambert(red)),      # left
        ns.Box(P(-1, -1, 0), P(1, 1, 0.02), world, ns.translate(-1, 0, 0) * ns.rotate(90, 0, 0), ns.Lambert(green)),    # right
        ns.Box(P(-0.4, -0.4, -0.01), P(0.4, 0.4, 0.0), world, ns.translate(0, 1, 0) * ns.rotate(0, 90, 0), ns.UniformSurfaceEmitter(light, 2)),
        ns.Box(P(-0.4, 0, -0.4), P(0.3, 1.4, 0.3), world, ns.translate(0.4, -1 + 1e-6, 0.4) * ns.rotate(30, 0, 0), ns.Dielectric(bk7, ns.ConstantSF(1.0))),
        ns.Sphere(0.4, world, ns.translate(-0.4, -0.6 + 1e-6, -0.4), ns.Dielectric(bk7, ns.ConstantSF(1.0))),
    ]
    return world, prims


def cornell_camera(ns, world, pixels=(256, 256), spp=1, bins=15, pipelines=None):
    pipe = ns.SpectralRadiancePipeline2D()
    cam = ns.PinholeCamera(pixels, parent=world, transform=ns.translate(0, 0, -3.3), pipelines=pipelines or [pipe], frame_sampler=ns.FullFrameSampler2D())
    cam.pixel_samples, cam.spectral_bins, cam.spectral_rays, cam.quiet = spp, bins, 1, True
    cam.ray_importance_sampling, cam.ray_important_path_weight = True, 0.25
    cam.ray_max_depth, cam.ray_extinction_min_depth, cam.ray_extinction_prob = 500, 3, 0.01
    return cam, pipe


def build_prism(ns):
    """The dispersive-prism scene after the reference's demos/prism.py (BASELINE.json configs[4]): an equilateral SF11 prism cut from
    boxes by two Subtracts, a slit light box (CSG) aimed at it, a curved N-BK7 stand with a white Lambert screen parented to it
    (nested CSG of boxes and cylinders), a Lambert floor and a fill light. Glasses are Sellmeier dielectrics with the published Schott
    coefficients and unit transmission (the reference takes them, with measured transmission tables, from its data library); the
    daylight spectrum is a seven-point D65. The prism's importance is 9, so Lambert surfaces aim most of their samples at it."""
    from math import tan, pi
    world = ns.World()
    P = ns.Point3D
    d65 = ns.InterpolatedSF([400, 450, 500, 550, 600, 650, 700], np.array([0.8275, 1.1701, 1.0935, 1.0405, 0.9001, 0.8003, 0.7161]))
    sf11 = ns.Sellmeier(1.73759695, 0.313747346, 1.89878101, 0.013188707, 0.0623068142, 155.23629)
    bk7 = ns.Sellmeier(1.03961212, 0.231792344, 1.01046945, 6.00069867e-3, 2.00179144e-2, 1.03560653e2)
    width, height = 0.06, 0.15
    half = width / 2
    mid = half * tan(60 / 180 * pi) / 2
    centre = ns.Box(P(-half * 1.001, 0, 0), P(half * 1.001, height, width))
    left = ns.Box(P(0, -height * 0.001, -width * 0.001), P(width, height * 1.001, 2 * width), transform=ns.translate(half, 0, 0) * ns.rotate(30, 0, 0))
    right = ns.Box(P(-width, -height * 0.001, -width * 0.001), P(0.0, height * 1.001, 2 * width), transform=ns.translate(-half, 0, 0) * ns.rotate(-30, 0, 0))
    prism = ns.Subtract(ns.Subtract(centre, left), right, world, ns.translate(0, 1e-6, -0.01) * ns.translate(0, 0, -mid), ns.Dielectric(sf11, ns.ConstantSF(1.0)))
    prism.material.importance = 9
    floor = ns.Box(P(-1000, -0.1, -1000), P(1000, 0, 1000), world, material=ns.Lambert())
    stand = ns.Intersect(ns.Box(P(-10, -10, -10), P(10, 10, 0)),
                         ns.Subtract(ns.Cylinder(0.21, 0.15), ns.Cylinder(0.20, 0.16, transform=ns.translate(0, 0, -0.005)), transform=ns.rotate(0, 90, 0)),
                         world, ns.translate(0.0, 1e-6, 0.0), ns.Dielectric(bk7, ns.ConstantSF(1.0)))
    screen = ns.Intersect(ns.Box(P(-10, -10, -10), P(10, 10, -0.015)),
                          ns.Subtract(ns.Cylinder(0.1999, 0.12, transform=ns.translate(0, 0, 0.015)), ns.Cylinder(0.1998, 0.13, transform=ns.translate(0, 0, 0.010)),
                                      transform=ns.rotate(0, 90, 0)),
                          world, ns.translate(0.0, 1e-6, 0.0), ns.Lambert(ns.ConstantSF(1.0)))
    box_t = ns.rotate(-35.5, 0, 0) * ns.translate(0.10, 0, 0) * ns.rotate(90, 0, 0)
    housing = ns.Subtract(ns.Box(P(-0.01, 0, -0.05), P(0.01, 0.15, 0.0)), ns.Box(P(-0.0015, 0.03, -0.045), P(0.0015, 0.12, 0.0001)), world, box_t,
                          ns.Lambert(ns.ConstantSF(0.1)))
    slit_light = ns.Box(P(-0.0015, 0.03, -0.045), P(0.0015, 0.12, -0.04), world, box_t, ns.UniformSurfaceEmitter(d65, 250))
    top_light = ns.Sphere(0.25, world, ns.translate(-1, 2, 1), ns.UniformSurfaceEmitter(d65, 5))
    return world, [prism, floor, stand, screen, housing, slit_light, top_light]


def prism_camera(ns, world, pixels=(1024, 1024), spp=4, bins=32, spectral_rays=32, pipelines=None):
    pipe = ns.SpectralRadiancePipeline2D()
    cam = ns.PinholeCamera(pixels, fov=45, parent=world, pipelines=pipelines or [pipe], frame_sampler=ns.FullFrameSampler2D(),
                           transform=ns.translate(0, 0.075, -0.05) * ns.rotate(180, -45, 0) * ns.translate(0, 0, -0.75))
    cam.pixel_samples, cam.spectral_bins, cam.spectral_rays, cam.quiet = spp, bins, spectral_rays, True
    cam.ray_importance_sampling, cam.ray_important_path_weight = True, 0.75
    cam.ray_max_depth, cam.ray_extinction_min_depth, cam.ray_extinction_prob = 500, 3, 0.01
    return cam, pipe
