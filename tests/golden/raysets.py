"""
Deterministic ray / point sets shared by the golden-fixture generator and the parity tests.
Pure numpy; nothing here touches the reference.
"""
import numpy as np


def _unit(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def pinhole_grid(n, origin=(0.0, 0.05, -0.35), fov=40.0):
    """n x n grid of unit directions through a virtual image plane (no jitter)."""
    half = np.tan(np.radians(fov) * 0.5)
    c = (np.arange(n) + 0.5) / n * 2 - 1
    px, py = np.meshgrid(-c * half, -c * half, indexing="ij")
    d = _unit(np.stack([px, py, np.ones_like(px)], axis=-1).reshape(-1, 3))
    o = np.broadcast_to(np.asarray(origin, dtype=np.float64), d.shape).copy()
    return o, d, np.full(len(d), np.inf)


def random_outside(n, seed, r_origin=0.3, r_target=0.1, unit=True):
    rng = np.random.RandomState(seed)
    o = _unit(rng.normal(size=(n, 3))) * r_origin
    t = rng.normal(size=(n, 3))
    t = _unit(t) * r_target * rng.uniform(0, 1, (n, 1)) ** (1 / 3)
    d = t - o
    if unit:
        d = _unit(d)
    return o, d, np.full(n, np.inf)


def random_interior(n, seed, r=0.03):
    rng = np.random.RandomState(seed)
    o = _unit(rng.normal(size=(n, 3))) * r * rng.uniform(0, 1, (n, 1))
    d = _unit(rng.normal(size=(n, 3)))
    return o, d, np.full(n, np.inf)


def through_vertices(vertices, n, seed):
    """Rays aimed exactly at f32 vertex positions (vertex / shared-edge tie-breaks)."""
    rng = np.random.RandomState(seed)
    idx = rng.randint(0, len(vertices), n)
    tgt = vertices[idx].astype(np.float64)
    o = _unit(rng.normal(size=(n, 3))) * 0.4
    d = tgt - o
    half = n // 2
    d[:half] = _unit(d[:half])          # second half left unnormalised on purpose
    return o, d, np.full(n, np.inf)


def along_edges(vertices, triangles, n, seed):
    """Rays running exactly along triangle edges (grazing / zero barycentric cases)."""
    rng = np.random.RandomState(seed)
    t = triangles[rng.randint(0, len(triangles), n)]
    a = vertices[t[:, 0]].astype(np.float64)
    b = vertices[t[:, 1]].astype(np.float64)
    d = b - a
    ok = np.linalg.norm(d, axis=1) > 0
    a, b, d = a[ok], b[ok], d[ok]
    o = a - 3.0 * d
    return o, d, np.full(len(o), np.inf)


def axis_aligned(n, seed, extent=0.1, vertices=None):
    """Directions with exact zero components; a share of origins sit exactly on vertex coordinates."""
    rng = np.random.RandomState(seed)
    axis = rng.randint(0, 3, n)
    sign = rng.choice([-1.0, 1.0], n)
    d = np.zeros((n, 3))
    d[np.arange(n), axis] = sign
    o = rng.uniform(-extent, extent, (n, 3))
    if vertices is not None:
        k = n // 3
        vi = rng.randint(0, len(vertices), k)
        o[:k] = vertices[vi].astype(np.float64)
    o[np.arange(n), axis] = -sign * 0.5
    return o, d, np.full(n, np.inf)


def scene_rays(n, seed, r_origin=6.0, r_target=2.5, frac_inside=0.25, frac_axis=0.1):
    """Mixed ray set for world-level scenes of a few units extent."""
    rng = np.random.RandomState(seed)
    o, d, m = random_outside(n, seed + 1, r_origin, r_target)
    k = int(n * frac_inside)
    o[:k] = rng.uniform(-1.5, 1.5, (k, 3))
    d[:k] = _unit(rng.normal(size=(k, 3)))
    ka = int(n * frac_axis)
    oa, da, _ = axis_aligned(ka, seed + 2, extent=1.5)
    oa[np.arange(ka), np.argmax(np.abs(da), axis=1)] *= 12.0
    o[k:k + ka], d[k:k + ka] = oa, da
    m = np.full(n, np.inf)
    fin = rng.uniform(size=n) < 0.2
    m[fin] = rng.uniform(0.5, 12.0, fin.sum())
    return o, d, m


def primitive_rays(n, seed, scale=1.0):
    """Rays for single analytic primitives of ~unit size at the origin: random, tangent-ish,
    inside-origin, axis-parallel and zero-component directions."""
    rng = np.random.RandomState(seed)
    o, d, m = random_outside(n, seed, 3.0 * scale, 1.2 * scale)
    k = n // 5
    o[:k] = rng.uniform(-0.4, 0.4, (k, 3)) * scale            # inside origins
    d[:k] = _unit(rng.normal(size=(k, 3)))
    oa, da, _ = axis_aligned(k, seed + 7, extent=0.6 * scale)
    ax = np.argmax(np.abs(da), axis=1)
    oa[np.arange(k), ax] *= 6.0 * scale
    o[k:2 * k], d[k:2 * k] = oa, da
    # exactly axis-parallel along z from lattice points (cylinder parallel case, box faces)
    g = np.linspace(-0.6, 0.6, 13) * scale
    gx, gy = np.meshgrid(g, g, indexing="ij")
    kk = gx.size
    o[2 * k:2 * k + kk] = np.stack([gx.ravel(), gy.ravel(), np.full(kk, -3.0 * scale)], axis=-1)
    d[2 * k:2 * k + kk] = [0.0, 0.0, 1.0]
    fin = rng.uniform(size=n) < 0.25
    m[fin] = rng.uniform(0.2, 5.0, fin.sum()) * scale
    return o, d, m


def points(n, seed, extent=2.0):
    rng = np.random.RandomState(seed)
    return rng.uniform(-extent, extent, (n, 3))
