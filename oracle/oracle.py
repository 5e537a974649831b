"""
ctypes wrapper of the CPU oracle (oracle/rsx_oracle.c). TEST INFRASTRUCTURE: imported only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg — never by source_amd.

The oracle consumes the same flattened scene description as librsx (include/rsx.h structs, mirrored as ctypes
Structures in source_amd/_lib.py), so one FlatScene feeds both the device path and its checker.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from source_amd import _lib as S

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None
vp = C.c_void_p


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.orc_kd_build.restype = vp
        L.orc_kd_build.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double]
        L.orc_kd_view.argtypes = [vp, C.POINTER(S.KDTree)]
        L.orc_kd_free.argtypes = [vp]
        L.orc_kd_serialise.restype = C.c_int64
        L.orc_kd_serialise.argtypes = [vp, vp, C.c_int64]
        L.orc_mesh_filter_triangles.restype = C.c_int32
        L.orc_mesh_filter_triangles.argtypes = [vp, vp, C.c_int32, C.c_int32]
        L.orc_mesh_face_normals.argtypes = [vp, vp, C.c_int32, C.c_int32, vp]
        L.orc_mesh_triangle_aabbs.argtypes = [vp, vp, C.c_int32, C.c_int32, vp]
        L.orc_mesh_world_bbox.argtypes = [vp, C.c_int32, vp, vp]
        L.orc_mt_seed_words.argtypes = [vp, vp, C.c_uint64]
        L.orc_mt_uniform.argtypes = [vp, C.c_int64, vp]
        L.orc_philox_uniform2.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, vp]
        L.orc_aabb_intersect.argtypes = [C.c_int64, vp, vp, vp, vp, vp]
        L.orc_hit_batch.argtypes = [C.POINTER(S.SceneDesc), C.c_int64, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, vp]
        L.orc_packet_counters.argtypes = [C.POINTER(S.SceneDesc), C.c_int64, C.c_int32, vp, vp, vp, vp, C.c_int]
        L.orc_prim_hit_batch.argtypes = [C.POINTER(S.SceneDesc), C.c_int32, C.c_int64, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int]
        L.orc_roots_batch.argtypes = [C.POINTER(S.SceneDesc), C.c_int32, C.c_int64, vp, vp, vp, C.c_int32, vp, vp, vp, vp]
        L.orc_contains_batch.argtypes = [C.POINTER(S.SceneDesc), C.c_int64, vp, vp]
        L.orc_prim_contains_batch.argtypes = [C.POINTER(S.SceneDesc), C.c_int32, C.c_int64, vp, vp]
        L.orc_add_samples.argtypes = [C.c_int64, vp, vp]
        L.orc_frame_combine.argtypes = [C.c_int64, vp, vp, vp, vp, vp, vp]
        L.orc_render_pinhole.argtypes = [C.POINTER(S.SceneDesc), C.POINTER(S.RenderDesc), vp, vp, C.POINTER(C.c_uint64), C.c_int]
        L.orc_render_pinhole_mt.argtypes = [C.POINTER(S.SceneDesc), C.POINTER(S.RenderDesc), vp, vp, vp, C.POINTER(C.c_uint64)]
        L.orc_set_libm_trig.argtypes = [C.c_int]
        L.orc_render_pinhole_xyz.argtypes = [C.POINTER(S.SceneDesc), C.POINTER(S.RenderDesc), vp, vp, C.c_double, vp, vp, C.POINTER(C.c_uint64), C.c_int]
        L.orc_pinhole_rays.argtypes = [C.POINTER(S.RenderDesc), vp]
        L.orc_max_threads.restype = C.c_int
        _lib = L
    return _lib


def p(a):
    return None if a is None else a.ctypes.data_as(vp)


def _rays(origin, direction, max_distance):
    o = np.ascontiguousarray(origin, dtype=np.float64).reshape(-1, 3)
    d = np.ascontiguousarray(direction, dtype=np.float64).reshape(-1, 3)
    n = o.shape[0]
    m = np.full(n, np.inf) if max_distance is None else np.ascontiguousarray(np.broadcast_to(max_distance, (n,)), dtype=np.float64)
    return o, d, m, n


def _hit_out(n, geometry):
    return dict(prim=np.empty(n, dtype=np.int32), t=np.empty(n), exiting=np.empty(n, dtype=np.uint8),
                tri=np.empty(n, dtype=np.int32), uvw=np.empty((n, 3), dtype=np.float32),
                geom=np.empty((n, 12)) if geometry else None)


def mt_uniform(seed, n):
    st = np.zeros(313, dtype=np.uint64)
    words = np.frombuffer(int(seed).to_bytes(8 * 312, "big"), dtype=">u8").astype(np.uint64)
    lib().orc_mt_seed_words(p(st), p(words), 312)
    out = np.empty(n)
    lib().orc_mt_uniform(p(st), n, p(out))
    return out


def mt_state(seed):
    """A seeded MT19937-64 state (random.pyx:215-243) for render_pinhole_mt."""
    st = np.zeros(313, dtype=np.uint64)
    words = np.frombuffer(int(seed).to_bytes(8 * 312, "big"), dtype=">u8").astype(np.uint64)
    lib().orc_mt_seed_words(p(st), p(words), 312)
    return st


def render_pinhole_mt(flat, desc, state):
    """The reference's SerialEngine: jitter, scattering and roulette draws all come from ``state`` in execution order."""
    mean = np.zeros((desc.n_tasks, desc.bins))
    var = np.zeros((desc.n_tasks, desc.bins))
    rays = C.c_uint64(0)
    lib().orc_render_pinhole_mt(C.byref(flat.desc), C.byref(desc), p(state), p(mean), p(var), C.byref(rays))
    return mean, var, int(rays.value)


def render_pinhole_xyz(flat, desc, xyz, delta_wavelength, state=None, threads=1):
    """XYZPixelProcessor results per task: (mean[n_tasks,3], variance[n_tasks,3], ray_count); state = MT stream or None."""
    mean = np.zeros((desc.n_tasks, 3))
    var = np.zeros((desc.n_tasks, 3))
    rays = C.c_uint64(0)
    xyz = np.ascontiguousarray(xyz, dtype=np.float64)
    lib().orc_render_pinhole_xyz(C.byref(flat.desc), C.byref(desc), p(state) if state is not None else None, p(xyz), float(delta_wavelength),
                                 p(mean), p(var), C.byref(rays), threads)
    return mean, var, int(rays.value)


def set_libm_trig(on):
    """Philox-mode scattering uses libm's sin/cos (like the reference) instead of the portable pair the device restates."""
    lib().orc_set_libm_trig(int(bool(on)))


def philox(seed, pixel, sample):
    out = np.empty(2)
    lib().orc_philox_uniform2(seed, pixel, sample, p(out))
    return out


def aabb_intersect(lower, upper, origin, direction):
    lo, hi = np.ascontiguousarray(lower, dtype=np.float64), np.ascontiguousarray(upper, dtype=np.float64)
    o, d = np.ascontiguousarray(origin, dtype=np.float64), np.ascontiguousarray(direction, dtype=np.float64)
    res = np.empty((len(o), 3))
    lib().orc_aabb_intersect(len(o), p(lo), p(hi), p(o), p(d), p(res))
    return res


def kd_build(boxes, max_depth=0, min_items=1, hit_cost=20.0, empty_bonus=0.2):
    """Returns (nodes, items, lower, upper, max_depth, blob) with blob = KDTree3DCore.save() bytes."""
    b = np.ascontiguousarray(boxes, dtype=np.float64).reshape(-1, 6)
    h = lib().orc_kd_build(p(b), len(b), max_depth, min_items, hit_cost, empty_bonus)
    try:
        v = S.KDTree()
        lib().orc_kd_view(h, C.byref(v))
        nodes, items = S.kd_view_to_arrays(v)
        need = lib().orc_kd_serialise(h, None, 0)
        blob = np.zeros(need, dtype=np.uint8)
        lib().orc_kd_serialise(h, p(blob), need)
        return nodes, items, np.array(list(v.lower)), np.array(list(v.upper)), v.max_depth, blob.tobytes()
    finally:
        lib().orc_kd_free(h)


def mesh_prepare(vertices, triangles):
    """filter + face normals + triangle boxes, oracle side: (n_kept, triangles, face_normals, boxes)"""
    v = np.ascontiguousarray(vertices, dtype=np.float32)
    t = np.ascontiguousarray(triangles, dtype=np.int32).copy()
    stride = t.shape[1]
    n = lib().orc_mesh_filter_triangles(p(v), p(t), len(t), stride)
    t = t[:n]
    fn = np.zeros((n, 3), dtype=np.float32)
    lib().orc_mesh_face_normals(p(v), p(t), n, stride, p(fn))
    boxes = np.zeros((n, 6))
    lib().orc_mesh_triangle_aabbs(p(v), p(t), n, stride, p(boxes))
    return n, t, fn, boxes


def mesh_world_bbox(vertices, matrix):
    v = np.ascontiguousarray(vertices, dtype=np.float32)
    m = np.ascontiguousarray(matrix, dtype=np.float64).reshape(16)
    out = np.zeros(6)
    lib().orc_mesh_world_bbox(p(v), len(v), p(m), p(out))
    return out


def hit_batch(flat, origin, direction, max_distance=None, geometry=False, threads=1, counters=False):
    o, d, m, n = _rays(origin, direction, max_distance)
    out = _hit_out(n, geometry)
    cnt = np.zeros(4, dtype=np.int64)
    lib().orc_hit_batch(C.byref(flat.desc), n, p(o), p(d), p(m), p(out["prim"]), p(out["t"]), p(out["exiting"]), p(out["tri"]),
                        p(out["uvw"]), p(out["geom"]), threads, p(cnt))
    if counters:
        out["counters"] = dict(nodes=int(cnt[0]), items=int(cnt[1]), tris=int(cnt[2]), prims=int(cnt[3]))
    return out


def packet_counters(flat, origin, direction, group=64, threads=1):
    """Distinct records per group of `group` consecutive rays — what a wave walking the trees as ONE packet fetches: array
    [n_groups, 7] = world nodes, world leaves, world leaf items, mesh trees entered, mesh nodes, non-empty mesh leaves, mesh leaf items."""
    o, d, m, n = _rays(origin, direction, None)
    groups = n // group
    out = np.zeros((groups, 7), dtype=np.int64)
    lib().orc_packet_counters(C.byref(flat.desc), groups, group, p(o), p(d), p(m), p(out), threads)
    return out


def prim_hit_batch(flat, index, origin, direction, max_distance=None, geometry=False, threads=1):
    o, d, m, n = _rays(origin, direction, max_distance)
    out = _hit_out(n, geometry)
    lib().orc_prim_hit_batch(C.byref(flat.desc), index, n, p(o), p(d), p(m), p(out["prim"]), p(out["t"]), p(out["exiting"]), p(out["tri"]),
                             p(out["uvw"]), p(out["geom"]), threads)
    return out


def roots_batch(flat, index, origin, direction, max_distance=None, max_roots=8, geometry=False):
    o, d, m, n = _rays(origin, direction, max_distance)
    counts = np.zeros(n, dtype=np.int32)
    t = np.zeros((n, max_roots))
    ex = np.zeros((n, max_roots), dtype=np.uint8)
    g = np.zeros((n, max_roots, 12)) if geometry else None
    lib().orc_roots_batch(C.byref(flat.desc), index, n, p(o), p(d), p(m), max_roots, p(counts), p(t), p(ex), p(g))
    return (counts, t, ex, g) if geometry else (counts, t, ex)


def contains_batch(flat, points):
    q = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    inside = np.zeros((len(q), max(1, flat.n_world)), dtype=np.uint8)
    lib().orc_contains_batch(C.byref(flat.desc), len(q), p(q), p(inside))
    return inside[:, :flat.n_world]


def prim_contains_batch(flat, index, points):
    q = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    inside = np.zeros(len(q), dtype=np.uint8)
    lib().orc_prim_contains_batch(C.byref(flat.desc), index, len(q), p(q), p(inside))
    return inside


def add_samples(x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    states = np.empty((len(x), 3))
    lib().orc_add_samples(len(x), p(x), p(states))
    return states


def frame_combine(ma, va, na, mb, vb, nb):
    ma, va = np.array(ma, dtype=np.float64), np.array(va, dtype=np.float64)
    na = np.array(na, dtype=np.int32)
    mb, vb = np.ascontiguousarray(mb, dtype=np.float64), np.ascontiguousarray(vb, dtype=np.float64)
    nb = np.ascontiguousarray(nb, dtype=np.int32)
    lib().orc_frame_combine(ma.size, p(ma), p(va), p(na), p(mb), p(vb), p(nb))
    return ma, va, na


def render_pinhole(flat, desc, threads=1):
    """desc: source_amd._lib.RenderDesc -> (mean[n_tasks,bins], variance[n_tasks,bins], ray_count)"""
    mean = np.zeros((desc.n_tasks, desc.bins))
    var = np.zeros((desc.n_tasks, desc.bins))
    rays = C.c_uint64(0)
    lib().orc_render_pinhole(C.byref(flat.desc), C.byref(desc), p(mean), p(var), C.byref(rays), threads)
    return mean, var, int(rays.value)


def pinhole_rays(desc):
    out = np.zeros((desc.n_tasks * desc.spp, 7))
    lib().orc_pinhole_rays(C.byref(desc), p(out))
    return out


def _portable(op, a, b=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = None if b is None else np.ascontiguousarray(b, dtype=np.float64)
    o0, o1 = np.zeros(len(a)), np.zeros(len(a))
    f = lib().orc_portable_math
    f.restype, f.argtypes = None, [C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    f(op, len(a), p(a), p(b), p(o0), p(o1))
    return o0, o1


def portable_pow(a, b):
    return _portable(0, a, b)[0]


def portable_sincos(phi):
    return _portable(1, phi)


def portable_asin(x):
    return _portable(2, x)[0]


def max_threads():
    return int(lib().orc_max_threads())
