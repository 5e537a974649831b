"""
Device-side handles: one rsx_ctx per process/GPU and the uploaded scenes. Everything here calls librsx through
ctypes (include/rsx.h); if the library or a gfx950 device is missing these calls raise — there is no CPU path.
"""
import ctypes as C
import os
import threading
import weakref

import numpy as np

from . import _lib
from .core.math import Normal3D, Point3D
from .core.scenegraph import Intersection, MeshIntersection

_contexts = {}
# observers holding accepted, not yet submitted passes (optical/observer.py: _lazy_pass). Strong references on purpose: a camera that goes
# out of scope with passes pending must still deliver them into the pipeline frames its user keeps (it leaves the set when it has).
pending_observers = set()


def settle_observers():
    """Renders the passes observers have accepted but not yet submitted (small passes are batched into one library call)."""
    for obs in list(pending_observers):
        pending_observers.discard(obs)
        obs._flush_lazy()


class DeviceContext:
    """rsx_ctx wrapper: device memory helpers + kernel timing."""

    def __init__(self, ordinal):
        self.ordinal = ordinal
        self._h = C.c_void_p()
        self._pid = os.getpid()
        _lib.check(_lib.lib().rsx_init(int(ordinal), C.byref(self._h)))

    @property
    def handle(self):
        return self._h

    def alloc(self, nbytes):
        p = C.c_void_p()
        _lib.check(_lib.lib().rsx_dev_alloc(self._h, int(nbytes), C.byref(p)))
        return p

    def free(self, p):
        if os.getpid() == self._pid:                        # (never from a forked material worker: the device belongs to the parent)
            _lib.lib().rsx_dev_free(self._h, p)

    def upload(self, p, array):
        a = np.ascontiguousarray(array)
        _lib.check(_lib.lib().rsx_dev_upload(self._h, p, _lib.ptr(a), a.nbytes))

    def download(self, array, p):
        assert array.flags["C_CONTIGUOUS"]
        _lib.check(_lib.lib().rsx_dev_download(self._h, _lib.ptr(array), p, array.nbytes))

    def memset(self, p, value, nbytes):
        _lib.check(_lib.lib().rsx_dev_memset(self._h, p, int(value), int(nbytes)))

    def set_stream(self, hip_stream):
        _lib.check(_lib.lib().rsx_set_stream(self._h, C.c_void_p(hip_stream) if hip_stream else None))

    def synchronize(self):
        settle_observers()                                  # (work an observer is still holding back counts as issued)
        _lib.check(_lib.lib().rsx_synchronize(self._h))

    def idle(self):
        """True when nothing issued on this context is still running (non-blocking; passes observers hold back are not "issued")."""
        if not _lib.has("rsx_idle"):
            return False
        flag = C.c_int32(0)
        _lib.check(_lib.lib().rsx_idle(self._h, C.byref(flag)))
        return bool(flag.value)

    def set_path_stages(self, mode=-1, min_paths=-1):
        """How path-traced passes are scheduled (rsx_set_path_stages): 1 in stages, 0 one persistent kernel, -1 the library's default;
        2 / 3 = 0 / 1 with the kernel forms that carry the mesh walk kept for scenes without a mesh (A/B, tests)."""
        _lib.check(_lib.lib().rsx_set_path_stages(self._h, int(mode), int(min_paths)))

    def defer_path_checks(self, on):
        """Spectral slices of one observe(): path passes return without their end-of-pass round trip until collect_path_checks()."""
        _lib.check(_lib.lib().rsx_defer_path_checks(self._h, 1 if on else 0))

    def collect_path_checks(self, capacity=4096):
        """Waits for the deferred passes; returns (indices of the calls that must be issued again, rays traced by all of them)."""
        failed = (C.c_int32 * capacity)()
        n, rays = C.c_int32(0), C.c_uint64(0)
        _lib.check(_lib.lib().rsx_collect_path_checks(self._h, failed, capacity, C.byref(n), C.byref(rays)))
        return [int(failed[i]) for i in range(n.value)], int(rays.value)

    def last_render_ms(self):
        """(trace_ms, accumulate_ms) of the most recent render call (HIP events on the launch stream)."""
        a, b = C.c_float(0), C.c_float(0)
        _lib.check(_lib.lib().rsx_last_render_ms(self._h, C.byref(a), C.byref(b)))
        return float(a.value), float(b.value)

    def render_history(self, n):
        """Per-call (trace_ms[n], accumulate_ms[n]) of the last n render calls; synchronises once."""
        a, b = np.zeros(n, dtype=np.float32), np.zeros(n, dtype=np.float32)
        _lib.check(_lib.lib().rsx_render_history(self._h, int(n), _lib.ptr(a), _lib.ptr(b)))
        return a, b

    def last_kernel_ms(self):
        ms = C.c_float(0)
        _lib.check(_lib.lib().rsx_last_kernel_ms(self._h, C.byref(ms)))
        return float(ms.value)


def get_context(ordinal=None):
    """One context per GPU per process; default ordinal = LOCAL_RANK (one process per GPU) or 0."""
    if ordinal is None:
        ordinal = int(os.environ.get("RSX_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    if ordinal not in _contexts:
        _contexts[ordinal] = DeviceContext(ordinal)
    return _contexts[ordinal]


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a.reshape(shape) if shape is not None else a


class DeviceScene:
    """An uploaded FlatScene (rsx_scene)."""

    def __init__(self, flat, context=None):
        self.flat = flat
        self.context = context or get_context()
        self._h = C.c_void_p()
        self._pid = os.getpid()
        _lib.check(_lib.lib().rsx_scene_create(self.context.handle, C.byref(flat.desc), C.byref(self._h)))

    @property
    def handle(self):
        return self._h

    def close(self):
        if self._h:
            if os.getpid() == self._pid:                    # (a forked material worker inherits the object, not the device: hybrid.run_block)
                _lib.lib().rsx_scene_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- World.hit ---------------------------------------------------------------------------------
    def hit_batch(self, origin, direction, max_distance=None, geometry=False):
        o, d = _f64(origin).reshape(-1, 3), _f64(direction).reshape(-1, 3)
        n = o.shape[0]
        m = np.full(n, np.inf) if max_distance is None else _f64(np.broadcast_to(max_distance, (n,)))
        out = dict(prim=np.empty(n, dtype=np.int32), t=np.empty(n), exiting=np.empty(n, dtype=np.uint8),
                   tri=np.empty(n, dtype=np.int32), uvw=np.empty((n, 3), dtype=np.float32),
                   geom=np.empty((n, 12)) if geometry else None)
        _lib.check(_lib.lib().rsx_hit_batch(self._h, n, _lib.ptr(o), _lib.ptr(d), _lib.ptr(m), _lib.ptr(out["prim"]), _lib.ptr(out["t"]),
                                            _lib.ptr(out["exiting"]), _lib.ptr(out["tri"]), _lib.ptr(out["uvw"]), _lib.ptr(out["geom"])))
        return out

    def _intersection(self, ray, prim_obj, t, exiting, tri, uvw, g):
        args = (ray, float(t), prim_obj, Point3D(*g[0:3]), Point3D(*g[3:6]), Point3D(*g[6:9]), Normal3D(*g[9:12]), bool(exiting),
                prim_obj.to_local(), prim_obj.to_root())
        if tri >= 0:
            i = MeshIntersection(*args)
            i.triangle, i.u, i.v, i.w = int(tri), float(uvw[0]), float(uvw[1]), float(uvw[2])
            return i
        return Intersection(*args)

    def host_scene(self):
        """The host-side twin for single rays and points (HostScene; CSG solids included: the stream merge runs on the host too)."""
        h = getattr(self, "_host", None)
        if h is None:
            h = self._host = HostScene(self.flat) if _lib.has("rsx_hit_host") else False
        return h or None

    def hit_single(self, ray):
        host = self.host_scene()
        if host is not None:                                # one ray: answered on the host (a device round trip per ray costs ~70 us)
            o, d = ray.origin, ray.direction
            r1 = host.hit_one(o.x, o.y, o.z, d.x, d.y, d.z, ray.max_distance)
            if r1 is None:
                return None
            prim, t, ex, tri, uvw, g = r1
            return self._intersection(ray, self.flat.records[prim]["obj"], t, ex, tri, uvw, g)
        r = self.hit_batch([[ray.origin.x, ray.origin.y, ray.origin.z]], [[ray.direction.x, ray.direction.y, ray.direction.z]],
                           [ray.max_distance], geometry=True)
        if r["prim"][0] < 0:
            return None
        obj = self.flat.records[int(r["prim"][0])]["obj"]
        return self._intersection(ray, obj, r["t"][0], r["exiting"][0], r["tri"][0], r["uvw"][0], r["geom"][0])

    # -- Primitive.hit / next_intersection ------------------------------------------------------
    def roots_batch(self, index, origin, direction, max_distance=None, max_roots=8, geometry=False):
        o, d = _f64(origin).reshape(-1, 3), _f64(direction).reshape(-1, 3)
        n = o.shape[0]
        m = np.full(n, np.inf) if max_distance is None else _f64(np.broadcast_to(max_distance, (n,)))
        counts = np.zeros(n, dtype=np.int32)
        t = np.zeros((n, max_roots))
        ex = np.zeros((n, max_roots), dtype=np.uint8)
        g = np.zeros((n, max_roots, 12)) if geometry else None
        tri = np.full((n, max_roots), -1, dtype=np.int32) if geometry else None
        uvw = np.zeros((n, max_roots, 3), dtype=np.float32) if geometry else None
        _lib.check(_lib.lib().rsx_roots_batch(self._h, int(index), n, _lib.ptr(o), _lib.ptr(d), _lib.ptr(m), int(max_roots),
                                              _lib.ptr(counts), _lib.ptr(t), _lib.ptr(ex), _lib.ptr(g), _lib.ptr(tri), _lib.ptr(uvw)))
        return (counts, t, ex, g, tri, uvw) if geometry else (counts, t, ex)

    def roots_single(self, index, ray, prim_obj, max_roots=64):
        """Primitive.hit(ray) followed by every next_intersection(): the full ordered list of Intersection objects."""
        counts, t, ex, g, tri, uvw = self.roots_batch(index, [[ray.origin.x, ray.origin.y, ray.origin.z]],
                                                      [[ray.direction.x, ray.direction.y, ray.direction.z]], [ray.max_distance],
                                                      max_roots, geometry=True)
        return [self._intersection(ray, prim_obj, t[0, k], ex[0, k], tri[0, k], uvw[0, k], g[0, k]) for k in range(int(counts[0]))]

    # -- World.contains ----------------------------------------------------------------------------
    def contains_batch(self, points):
        p = _f64(points).reshape(-1, 3)
        inside = np.zeros((p.shape[0], max(1, self.flat.n_world)), dtype=np.uint8)
        _lib.check(_lib.lib().rsx_contains_batch(self._h, p.shape[0], _lib.ptr(p), _lib.ptr(inside)))
        return inside[:, :self.flat.n_world]

    def contains_single(self, point):
        host = self.host_scene()
        if host is not None:
            return host.contains_batch([[point.x, point.y, point.z]])[0]
        return self.contains_batch([[point.x, point.y, point.z]])[0]

    def prim_contains(self, index, point):
        return self.contains_single(point)[index]


_private_scenes = weakref.WeakKeyDictionary()


class HostScene:
    """rsx_host_scene wrapper: World.hit / World.contains for single rays and points on the host (include/rsx.h, "one ray, one point").
    Built from the same FlatScene the device scene is; needs no GPU. The call buffers of hit_one are per thread (ctypes releases the
    GIL during the call: two threads probing one world must not share them)."""

    def __init__(self, flat):
        self.flat = flat
        self._h = C.c_void_p()
        self._tls = threading.local()
        _lib.check(_lib.lib().rsx_host_scene_create(C.byref(flat.desc), C.byref(self._h)))

    def close(self):
        if self._h:
            _lib.lib().rsx_host_scene_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def hit_batch(self, origin, direction, max_distance=None, geometry=False):
        o, d = _f64(origin).reshape(-1, 3), _f64(direction).reshape(-1, 3)
        n = o.shape[0]
        m = np.full(n, np.inf) if max_distance is None else _f64(np.broadcast_to(max_distance, (n,)))
        out = dict(prim=np.empty(n, dtype=np.int32), t=np.empty(n), exiting=np.empty(n, dtype=np.uint8),
                   tri=np.empty(n, dtype=np.int32), uvw=np.empty((n, 3), dtype=np.float32),
                   geom=np.empty((n, 12)) if geometry else None)
        _lib.check(_lib.lib().rsx_hit_host(self._h, n, _lib.ptr(o), _lib.ptr(d), _lib.ptr(m), _lib.ptr(out["prim"]), _lib.ptr(out["t"]),
                                           _lib.ptr(out["exiting"]), _lib.ptr(out["tri"]), _lib.ptr(out["uvw"]), _lib.ptr(out["geom"])))
        return out

    def contains_batch(self, points):
        pts = _f64(points).reshape(-1, 3)
        inside = np.zeros((pts.shape[0], max(1, self.flat.n_world)), dtype=np.uint8)
        _lib.check(_lib.lib().rsx_contains_host(self._h, pts.shape[0], _lib.ptr(pts), _lib.ptr(inside)))
        return inside[:, :self.flat.n_world]

    def hit_one(self, ox, oy, oz, dx, dy, dz, max_distance):
        """One ray through two preallocated ctypes buffers (no numpy on the way): (prim, t, exiting, tri, (u, v, w), geom[12]) or None."""
        b = getattr(self._tls, "one", None)
        if b is None:
            b = self._tls.one = ((C.c_double * 7)(), (C.c_double * 19)(), _lib.lib().rsx_hit_host_one)
        i, o, call = b
        i[0], i[1], i[2], i[3], i[4], i[5], i[6] = ox, oy, oz, dx, dy, dz, max_distance
        rc = call(self._h, i, o)
        if rc:
            _lib.check(rc)
        if o[0] < 0:
            return None
        v = o[:]
        return int(v[0]), v[1], int(v[2]), int(v[3]), (v[4], v[5], v[6]), v[7:19]


def _geometry_versions(prim):
    out = [getattr(prim, "_geometry_version", 0)]
    for operand in (getattr(prim, "_primitive_a", None), getattr(prim, "_primitive_b", None)):
        if operand is not None:
            out.append(_geometry_versions(operand))
    return tuple(out)


def scene_for_primitive(prim):
    """Device scene able to answer Primitive.hit()/contains() for `prim`: the owning world's scene when the primitive
    is registered with a World, otherwise a private one-primitive scene (its own scenegraph root)."""
    from .core.scenegraph import World
    from ._flatten import FlatScene
    root = prim.root
    if isinstance(root, World) and prim in root._primitives:
        scene = root.build_accelerator()
        return scene, scene.flat.index_of[id(prim)]
    cached = _private_scenes.get(prim)
    # geometry edits of an unregistered primitive reach no World (Node._change is a no-op): its own version counter, bumped by
    # notify_geometry_change(), is part of the key — operands of a CSG primitive report through the CSG root to the CSG primitive
    key = (tuple(prim.to_root().m), id(root), _geometry_versions(prim))
    if cached is None or cached[0] != key:
        cached = (key, DeviceScene(FlatScene([prim])))
        _private_scenes[prim] = cached
    return cached[1], 0


def combine_scalar(mx, vx, nx, my, vy, ny):
    """Host restatement of _combine_samples (core/math/statsarray.pyx:780-859) for the non-fused engine path."""
    if nx < ny:
        mx, vx, nx, my, vy, ny = my, vy, ny, mx, vx, nx
    if nx > 1 and ny > 1:
        nt = nx + ny
        mt = (nx * mx + ny * my) / nt
        vx = (nx - 1) * vx / nx
        vy = (ny - 1) * vy / ny
        vt = (nx * (mx * mx + vx) + ny * (my * my + vy)) / nt - mt * mt
        vt = nt * vt / (nt - 1)
        return mt, vt, nt
    if nx == 0 and ny == 0:
        return 0.0, 0.0, 0
    if nx == 1:
        if ny == 0:
            return mx, 0.0, 1
        mt = 0.5 * (mx + my)
        temp = mx - mt
        return mt, 2 * temp * temp, 2
    if ny == 1:
        pm, pv, pn = mx, vx, nx
        n = nx + 1
        m = pm + (my - pm) / n
        v = (pv * (pn - 1) + (my - pm) * (my - m)) / (n - 1)
        return m, v, n
    return mx, vx, nx


def combine_arrays(mx, vx, nx, my, vy, ny):
    """combine_scalar over arrays (same operations element by element, so the same bits): returns (mean, variance, samples)."""
    mx, vx, my, vy = (np.array(a, dtype=np.float64) for a in (mx, vx, my, vy))
    nx, ny = np.array(nx, dtype=np.int64), np.array(ny, dtype=np.int64)
    swap = nx < ny
    mx, my = np.where(swap, my, mx), np.where(swap, mx, my)
    vx, vy = np.where(swap, vy, vx), np.where(swap, vx, vy)
    nx, ny = np.where(swap, ny, nx), np.where(swap, nx, ny)
    mt, vt, nt = mx.copy(), vx.copy(), nx.copy()                # default: (mx, vx, nx)
    with np.errstate(divide="ignore", invalid="ignore"):
        a = (nx > 1) & (ny > 1)
        n_sum = (nx + ny).astype(np.float64)
        fx, fy = nx.astype(np.float64), ny.astype(np.float64)
        m_a = (fx * mx + fy * my) / n_sum
        vxa = (fx - 1) * vx / fx
        vya = (fy - 1) * vy / fy
        v_a = (fx * (mx * mx + vxa) + fy * (my * my + vya)) / n_sum - m_a * m_a
        v_a = n_sum * v_a / (n_sum - 1)
        mt, vt, nt = np.where(a, m_a, mt), np.where(a, v_a, vt), np.where(a, nx + ny, nt)
        z = (nx == 0) & (ny == 0)
        mt, vt, nt = np.where(z, 0.0, mt), np.where(z, 0.0, vt), np.where(z, 0, nt)
        c = (nx == 1) & (ny == 0)
        vt = np.where(c, 0.0, vt)
        d = (nx == 1) & (ny == 1)
        m_d = 0.5 * (mx + my)
        t_d = mx - m_d
        mt, vt, nt = np.where(d, m_d, mt), np.where(d, 2 * t_d * t_d, vt), np.where(d, 2, nt)
        e = (nx > 1) & (ny == 1)
        n_e = fx + 1
        m_e = mx + (my - mx) / n_e
        v_e = (vx * (fx - 1) + (my - mx) * (my - m_e)) / (n_e - 1)
        mt, vt, nt = np.where(e, m_e, mt), np.where(e, v_e, vt), np.where(e, nx + 1, nt)
    return mt, vt, nt.astype(np.int32)
