"""
Host-callback render path: observe() for scenes whose materials have no device lowering (user-written Material subclasses).

The reference calls Material.evaluate_surface / evaluate_volume once per hit from inside Ray.trace (raysect/optical/ray.pyx:338-455,
material/material.pxd:36-47), and a material obtains incoming light by calling daughter.trace(world) recursively. That plugin API is
kept as it is; what changes is who traces the rays and how much Python runs per ray. Every ray — primary or daughter — is a NODE of
its path's call tree, kept as one row of a structure of arrays (origin, direction, depth, Philox counters, parent, result spectrum),
and the nodes are processed a wave at a time:

  * all rays that are ready are traced together (one rsx_hit_batch + one rsx_contains_batch; a handful of stragglers on the host walk);
  * the hit nodes are grouped by primitive. A material whose methods are the library's own (Lambert, Dielectric, emitters, absorber,
    null surfaces — checked method by method, so a subclass that overrides one of them is not mistaken for its base) is evaluated
    for the whole group in numpy: surface frames, importance sampling, Fresnel terms, daughters and roulette as array expressions
    that repeat the scalar host forms (material.py) operation for operation, hence bit for bit;
  * a user subclass of a library ContinuousBSDF that overrides only evaluate_shading — the common plug-in — gets the same array
    pre-stage (frames, sampling, pdfs) and ONE Python call per node, evaluate_shading itself;
  * anything else is called per node through the full plugin API (evaluate_surface, evaluate_volume).

A material is never run twice for one daughter: daughter.trace(world) returns a *deferred* spectrum that records the in-place
operations the material applies (mul_array, mul_scalar, div_scalar, add_array, ...); when the daughter's own spectrum is known the
recorded operations are replayed on it — in the same order on the same numbers, so the same bits — for all nodes with the same
operation list at once. Code that reads a deferred spectrum's samples (or traces inside evaluate_volume) falls back to the classic
scheme: the node is abandoned (_Pending) and evaluated again from the start when its daughters have finished; evaluation is
deterministic — each node draws from its own counter-based random stream, rewound at every evaluation.

Completion runs off a stack: a wave pushes its groups after its parents', so popping finishes daughters before parents.

Random numbers follow librsx's Philox convention (include/rsx.h): for the depth-d ray of sample s of pixel p, draw 2d decides
roulette and draw 2d + 1 feeds the scattering; a scene rendered through this path gives the same frame as the device path, bit for
bit — that is how the path is tested (tests/test_hybrid_cpu.py against the oracle without a GPU,
tests/test_gpu_parity.py::test_host_callback_path_* with the device in the loop). Sibling daughters (a material that traces several
rays) get decorrelated streams.

Per-pixel statistics use the same Welford recurrence in sample order and the same combine_samples merge as the device kernels.
"""
import gc
import math
import multiprocessing
import multiprocessing.connection
import os
import time
import traceback

import numpy as np

from ..core import random as rsrandom
from ..core.math import AffineMatrix3D, Normal3D, Point3D, Vector3D
from ..core.scenegraph import Intersection, MeshIntersection
from . import _portable as P
from . import material as M
from . import ray as ray_module
from .spectral import Spectrum

HOST_WALK_BELOW = 24            # waves smaller than this are answered by the host walk (a device round trip costs ~70 us)


MIN_RAYS_PER_WORKER = 2048     # primary rays a worker process must get to be worth its fork
WORKER_SILENCE_S = float(os.environ.get("RSX_WORKER_SILENCE_S", "600"))   # run_block gives up on forked material workers that say nothing for this long


def trace_wave(scene, host, o, d, m):
    """One wave of rays: hits (rsx_hit_batch layout), the rows that hit something, and World.contains flags of their origins."""
    tracer = host if (host is not None and len(o) < HOST_WALK_BELOW) else scene
    hits = tracer.hit_batch(o, d, m, geometry=True)
    rows = np.nonzero(hits["prim"] >= 0)[0]
    inside = tracer.contains_batch(o[rows]).astype(bool) if len(rows) else None
    return hits, rows, inside


class _Pending(BaseException):
    """Raised through a material's code when it needs the samples of a daughter ray that has not been traced yet."""


class _Stream:
    """uniform() source of one node: pairs of Philox numbers; pair 0 = counter (pixel, sample | draw << 48), pair 1 = the same with
    bit 63 of the pixel word set (the device's second scattering pair), further pairs continue above bit 52."""
    __slots__ = ("seed", "pixel", "word", "pos", "pair", "values")

    def __init__(self, seed, pixel, sample, draw):
        self.seed, self.pixel, self.word = seed, pixel, sample | (draw << 48)
        self.pos, self.pair, self.values = 0, -1, (0.0, 0.0)

    def next(self):
        pair, lane = self.pos >> 1, self.pos & 1
        if pair != self.pair:
            pixel = self.pixel if pair == 0 else self.pixel | (1 << 63) | ((pair - 1) << 52)
            self.values, self.pair = P.philox2(self.seed, pixel, self.word), pair
        self.pos += 1
        return self.values[lane]

    def align(self):
        self.pos += self.pos & 1


class _Deferred(Spectrum):
    """The spectrum of a daughter ray that has not been traced yet: records the in-place operations applied to it. Reading the
    samples raises _Pending (the classic re-evaluation scheme takes over for that node)."""

    def __init__(self, template, child):                    # (no Spectrum.__init__: there are no samples yet)
        d = self.__dict__
        d["min_wavelength"], d["max_wavelength"], d["bins"] = template.min_wavelength, template.max_wavelength, template.bins
        d["child"], d["ops"], d["sig"] = child, [], []

    @property
    def samples(self):
        raise _Pending()

    @samples.setter
    def samples(self, value):
        raise _Pending()

    @property
    def delta_wavelength(self):
        return (self.max_wavelength - self.min_wavelength) / self.bins

    def mul_scalar(self, value):
        self.sig.append("ms")
        self.ops.append(float(value))

    def div_scalar(self, value):
        self.sig.append("ds")
        self.ops.append(float(value))

    def _array_op(self, tag, array):
        if isinstance(array, _Deferred):
            raise _Pending()
        a = np.ascontiguousarray(array, dtype=np.float64)
        if a.shape != (self.bins,):
            raise _Pending()
        self.sig.append((tag, a.tobytes()))
        self.ops.append(a)

    def mul_array(self, array):
        self._array_op("ma", array)

    def add_array(self, array):
        self._array_op("aa", array)

    def sub_array(self, array):
        self._array_op("sa", array)

    def mad_scalar(self, scalar, array):
        if isinstance(array, _Deferred):
            raise _Pending()
        a = np.ascontiguousarray(array, dtype=np.float64)
        if a.shape != (self.bins,):
            raise _Pending()
        self.sig.append(("mad", a.tobytes()))
        self.ops.append((float(scalar), a))

    def copy(self):
        c = _Deferred(self, self.child)
        c.__dict__["ops"], c.__dict__["sig"] = list(self.ops), list(self.sig)
        return c

    def new_spectrum(self):
        return Spectrum(self.min_wavelength, self.max_wavelength, self.bins)


def _apply_ops(S, sig, ops):
    """Replays recorded operations on the rows of S [k, bins]; ops[j] = per-node scalars [k] or the shared array."""
    for tag, op in zip(sig, ops):
        if tag == "ms":
            S *= op[:, None]
        elif tag == "ds":                                   # Spectrum.div_scalar: multiply by the reciprocal (spectrum.pyx:459-467)
            with np.errstate(divide="ignore"):
                S *= np.where(op != 0.0, 1.0 / op, np.inf)[:, None]
        else:
            kind = tag[0]
            if kind == "ma":
                S *= op[None, :]
            elif kind == "aa":
                S += op[None, :]
            elif kind == "sa":
                S -= op[None, :]
            else:                                           # mad_scalar: samples += scalar * array
                S += op[0][:, None] * op[1][None, :]
    return S


# -- lazily materialised arguments of the per-node plugin calls (most materials never look at them) -----------------------------------
class _RowLists:
    """The rows of a [k, 16] array as lists, converted for the whole group on first use."""
    __slots__ = ("array", "lists")

    def __init__(self, array):
        self.array, self.lists = array, None

    def row(self, i):
        if self.lists is None:
            self.lists = self.array.tolist()
        return self.lists[i]


class _LazyMatrix(AffineMatrix3D):
    __slots__ = ("_rows", "_i")

    def __getattr__(self, name):
        if name == "m":
            self.m = m = self._rows.row(self._i)
            return m
        raise AttributeError(name)


def _lazy_matrix(rows, i):
    o = _LazyMatrix.__new__(_LazyMatrix)
    o._rows, o._i = rows, i
    return o


def _vec(cls, x, y, z):
    o = cls.__new__(cls)
    o.x, o.y, o.z = x, y, z
    return o


class _LazyHit:
    """Fields of Intersection, filled on first touch from the wave's arrays."""
    __slots__ = ()

    def __getattr__(self, name):
        src = self._src
        if src is None:
            raise AttributeError(name)
        sched, row, node, prim = src
        self._src = None
        g = row["g"]
        self.ray = sched._ray_of(node)
        self.ray_distance = row["t"]
        self.primitive = prim
        self.hit_point, self.inside_point, self.outside_point = _vec(Point3D, *g[0:3]), _vec(Point3D, *g[3:6]), _vec(Point3D, *g[6:9])
        self.normal = _vec(Normal3D, *g[9:12])
        self.exiting = row["ex"]
        self.world_to_primitive, self.primitive_to_world = prim.to_local(), prim.to_root()
        if row["tri"] >= 0:
            self.triangle = row["tri"]
            self.u, self.v, self.w = row["uvw"]
        return object.__getattribute__(self, name)


class _LazyIntersection(_LazyHit, Intersection):
    __slots__ = ("_src",)


class _LazyMeshIntersection(_LazyHit, MeshIntersection):
    __slots__ = ("_src",)


# -- array forms of the host math (core/math.py), operation for operation ------------------------------------------------------------------
def _xf_point(a, p):
    """Point3D.transform (point.pyx:253-284) of the rows of p [k, 3] by the 16-list a."""
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    w = a[12] * x + a[13] * y + a[14] * z + a[15]
    w = 1.0 / w
    return np.stack(((a[0] * x + a[1] * y + a[2] * z + a[3]) * w, (a[4] * x + a[5] * y + a[6] * z + a[7]) * w,
                     (a[8] * x + a[9] * y + a[10] * z + a[11]) * w), axis=1)


def _xf_vector(a, v):
    """Vector3D.transform (vector.pyx:339-369) by the 16-list a."""
    x, y, z = v[:, 0], v[:, 1], v[:, 2]
    return np.stack((a[0] * x + a[1] * y + a[2] * z, a[4] * x + a[5] * y + a[6] * z, a[8] * x + a[9] * y + a[10] * z), axis=1)


def _xf_vector_rows(A, v):
    """Vector3D.transform by per-row matrices A [k, 16]."""
    x, y, z = v[:, 0], v[:, 1], v[:, 2]
    return np.stack((A[:, 0] * x + A[:, 1] * y + A[:, 2] * z, A[:, 4] * x + A[:, 5] * y + A[:, 6] * z,
                     A[:, 8] * x + A[:, 9] * y + A[:, 10] * z), axis=1)


def _normalise(v):
    t = v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1] + v[:, 2] * v[:, 2]
    t = 1.0 / np.sqrt(t)
    return v * t[:, None]


def _orthogonal(v):
    """core/math.py _orthogonal (vector.pyx:440-472) over rows."""
    n = _normalise(v)
    use_y = np.abs(n[:, 0] * 1.0 + n[:, 1] * 0.0 + n[:, 2] * 0.0) > 0.5
    vx, vy = np.where(use_y, 0.0, 1.0), np.where(use_y, 1.0, 0.0)
    m = n[:, 0] * vx + n[:, 1] * vy + n[:, 2] * 0.0
    u = np.stack((vx - m * n[:, 0], vy - m * n[:, 1], 0.0 - m * n[:, 2]), axis=1)
    t = u[:, 0] * u[:, 0] + u[:, 1] * u[:, 1] + u[:, 2] * u[:, 2]
    t = 1.0 / np.sqrt(t)
    return u * t[:, None]


def _cross(a, b):                                           # vector.pyx:306-310
    return np.stack((a[:, 1] * b[:, 2] - b[:, 1] * a[:, 2], a[:, 2] * b[:, 0] - b[:, 2] * a[:, 0], a[:, 0] * b[:, 1] - b[:, 0] * a[:, 1]), axis=1)


def _matmul_rows_const(A, b):
    """AffineMatrix3D.mul (affinematrix.pyx:255-273) of per-row matrices A [k, 16] with the constant 16-list b."""
    out = np.empty_like(A)
    for i in range(4):
        r = i * 4
        for j in range(4):
            out[:, r + j] = A[:, r] * b[j] + A[:, r + 1] * b[4 + j] + A[:, r + 2] * b[8 + j] + A[:, r + 3] * b[12 + j]
    return out


def _matmul_const_rows(a, B):
    out = np.empty_like(B)
    for i in range(4):
        r = i * 4
        for j in range(4):
            out[:, r + j] = a[r] * B[:, j] + a[r + 1] * B[:, 4 + j] + a[r + 2] * B[:, 8 + j] + a[r + 3] * B[:, 12 + j]
    return out


def _surface_frames(ex, g, w2p, p2w):
    """material._surface_frame (material.pyx:222-246, 304-325, 393-422) over a group of hits of one primitive: launch points of the
    reflected and the transmitted ray (world) and the two surface-space transforms [k, 16]."""
    inside_w, outside_w = _xf_point(p2w, g[:, 3:6]), _xf_point(p2w, g[:, 6:9])
    e = ex[:, None]
    w_refl, w_trans = np.where(e, inside_w, outside_w), np.where(e, outside_w, inside_w)
    normal = np.where(e, -g[:, 9:12], g[:, 9:12])
    tangent = _orthogonal(normal)
    bitangent = _cross(normal, tangent)
    k = len(ex)
    zero, one = np.zeros(k), np.ones(k)
    p2s = np.stack((tangent[:, 0], tangent[:, 1], tangent[:, 2], zero, bitangent[:, 0], bitangent[:, 1], bitangent[:, 2], zero,
                    normal[:, 0], normal[:, 1], normal[:, 2], zero, zero, zero, zero, one), axis=1)
    s2p = np.stack((tangent[:, 0], bitangent[:, 0], normal[:, 0], zero, tangent[:, 1], bitangent[:, 1], normal[:, 1], zero,
                    tangent[:, 2], bitangent[:, 2], normal[:, 2], zero, zero, zero, zero, one), axis=1)
    return w_refl, w_trans, _matmul_rows_const(p2s, w2p), _matmul_const_rows(p2w, s2p)


class _Store:
    """The nodes: one row per ray of every path tree."""
    FIELDS = (("o", 3, np.float64), ("d", 3, np.float64), ("maxd", 0, np.float64), ("depth", 0, np.int64), ("parent", 0, np.int64),
              ("pix", 0, np.uint64), ("smp", 0, np.uint64), ("mix", 0, np.uint64), ("norm", 0, np.float64), ("done", 0, np.bool_))

    def __init__(self, bins, capacity):
        self.n, self.cap, self.bins = 0, 0, bins
        for name, _, _ in self.FIELDS:
            setattr(self, name, None)
        self.res = None
        self._grow(max(1024, capacity))

    def _grow(self, cap):
        for name, width, dtype in self.FIELDS:
            new = np.zeros((cap, width) if width else (cap,), dtype=dtype)
            old = getattr(self, name)
            if old is not None:
                new[:self.n] = old[:self.n]
            setattr(self, name, new)
        new = np.zeros((cap, self.bins))
        if self.res is not None:
            new[:self.n] = self.res[:self.n]
        self.res, self.cap = new, cap

    def alloc(self, k):
        if self.n + k > self.cap:
            self._grow(max(2 * self.cap, self.n + k))
        idx = np.arange(self.n, self.n + k, dtype=np.int64)
        self.n += k
        return idx


class _Group:
    """Nodes that wait for one daughter each: result = volumes(replay(ops, daughter's spectrum)) * norm."""
    __slots__ = ("nodes", "child", "sig", "ops", "vols", "start")

    def __init__(self, nodes, child, sig, ops, vols, start):
        self.nodes, self.child, self.sig, self.ops, self.vols, self.start = nodes, child, sig, ops, vols, start


def _is(material, name, owner):
    return getattr(type(material), name, None) is getattr(owner, name)


def surface_kind(material):
    """Which array form evaluates `material`'s surface: decided method by method, so a subclass that overrides a hook gets the
    per-node path for exactly that hook."""
    if _is(material, "evaluate_surface", M.ContinuousBSDF):
        if _is(material, "sample", M.Lambert) and _is(material, "pdf", M.Lambert):
            return "lambert" if _is(material, "evaluate_shading", M.Lambert) else "shade"
        return None
    for kind, owner in (("absorber", M.AbsorbingSurface), ("emitter", M.UniformSurfaceEmitter), ("light", M.Light), ("null", M.NullSurface),
                        ("dielectric", M.Dielectric)):
        if _is(material, "evaluate_surface", owner):
            return kind
    return None


def volume_kind(material):
    for owner in (M.NullVolume, M.NullMaterial, M.ContinuousBSDF):
        if _is(material, "evaluate_volume", owner):
            return "none"
    if _is(material, "evaluate_volume", M.UniformVolumeEmitter):
        return "emit"
    if _is(material, "evaluate_volume", M.Dielectric):
        return "pow"
    return "user"


def python_materials(world, per_node=False):
    """True when some material of the world is evaluated by Python code per node (worker processes pay off only then)."""
    return per_node or any(surface_kind(p.material) in (None, "shade") or volume_kind(p.material) == "user" for p in world._primitives)


class WaveScheduler:
    """Traces the paths of one block of pixels of one spectral slice. seed / sample counters as in rsx_render_desc."""

    def __init__(self, world, scene, seed, template, per_node=False):
        self.world, self.scene, self.flat, self.seed, self.template = world, scene, scene.flat, int(seed), template
        self.bins = template.bins
        self.lo, self.hi = template.min_wavelength, template.max_wavelength
        self.p_ext, self.min_depth, self.max_depth = template.extinction_prob, template.extinction_min_depth, template.max_depth
        self.importance, self.weight = bool(template.importance_sampling), template.important_path_weight
        self.cfg = self._config(template)
        self.rays = 0
        self.st = None
        self.need_hit, self.buf, self.stack = [], [], []
        self.user_ray, self.children, self.hitrow, self.inside_of = {}, {}, {}, {}
        self.current, self.ordinal, self.defer, self.cur_children = -1, 0, False, None
        self.cur_mix = self.cur_depth = self.cur_pix = self.cur_smp = 0
        self.stream = None
        prims = world._primitives
        self.prims = prims
        self.skind = [None if per_node else surface_kind(p.material) for p in prims]
        self.vkind = ["user" if per_node else volume_kind(p.material) for p in prims]
        self.user_volume = np.array([k == "user" for k in self.vkind], dtype=bool)
        self.w2p = [p.to_local().m for p in prims]
        self.p2w = [p.to_root().m for p in prims]
        self.spheres = world._spheres_cached() if hasattr(world, "_spheres_cached") else []
        self.mis = self.importance and len(self.spheres) > 0
        host = scene.host_scene() if hasattr(scene, "host_scene") else None
        self.trace_wave = scene.trace_wave if hasattr(scene, "trace_wave") else (lambda o, d, m: trace_wave(scene, host, o, d, m))
        self._primary = template.copy()

    @staticmethod
    def _config(ray):
        return (ray.min_wavelength, ray.max_wavelength, ray.bins, ray.extinction_prob, ray.extinction_min_depth, ray.max_depth,
                bool(ray.importance_sampling), ray.important_path_weight)

    # -- node bookkeeping ---------------------------------------------------------------------------------------------------------
    def _ray_of(self, node, o=None, d=None):
        """The optical Ray object of a node (built on demand for nodes the array forms spawned)."""
        r = self.user_ray.get(node)
        if r is None:
            st, t = self.st, self.template
            r = ray_module.Ray.__new__(ray_module.Ray)
            if o is None:
                o, d = st.o[node].tolist(), st.d[node].tolist()
            r.origin, r.direction, r.max_distance = _vec(Point3D, *o), _vec(Vector3D, *d), float(st.maxd[node])
            r.min_wavelength, r.max_wavelength, r.bins = t.min_wavelength, t.max_wavelength, t.bins
            r.extinction_prob, r.extinction_min_depth, r.max_depth = t.extinction_prob, t.extinction_min_depth, t.max_depth
            r.importance_sampling, r.important_path_weight = t.importance_sampling, t.important_path_weight
            r.depth, r.ray_count, r._node, r._primary_ray = int(st.depth[node]), 0, None, self._primary
            self.user_ray[node] = r
        return r

    def _pixel_words(self, idx):
        st = self.st
        return st.pix[idx] | (st.mix[idx] << np.uint64(40))

    def _uniform_pair(self, idx, draw_offset, pair=0):
        """Philox pair `pair` of draw 2 * depth + draw_offset of the nodes idx."""
        st = self.st
        pixel = self._pixel_words(idx)
        if pair:
            pixel = pixel | np.uint64(1 << 63) | np.uint64((pair - 1) << 52)
        draw = (2 * st.depth[idx] + draw_offset).astype(np.uint64)
        return P.philox2_array(self.seed, pixel, st.smp[idx] | (draw << np.uint64(48)))

    def _roulette(self, idx):
        """Russian roulette of freshly spawned nodes (ray.pyx:380-388); survivors are queued for the next wave."""
        st = self.st
        depth = st.depth[idx]
        gamble = depth >= self.min_depth
        if gamble.any():
            g = idx[gamble]
            u = self._uniform_pair(g, 0)[0]
            dead = (st.depth[g] >= self.max_depth) | (u < self.p_ext)
            st.done[g[dead]] = True                         # (the result rows are zero already)
            st.norm[g[~dead]] = 1 / (1 - self.p_ext)
            idx = np.concatenate((idx[~gamble], g[~dead]))
        if len(idx):
            self.need_hit.append(idx)

    def _spawn(self, parents, o, d, depth, keep_alive=False):
        """Daughters of `parents` spawned by an array form: one each; stream identity inherited (the device's numbering)."""
        self._flush_users()
        st = self.st
        idx = st.alloc(len(parents))
        st.o[idx], st.d[idx], st.maxd[idx], st.depth[idx], st.parent[idx] = o, d, st.maxd[parents], depth, parents
        st.pix[idx], st.smp[idx], st.mix[idx], st.norm[idx] = st.pix[parents], st.smp[parents], st.mix[parents], 1.0
        self.rays += len(idx)
        if keep_alive:
            self.need_hit.append(idx)
        else:
            self._roulette(idx)
        return idx

    # -- called by Ray.trace of a daughter, from inside a material ----------------------------------------------------------------
    def trace(self, ray, world, keep_alive):
        cur = self.current
        if cur < 0:
            raise RuntimeError("Ray.trace() inside a host-callback render must be called from a material's evaluate_surface / evaluate_volume")
        ordinal = self.ordinal
        self.ordinal = ordinal + 1
        if self.defer:                                      # first evaluation of the node: every daughter is new
            child = self._spawn_deferred(cur, ordinal, ray, keep_alive)
            self.cur_children.append((ordinal, child))
            return _Deferred(ray, child)
        known = self.children.get(cur)
        child = known.get(ordinal) if known else None
        if child is None:
            child = self._spawn_user(cur, ordinal, ray, keep_alive)
            self.cur_children.append((ordinal, child))
        st = self.st
        if st.done[child]:
            s = Spectrum.__new__(Spectrum)
            s.__dict__.update(min_wavelength=ray.min_wavelength, max_wavelength=ray.max_wavelength, bins=ray.bins,
                              delta_wavelength=(ray.max_wavelength - ray.min_wavelength) / ray.bins, samples=st.res[child].copy(),
                              _sample_key=None, _sample_cache=None)
            return s                                        # (a copy: the caller scales it in place)
        raise _Pending()

    def _spawn_deferred(self, parent, ordinal, ray, keep_alive):
        """A daughter spawned by user code in deferral mode: its row is reserved now and written with the rest of the wave's
        (_flush_users); roulette is decided there for all of them at once."""
        depth = ray.depth
        mix = self.cur_mix
        if ordinal > 0 or (depth == self.cur_depth and self.stream.pos > 0):
            mix = (mix * 0x9E3779B1 + ordinal + 1 + 0x7F4A7C15 * (self.cur_depth + 1)) & 0xFFFFF or 1
        buf = self.buf
        node = self.st.n + len(buf)
        o, d = ray.origin, ray.direction
        if keep_alive or depth < ray.extinction_min_depth:
            mode = 0
        elif self._config(ray) == self.cfg:
            mode = 1
        else:                                               # a ray with its own roulette settings: decided here
            u = P.philox2(self.seed, self.cur_pix | (mix << 40), self.cur_smp | ((2 * depth) << 48))[0]
            mode = 2 if (depth >= ray.max_depth or u < ray.extinction_prob) else 3
        buf.append((o.x, o.y, o.z, d.x, d.y, d.z, ray.max_distance, depth, parent, mix, mode,
                    1 / (1 - ray.extinction_prob) if mode == 3 else 1.0))
        self.user_ray[node] = ray
        return node

    def _flush_users(self):
        buf = self.buf
        if not buf:
            return
        self.buf = []
        a = np.array(buf, dtype=np.float64)
        st = self.st
        idx = st.alloc(len(buf))
        parents = a[:, 8].astype(np.int64)
        st.o[idx], st.d[idx], st.maxd[idx], st.depth[idx], st.parent[idx] = a[:, 0:3], a[:, 3:6], a[:, 6], a[:, 7].astype(np.int64), parents
        st.pix[idx], st.smp[idx], st.mix[idx], st.norm[idx] = st.pix[parents], st.smp[parents], a[:, 9].astype(np.uint64), a[:, 11]
        self.rays += len(idx)
        mode = a[:, 10]
        st.done[idx[mode == 2]] = True
        go = idx[(mode == 0) | (mode == 3)]
        if len(go):
            self.need_hit.append(go)
        gamble = idx[mode == 1]
        if len(gamble):
            self._roulette(gamble)

    def _spawn_user(self, parent, ordinal, ray, keep_alive):
        # stream identity: the first daughter of a chain keeps mix = parent's (the device's numbering); siblings, and same-depth
        # daughters of a node that consumed random numbers itself, move to decorrelated counters
        self._flush_users()
        st = self.st
        mix = int(st.mix[parent])
        pdepth = int(st.depth[parent])
        if ordinal > 0 or (ray.depth == pdepth and self.stream is not None and self.stream.pos > 0):
            mix = (mix * 0x9E3779B1 + ordinal + 1 + 0x7F4A7C15 * (pdepth + 1)) & 0xFFFFF or 1
        idx = st.alloc(1)
        st = self.st
        node = int(idx[0])
        o, d = ray.origin, ray.direction
        st.o[node] = (o.x, o.y, o.z)
        st.d[node] = (d.x, d.y, d.z)
        st.maxd[node], st.depth[node], st.parent[node] = ray.max_distance, ray.depth, parent
        st.pix[node], st.smp[node], st.mix[node], st.norm[node] = st.pix[parent], st.smp[parent], mix, 1.0
        self.user_ray[node] = ray
        self.rays += 1
        if keep_alive or ray.depth < ray.extinction_min_depth:              # ray.pyx:380-388
            self.need_hit.append(idx)
        else:
            u = P.philox2(self.seed, int(st.pix[node]) | (mix << 40), int(st.smp[node]) | ((2 * ray.depth) << 48))[0]
            if ray.depth >= ray.max_depth or u < ray.extinction_prob:
                st.done[node] = True
            else:
                st.norm[node] = 1 / (1 - ray.extinction_prob)
                self.need_hit.append(idx)
        return node

    # -- volumes -------------------------------------------------------------------------------------------------------------------
    def _volume_plan(self, nodes, inside):
        """[(rows, volume primitive)] in application order for the rows of `nodes`; inside = contains flags [k, n_world]."""
        if inside is None or not inside.any():
            return None
        count = inside.sum(axis=1)
        plan = []
        one = np.nonzero(count == 1)[0]
        if len(one):
            which = inside[one].argmax(axis=1)
            for vp in np.unique(which):
                plan.append((one[which == vp], int(vp)))
        for r in np.nonzero(count > 1)[0]:                  # several volumes: the world tree's leaf order (world.pyx:149-168)
            o = self.st.o[nodes[r]]
            for k in self.flat.contains_order(_vec(Point3D, float(o[0]), float(o[1]), float(o[2]))):
                if inside[r, k]:
                    plan.append((np.array([r]), int(k)))
        return plan

    def _apply_volumes(self, S, G):
        """Ray._sample_volumes (ray.pyx:422-455) on the rows of S: start = the hit point, end = the ray's origin."""
        st = self.st
        for rows, vp in G.vols:
            kind = self.vkind[vp]
            if kind == "none":
                continue
            mat = self.prims[vp].material
            start, end = G.start[rows], st.o[G.nodes[rows]]
            if kind == "emit":                              # UniformVolumeEmitter.evaluate_volume
                a = self.w2p[vp]
                s_l, e_l = _xf_point(a, start), _xf_point(a, end)
                v = s_l - e_l
                length = np.sqrt(v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1] + v[:, 2] * v[:, 2])
                emission = mat.emission_spectrum.sample(self.lo, self.hi, self.bins) * mat.scale
                S[rows] = np.where((length == 0)[:, None], S[rows], S[rows] + emission[None, :] * length[:, None])
            else:                                           # Dielectric.evaluate_volume, the device's portable pow
                v = end - start
                length = np.sqrt(v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1] + v[:, 2] * v[:, 2])
                table = mat.transmission.sample(self.lo, self.hi, self.bins)
                cols = np.nonzero(table != 1.0)[0]
                if len(cols):
                    S[np.ix_(rows, cols)] = S[np.ix_(rows, cols)] * P.pow_array(table[cols][None, :], length[:, None])
        return S

    def _finish(self, nodes, S, vols, start):
        """Surface spectra S of `nodes` are final: volumes, roulette normalisation, done."""
        st = self.st
        if vols:
            S = self._apply_volumes(S, _Group(nodes, None, None, None, vols, start))
        S *= st.norm[nodes][:, None]
        st.res[nodes] = S
        st.done[nodes] = True

    def _complete(self, G):
        st = self.st
        S = _apply_ops(st.res[G.child], G.sig, G.ops)       # (fancy indexing copies)
        self._finish(G.nodes, S, G.vols, G.start)

    # -- array forms of the library's materials -----------------------------------------------------------------------------------
    def _push(self, nodes, child, sig, ops, vols, start):
        if len(nodes):
            self.stack.append(_Group(nodes, child, sig, ops, vols, start))

    def _do_final(self, nodes, S, vols, start):
        self._finish(nodes, S, vols, start)

    def _do_null(self, nodes, p, g, ex, vols, start):       # NullSurface.evaluate_surface (material.pyx:104-147)
        st = self.st
        origin = _xf_point(self.p2w[p], np.where(ex[:, None], g[:, 6:9], g[:, 3:6]))
        child = self._spawn(nodes, origin, st.d[nodes], st.depth[nodes], keep_alive=True)
        self._push(nodes, child, [], [], vols, start)

    def _continuous_pre(self, nodes, p, g, ex):
        """ContinuousBSDF.evaluate_surface up to the call of evaluate_shading, with Lambert's sample() / pdf()."""
        st = self.st
        w2p, p2w = self.w2p[p], self.p2w[p]
        w_refl, w_trans, w2s, s2w = _surface_frames(ex, g, w2p, p2w)
        s_in = -_xf_vector_rows(w2s, st.d[nodes])
        k = len(nodes)
        if self.mis:
            w_hit = _xf_point(p2w, g[:, 0:3])
            u0, u1 = self._uniform_pair(nodes, 1)
            ua, ub = self._uniform_pair(nodes, 1, pair=1)
            important = u0 < self.weight
            w_imp = self._important_sample(w_hit, u1, ua, ub)
            s_cos = self._cosine_sample(ua, ub)
            s_out = np.where(important[:, None], _xf_vector_rows(w2s, w_imp), s_cos)
            w_out = np.where(important[:, None], w_imp, _xf_vector_rows(s2w, s_cos))
            pdf_important = self._important_pdf(w_hit, w_out)
            pdf_bsdf = np.where(s_out[:, 2] >= 0.0, (1.0 / math.pi) * s_out[:, 2], 0.0)
            pdf = self.weight * pdf_important + (1 - self.weight) * pdf_bsdf
            pos = 4
        else:
            u0, u1 = self._uniform_pair(nodes, 1)
            s_out = self._cosine_sample(u0, u1)
            pdf = np.where(s_out[:, 2] >= 0.0, (1.0 / math.pi) * s_out[:, 2], 0.0)
            pos = 2
        return w_refl, w_trans, w2s, s2w, s_in, s_out, pdf, pos

    @staticmethod
    def _cosine_sample(ua, ub):                             # material.hemisphere_cosine_sample (solidangle.pyx:228-233)
        r = np.sqrt(ua)
        sn, cs = P.sincos_array(2.0 * math.pi * ub)
        x, y = r * cs, r * sn
        z2 = 1.0 - x * x - y * y
        return np.stack((x, y, np.sqrt(np.where(z2 > 0, z2, 0.0))), axis=1)

    def _important_sample(self, origin, pick, ua, ub):
        """World.important_direction_sample (world.pyx:150-188) for every row (the rows that do not use it discard theirs)."""
        spheres = self.spheres
        index = np.zeros(len(pick), dtype=np.int64)
        for i in range(len(spheres) - 1):
            index += (index == i) & ~(pick < spheres[i][2])
        centre = np.array([s[0] for s in spheres], dtype=np.float64)[index]
        radius = np.array([s[1] for s in spheres], dtype=np.float64)[index]
        dv = centre - origin
        with np.errstate(all="ignore"):
            distance = np.sqrt(dv[:, 0] * dv[:, 0] + dv[:, 1] * dv[:, 1] + dv[:, 2] * dv[:, 2])
            inside = (distance == 0) | (distance < radius)
            z = 1.0 - 2.0 * ua                              # vector_sphere (random.pyx:373-387)
            r2 = 1.0 - z * z
            r = np.sqrt(np.where(r2 > 0, r2, 0.0))
            sn, cs = P.sincos_array(2.0 * math.pi * ub)
            sphere = np.stack((r * cs, r * sn, z), axis=1)
            angular_radius = P.asin_array(np.where(inside, 0.0, radius / distance))      # vector_cone_uniform (random.pyx:425-446)
            theta = angular_radius * 180 / math.pi
            theta = theta * 0.017453292519943295
            cos_theta = P.sincos_array(theta)[1]
            z = ub * (1 - cos_theta) + cos_theta
            r2 = 1.0 - z * z
            r = np.sqrt(np.where(r2 > 0, r2, 0.0))
            sn, cs = P.sincos_array(2.0 * math.pi * ua)
            sx, sy, sz = r * cs, r * sn, z
            safe = np.where(inside[:, None], 1.0, dv)
            d = _normalise(safe)
            up = _orthogonal(d)
            right = _cross(up, d)
            cone = np.stack((right[:, 0] * sx + up[:, 0] * sy + d[:, 0] * sz, right[:, 1] * sx + up[:, 1] * sy + d[:, 1] * sz,
                             right[:, 2] * sx + up[:, 2] * sy + d[:, 2] * sz), axis=1)
        return np.where(inside[:, None], sphere, cone)

    def _important_pdf(self, origin, direction):            # world.pyx:190-230
        pdf_all = np.zeros(len(origin))
        with np.errstate(all="ignore"):
            for centre, radius, _, weight in self.spheres:
                ax, ay, az = centre[0] - origin[:, 0], centre[1] - origin[:, 1], centre[2] - origin[:, 2]
                distance = np.sqrt(ax * ax + ay * ay + az * az)
                inside = (distance == 0) | (distance < radius)
                t = radius / distance
                angular_radius_cos = np.sqrt(1 - t * t)
                k = ax * ax + ay * ay + az * az
                k = 1.0 / np.sqrt(k)
                bx, by, bz = ax * k, ay * k, az * k
                outside_cone = direction[:, 0] * bx + direction[:, 1] * by + direction[:, 2] * bz < angular_radius_cos
                solid_angle = np.where(inside, 4 * math.pi, 2 * math.pi * (1 - angular_radius_cos))
                term = weight * (1 / solid_angle)
                pdf_all = np.where(~inside & outside_cone, pdf_all, pdf_all + term)
        return pdf_all

    def _do_lambert(self, nodes, p, g, ex, vols, start):    # Lambert under ContinuousBSDF.evaluate_surface (lambert.pyx:71-104)
        st = self.st
        mat = self.prims[p].material
        w_refl, _, _, s2w, _, s_out, pdf_mix, _ = self._continuous_pre(nodes, p, g, ex)
        pdf = np.where(s_out[:, 2] >= 0.0, (1.0 / math.pi) * s_out[:, 2], 0.0)
        dark = pdf == 0.0
        if dark.any():                                      # zero spectrum, divided by the mixture pdf
            z = np.nonzero(dark)[0]
            S = _apply_ops(np.zeros((len(z), self.bins)), ["ds"], [pdf_mix[z]])
            self._finish(nodes[z], S, self._sub_plan(vols, z), None if start is None else start[z])
        lit = np.nonzero(~dark)[0]
        if len(lit):
            sub = nodes[lit]
            child = self._spawn(sub, w_refl[lit], _xf_vector_rows(s2w[lit], s_out[lit]), st.depth[sub] + 1)
            table = mat.reflectivity.sample(self.lo, self.hi, self.bins)
            self._push(sub, child, [("ma", None), "ms", "ds"], [np.asarray(table, dtype=np.float64), pdf[lit], pdf_mix[lit]],
                       self._sub_plan(vols, lit), None if start is None else start[lit])

    @staticmethod
    def _sub_plan(vols, rows):
        """The volume plan restricted to `rows` (re-indexed)."""
        if not vols:
            return None
        pos = np.full(int(max(r.max() for r, _ in vols)) + 1 if vols else 0, -1, dtype=np.int64)
        keep = rows[rows < len(pos)]
        pos[keep] = np.nonzero(rows < len(pos))[0]
        out = []
        for r, vp in vols:
            m = pos[r]
            m = m[m >= 0]
            if len(m):
                out.append((m, vp))
        return out or None

    def _do_dielectric(self, nodes, p, g, ex, vols, start):  # Dielectric.evaluate_surface (dielectric.pyx:159-328)
        st = self.st
        mat = self.prims[p].material
        w2p, p2w = self.w2p[p], self.p2w[p]
        incident = _normalise(_xf_vector(w2p, st.d[nodes]))
        normal = _normalise(g[:, 9:12])
        c1 = -(normal[:, 0] * incident[:, 0] + normal[:, 1] * incident[:, 1] + normal[:, 2] * incident[:, 2])
        internal = mat.index.average(self.lo, self.hi)
        external = mat.external_index.average(self.lo, self.hi)
        back = c1 < 0.0
        n1, n2 = np.where(back, internal, external), np.where(back, external, internal)
        with np.errstate(all="ignore"):
            gamma = n1 / n2
            c2s = 1 - (gamma * gamma) * (1 - c1 * c1)
            tir = c2s <= 0
            root = np.sqrt(np.where(tir, 0.0, c2s))
            temp = np.where(back, gamma * c1 + root, gamma * c1 - root)
            transmitted = gamma[:, None] * incident + temp[:, None] * normal
            ci, ct = c1, -(normal[:, 0] * transmitted[:, 0] + normal[:, 1] * transmitted[:, 1] + normal[:, 2] * transmitted[:, 2])
            ra, rb = (n1 * ci - n2 * ct) / (n1 * ci + n2 * ct), (n1 * ct - n2 * ci) / (n1 * ct + n2 * ci)
            reflectivity = 0.5 * (ra * ra + rb * rb)
            transmission = 1 - reflectivity
        if mat.transmission_only:
            through = ~tir
            dark = tir
        else:
            through = ~tir & (self._uniform_pair(nodes, 1)[0] < transmission)
            dark = np.zeros(len(nodes), dtype=bool)
        temp2 = 2 * c1
        reflected = _xf_vector(p2w, incident + temp2[:, None] * normal)
        inside_w, outside_w = _xf_point(p2w, g[:, 3:6]), _xf_point(p2w, g[:, 6:9])
        b = back[:, None]
        origin = np.where(through[:, None], np.where(b, outside_w, inside_w), np.where(b, inside_w, outside_w))
        direction = np.where(through[:, None], _xf_vector(p2w, transmitted), reflected)
        if dark.any():
            z = np.nonzero(dark)[0]
            self._finish(nodes[z], np.zeros((len(z), self.bins)), self._sub_plan(vols, z), None if start is None else start[z])
        live = np.nonzero(~dark)[0]
        if len(live):
            sub = nodes[live]
            child = self._spawn(sub, origin[live], direction[live], st.depth[sub] + 1)
            self._push(sub, child, [], [], self._sub_plan(vols, live), None if start is None else start[live])

    def _do_light(self, nodes, p, g, vols, start):          # debug Light (debug.pyx:41-79)
        mat = self.prims[p].material
        S = np.zeros((len(nodes), self.bins))
        if mat.intensity != 0.0:
            L = mat.light_direction.transform(self.prims[p].to_local())
            dot = -(L.x * g[:, 9] + L.y * g[:, 10] + L.z * g[:, 11])
            diffuse = mat.intensity * np.where(dot > 0, dot, 0.0)
            S = diffuse[:, None] * mat.spectrum.sample(self.lo, self.hi, self.bins)[None, :]
        self._finish(nodes, S, vols, start)

    def _do_shade(self, nodes, p, g, ex, hits, sel, inside, vols, start):
        """A user subclass of a library ContinuousBSDF that overrides evaluate_shading: frames, sampling and pdfs for the group in
        arrays, one call of the user's method per node, daughters deferred."""
        st = self.st
        prim = self.prims[p]
        mat = prim.material
        w_refl, w_trans, w2s, s2w, s_in, s_out, pdf_mix, pos = self._continuous_pre(nodes, p, g, ex)
        L = [a.tolist() for a in (w_refl, w_trans, s_in, s_out)]
        pdfs, exl, node_list = pdf_mix.tolist(), ex.tolist(), nodes.tolist()
        rows = self._hit_rows(hits, sel, p)
        world, shade = self.world, mat.evaluate_shading
        stream = self.stream = _Stream(self.seed, 0, 0, 0)
        pixels, words = self._pixel_words(nodes).tolist(), (st.smp[nodes] | ((2 * st.depth[nodes] + 1).astype(np.uint64) << np.uint64(48))).tolist()
        mixes, depths, pix, smp = st.mix[nodes].tolist(), st.depth[nodes].tolist(), st.pix[nodes].tolist(), st.smp[nodes].tolist()
        o_rows, d_rows = st.o[nodes].tolist(), st.d[nodes].tolist()
        w2s_rows, s2w_rows = _RowLists(w2s), _RowLists(s2w)
        previous = rsrandom.set_stream(stream)
        self.defer = True
        groups, finals, final_rows = {}, [], []
        try:
            for i, node in enumerate(node_list):
                stream.pixel, stream.word, stream.pos, stream.pair = pixels[i], words[i], pos, -1
                self.current, self.ordinal, self.cur_children = node, 0, []
                self.cur_mix, self.cur_depth, self.cur_pix, self.cur_smp = mixes[i], depths[i], pix[i], smp[i]
                ray = self._ray_of(node, o_rows[i], d_rows[i])
                hit = self._lazy_hit(rows[i], node, prim)
                try:
                    s = shade(world, ray, _vec(Vector3D, *L[2][i]), _vec(Vector3D, *L[3][i]), _vec(Point3D, *L[0][i]), _vec(Point3D, *L[1][i]),
                              exl[i], _lazy_matrix(w2s_rows, i), _lazy_matrix(s2w_rows, i), hit)
                    s.div_scalar(pdfs[i])
                except _Pending:
                    self.inside_of[node] = self._inside_list(node, inside[i] if inside is not None else None)
                    self._to_classic(node, rows[i])
                    continue
                if type(s) is _Deferred:
                    key = (tuple(s.sig), )
                    slot = groups.get(key)
                    if slot is None:
                        slot = groups[key] = ([], [], [])
                    slot[0].append(i)
                    slot[1].append(s.child)
                    slot[2].append(s.ops)
                else:
                    finals.append(s.samples)
                    final_rows.append(i)
        finally:
            rsrandom.set_stream(previous)
            self.current, self.defer, self.stream = -1, False, None
        self._flush_users()
        if final_rows:
            z = np.array(final_rows)
            self._finish(nodes[z], np.array(finals, dtype=np.float64).reshape(len(z), self.bins), self._sub_plan(vols, z), None if start is None else start[z])
        for (sig, ), (r, child, ops) in groups.items():
            z = np.array(r)
            self._push(nodes[z], np.array(child, dtype=np.int64), list(sig), self._stack_ops(sig, ops), self._sub_plan(vols, z),
                       None if start is None else start[z])

    @staticmethod
    def _stack_ops(sig, ops):
        out = []
        for j, tag in enumerate(sig):
            if tag in ("ms", "ds"):
                out.append(np.array([o[j] for o in ops], dtype=np.float64))
            elif tag[0] == "mad":
                out.append((np.array([o[j][0] for o in ops], dtype=np.float64), ops[0][j][1]))
            else:
                out.append(ops[0][j])
        return out

    # -- per-node evaluation through the full plugin API ----------------------------------------------------------------------------------
    @staticmethod
    def _hit_rows(hits, sel, p):
        t, ex, tri, uvw, g = (hits[k][sel] for k in ("t", "exiting", "tri", "uvw", "geom"))
        t, ex, tri, uvw, g = t.tolist(), ex.astype(bool).tolist(), tri.tolist(), uvw.astype(np.float64).tolist(), g.tolist()
        return [dict(t=t[i], ex=ex[i], tri=tri[i], uvw=uvw[i], g=g[i], prim=p) for i in range(len(t))]

    def _lazy_hit(self, row, node, prim):
        cls = _LazyMeshIntersection if row["tri"] >= 0 else _LazyIntersection
        h = cls.__new__(cls)
        h._src = (self, row, node, prim)
        return h

    def _to_classic(self, node, row):
        """The node's material read a deferred spectrum: remember its daughters, evaluate it again when they have finished."""
        self.children[node] = dict(self.cur_children)
        self.hitrow[node] = row
        self.stack.append(node)

    def _inside_list(self, node, flags):
        if flags is None or not flags.any():
            return ()
        o = self.st.o[node]
        return [self.prims[k] for k in self.flat.contains_order(_vec(Point3D, float(o[0]), float(o[1]), float(o[2]))) if flags[k]]

    def _evaluate_classic(self, node):
        """The body of Ray.trace after the roulette for one node, from the start: True when the node has finished."""
        st = self.st
        ray = self._ray_of(node)
        row = self.hitrow[node]
        prim = self.prims[row["prim"]]
        hit = self._lazy_hit(row, node, prim)
        stream = self.stream = _Stream(self.seed, int(st.pix[node]) | (int(st.mix[node]) << 40), int(st.smp[node]), 2 * int(st.depth[node]) + 1)
        known = self.children.setdefault(node, {})
        self.current, self.ordinal, self.cur_children, self.defer = node, 0, [], False
        previous = rsrandom.set_stream(stream)
        try:
            spectrum = ray._sample_surface(hit, self.world)
            spectrum = ray._sample_volumes(spectrum, hit, self.inside_of.get(node, ()), self.world)
            spectrum.mul_scalar(float(st.norm[node]))
        except _Pending:
            known.update(self.cur_children)
            return False
        finally:
            rsrandom.set_stream(previous)
            self.current, self.stream = -1, None
        st = self.st
        st.res[node] = spectrum.samples
        st.done[node] = True
        self.children.pop(node, None)
        self.hitrow.pop(node, None)
        self.inside_of.pop(node, None)
        return True

    def _do_generic(self, nodes, p, hits, sel, inside, vols, start):
        """Materials with their own evaluate_surface: called per node. With library volumes around the node the surface runs in
        deferral mode (one call); otherwise the classic scheme."""
        st = self.st
        prim = self.prims[p]
        rows = self._hit_rows(hits, sel, p)
        classic_volumes = inside is not None and bool((inside & self.user_volume[None, :]).any())
        for i, node in enumerate(nodes.tolist()):
            row = rows[i]
            flags = inside[i] if inside is not None else None
            user_vol = classic_volumes and bool((flags & self.user_volume).any())
            if user_vol or self.skind[p] is not None:
                # (a library surface under a user volume: the scalar host forms, whole node per call)
                self.hitrow[node] = row
                self.inside_of[node] = self._inside_list(node, flags)
                if not self._evaluate_classic(node):
                    self.stack.append(node)
                continue
            ray = self._ray_of(node)
            hit = self._lazy_hit(row, node, prim)
            stream = self.stream = _Stream(self.seed, int(st.pix[node]) | (int(st.mix[node]) << 40), int(st.smp[node]), 2 * int(st.depth[node]) + 1)
            self.current, self.ordinal, self.cur_children, self.defer = node, 0, [], True
            self.cur_mix, self.cur_depth, self.cur_pix, self.cur_smp = int(st.mix[node]), int(st.depth[node]), int(st.pix[node]), int(st.smp[node])
            previous = rsrandom.set_stream(stream)
            try:
                s = ray._sample_surface(hit, self.world)
                if type(s) is not _Deferred:
                    s.samples                               # (touch: a user Spectrum subclass must behave)
            except _Pending:
                self.inside_of[node] = self._inside_list(node, flags)
                self._to_classic(node, row)
                continue
            finally:
                rsrandom.set_stream(previous)
                self.current, self.defer, self.stream = -1, False, None
            st = self.st
            one = np.array([i])
            sub_vols, sub_start = self._sub_plan(vols, one), None if start is None else start[one]
            if type(s) is _Deferred:
                self._push(nodes[one], np.array([s.child], dtype=np.int64), list(s.sig), self._stack_ops(s.sig, [s.ops]), sub_vols, sub_start)
            else:
                self._finish(nodes[one], np.array(s.samples, dtype=np.float64).reshape(1, self.bins), sub_vols, sub_start)
        self._flush_users()

    # -- one wave ------------------------------------------------------------------------------------------------------------------------
    def _wave(self):
        st = self.st
        idx = np.concatenate(self.need_hit) if len(self.need_hit) > 1 else self.need_hit[0]
        self.need_hit = []
        hits, rows, inside_all = self.trace_wave(st.o[idx], st.d[idx], st.maxd[idx])
        prim = hits["prim"]
        st.done[idx[prim < 0]] = True                       # rays that leave the scene: zero spectrum, no volume pass (ray.pyx:391-393)
        if not len(rows):
            return
        any_inside = inside_all.any()
        hp = prim[rows]
        order = np.argsort(hp, kind="stable")
        bounds = np.nonzero(np.diff(hp[order]))[0] + 1
        for part in np.split(order, bounds):
            sel = rows[part]
            p = int(hp[part[0]])
            nodes = idx[sel]
            g = hits["geom"][sel]
            ex = hits["exiting"][sel].astype(bool)
            inside = inside_all[part] if any_inside else None
            if inside is not None and not inside.any():
                inside = None
            kind = self.skind[p]
            if kind is None or (inside is not None and (inside & self.user_volume[None, :]).any()):
                vols = self._volume_plan(nodes, inside)
                start = _xf_point(self.p2w[p], g[:, 0:3]) if vols else None
                self._do_generic(nodes, p, hits, sel, inside, vols, start)
                continue
            vols = self._volume_plan(nodes, inside)
            start = _xf_point(self.p2w[p], g[:, 0:3]) if vols else None
            if kind == "lambert":
                self._do_lambert(nodes, p, g, ex, vols, start)
            elif kind == "shade":
                self._do_shade(nodes, p, g, ex, hits, sel, inside, vols, start)
            elif kind == "dielectric":
                self._do_dielectric(nodes, p, g, ex, vols, start)
            elif kind == "null":
                self._do_null(nodes, p, g, ex, vols, start)
            elif kind == "absorber":
                self._finish(nodes, np.zeros((len(nodes), self.bins)), vols, start)
            elif kind == "emitter":
                mat = self.prims[p].material
                emission = mat.emission_spectrum.sample(self.lo, self.hi, self.bins) * mat.scale
                self._finish(nodes, np.repeat(emission[None, :], len(nodes), axis=0), vols, start)
            else:
                self._do_light(nodes, p, g, vols, start)

    def run(self, origin, direction, pixel, sample):
        """Primary rays as arrays (one row per ray): traces every path to its end, returns the spectra [n, bins] in row order."""
        n = len(pixel)
        t = self.template
        self.st = st = _Store(self.bins, 4 * n)
        idx = st.alloc(n)
        st.o[idx], st.d[idx], st.maxd[idx], st.parent[idx] = origin, direction, t.max_distance, -1
        st.pix[idx], st.smp[idx], st.norm[idx] = pixel, sample, 1.0
        self.rays += n
        self.need_hit.append(idx)
        saved = ray_module._scheduler
        ray_module._scheduler = self
        frozen = hasattr(self.world, "_spheres_cached")
        if frozen:
            self.world._important_frozen = True             # (validated once; frozen while materials are being evaluated)
        collecting = gc.isenabled()
        gc.disable()                                        # (millions of short-lived objects and no cycles: the collector only costs)
        try:
            while True:
                if self.need_hit:
                    self._wave()
                elif self.stack:
                    top = self.stack.pop()
                    if isinstance(top, _Group):
                        self._complete(top)
                    elif not self._evaluate_classic(top):   # (new daughters: traced next, finished before this node comes up again)
                        self.stack.append(top)
                else:
                    break
        finally:
            ray_module._scheduler = saved
            if frozen:
                self.world._important_frozen = False
            if collecting:
                gc.enable()
        return self.st.res[:n]


last_stats = []                 # run_block's timing of its latest worker runs (a tuning aid: tools/host_material_rate.py prints it)


def usable_cores():
    """Cores this process may use: the affinity mask cut by the cgroup quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


_libc = None


def _trim_heap():
    """Hands this process's free heap back to the kernel before it forks. Freed chunks that stay in the heap are inherited by every
    worker as copy-on-write pages, and the workers' (and this process's) next allocations land on exactly those pages: after one large
    render in the same process 17 processes fault on the same shared (huge) pages and everything runs four times slower
    (tools/r5_hybrid_workers.py: 2.9 s -> 10.8 s per 256 x 256 x 4 pass on the GPU box, transparent huge pages on). Trimmed pages come
    back zero-filled and private."""
    global _libc
    try:
        if _libc is None:
            import ctypes
            _libc = ctypes.CDLL("libc.so.6")
        _libc.malloc_trim(0)
    except (OSError, AttributeError):
        _libc = False


class _PipeScene:
    """The scene as a worker process sees it: every wave of rays is answered by the parent process, which owns the device."""

    def __init__(self, conn, flat):
        self.conn, self.flat, self.waited, self.waves = conn, flat, 0.0, 0

    def trace_wave(self, o, d, m):
        t0 = time.perf_counter()
        self.conn.send(("trace", o, d, m))
        r = self.conn.recv()
        self.waited += time.perf_counter() - t0
        self.waves += 1
        return r


def _worker(conn, world, flat, key, template, per_node, origin, direction, pixel, sample):
    try:
        t0 = time.perf_counter()
        pipe = _PipeScene(conn, flat)
        sched = WaveScheduler(world, pipe, key, template, per_node=per_node)
        spectra = sched.run(origin, direction, pixel, sample)
        conn.send(("done", np.ascontiguousarray(spectra), sched.rays, (time.perf_counter() - t0, pipe.waited, pipe.waves)))
    except BaseException:                                   # noqa: B036 (reported to the parent, which raises)
        conn.send(("error", traceback.format_exc()))
    finally:
        conn.close()
        os._exit(0)                                         # (no finalisers: the device objects inherited by the fork belong to the parent)


def run_block(world, scene, key, template, per_node, origin, direction, pixel, sample, workers):
    """Spectra [n, bins] of n primary rays and the number of rays traced. With workers > 1 the rays are split over forked
    processes (the reference's MulticoreEngine does the same with whole pixels, core/workflow.py:201-251): each runs its own
    WaveScheduler — its materials' Python — and sends its waves of rays to this process, the only one that talks to the device.
    Counter-based random streams: the frame does not depend on the split."""
    n = len(pixel)
    workers = min(int(workers), n // MIN_RAYS_PER_WORKER)
    if workers <= 1:
        sched = WaveScheduler(world, scene, key, template, per_node=per_node)
        return sched.run(origin, direction, pixel, sample), sched.rays
    ctx = multiprocessing.get_context("fork")
    t_fork = time.perf_counter()
    _trim_heap()
    bounds = np.linspace(0, n, workers + 1).astype(np.int64)
    flat = scene.flat
    host = scene.host_scene() if hasattr(scene, "host_scene") else None
    conns, procs, slot = [], [], {}
    # (the workers start with the cyclic collector off: a collection in a child would run the finalisers of whatever device-owning
    # garbage the parent had not collected yet — scenes and frames of earlier renders — against a device the child does not have)
    collecting = gc.isenabled()
    gc.disable()
    try:
        for w in range(workers):
            a, b = int(bounds[w]), int(bounds[w + 1])
            mine, theirs = ctx.Pipe()
            proc = ctx.Process(target=_worker, args=(theirs, world, flat, key, template, per_node, origin[a:b], direction[a:b], pixel[a:b], sample[a:b]),
                               daemon=True)
            proc.start()
            theirs.close()
            conns.append(mine)
            procs.append(proc)
            slot[mine] = (a, b)
    finally:
        if collecting:
            gc.enable()
    out, rays, failure = np.zeros((n, template.bins)), 0, None
    t_start = time.perf_counter()
    stats = dict(workers=workers, primary_rays=n, fork_s=t_start - t_fork, serve_s=0.0, trace_s=0.0, requests=0, request_rays=0, worker_s=[], worker_wait_s=[])
    try:
        live = list(conns)
        owner = dict(zip(conns, procs))
        quiet_since = time.perf_counter()
        while live:
            # A worker forked while another thread of this process held a runtime or allocator lock can hang before its first message
            # (the parent has live HIP / OpenMP threads): the wait is bounded, workers that died are reported, and a block in which no
            # live worker has said anything for WORKER_SILENCE_S seconds is abandoned with an error instead of hanging the render.
            ready = multiprocessing.connection.wait(live, timeout=1.0)
            if not ready:
                dead = [c for c in live if not owner[c].is_alive() and not c.poll()]
                for c in dead:
                    live.remove(c)
                    failure = failure or "a material worker process (pid %s) ended without an answer (exit code %s)" % (owner[c].pid, owner[c].exitcode)
                if live and time.perf_counter() - quiet_since > WORKER_SILENCE_S:
                    failure = failure or ("no material worker has answered for %.0f s (%d of %d still running): forked workers can deadlock on a lock "
                                          "another thread of the parent held at fork time; HipEngine(host_workers=1) evaluates in this process"
                                          % (WORKER_SILENCE_S, len(live), workers))
                    break
                continue
            quiet_since = time.perf_counter()
            for conn in ready:
                t0 = time.perf_counter()
                try:
                    msg = conn.recv()
                except EOFError:
                    msg = ("error", "a material worker process ended without an answer")
                if msg[0] == "trace":
                    t1 = time.perf_counter()
                    answer = trace_wave(scene, host, msg[1], msg[2], msg[3])
                    stats["trace_s"] += time.perf_counter() - t1
                    if host is not None and len(msg[1]) < HOST_WALK_BELOW:
                        stats["host_walk_s"] = stats.get("host_walk_s", 0.0) + time.perf_counter() - t1
                        stats["host_walk_requests"] = stats.get("host_walk_requests", 0) + 1
                    conn.send(answer)
                    stats["serve_s"] += time.perf_counter() - t0
                    stats["requests"] += 1
                    stats["request_rays"] += len(msg[1])
                    continue
                live.remove(conn)
                if msg[0] == "done":
                    a, b = slot[conn]
                    out[a:b] = msg[1]
                    rays += msg[2]
                    stats["worker_s"].append(round(msg[3][0], 3))
                    stats["worker_wait_s"].append(round(msg[3][1], 3))
                else:
                    failure = failure or msg[1]
        stats["wall_s"] = time.perf_counter() - t_start
        last_stats.append(stats)
        del last_stats[:-64]
    finally:
        for conn in conns:
            conn.close()
        for proc in procs:
            proc.join(timeout=5)
            if proc.is_alive():
                proc.kill()
    if failure:
        raise RuntimeError("host-callback material worker failed:\n" + failure)
    return out, rays


def welford(x, power_scale=None):
    """StatsArray _add_sample (core/math/statsarray.pyx:743-776) over axis 1 of x[n_pixels, spp, channels], in sample order."""
    n_pix, spp, ch = x.shape
    m = x[:, 0, :].copy()
    v = np.zeros((n_pix, ch))
    for i in range(1, spp):
        xi = x[:, i, :]
        pm, pv = m, v
        pn = i if i > 1 else 2
        m = pm + (xi - pm) / (i + 1)
        v = (pv * (pn - 1) + (xi - pm) * (xi - m)) / i
    return m, v


def render_slice(camera, tasks, slice_id, template, engine, pieces):
    """One spectral slice of observe() through the host-callback path; called by PinholeCamera._render_slice_device."""
    from ..device import combine_arrays
    from .observer import RGBPipeline2D, RectTasks
    world = camera.root
    scene = world.build_accelerator()
    sl = camera._slices[slice_id]
    nx, ny = camera._pixels
    spp = camera._pixel_samples
    offset = getattr(camera, "_pass_offset", None)
    if offset is None:
        offset = engine.sample_offset
    key = (engine.seed + sl.offset * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF            # as render_desc: one Philox key per slice
    m = camera.to_root().m
    delta = camera.image_delta
    half = 0.5 * delta
    wq = m[12] * 0.0 + m[13] * 0.0 + m[14] * 0.0 + m[15]
    wq = 1.0 / wq
    origin = ((m[0] * 0.0 + m[1] * 0.0 + m[2] * 0.0 + m[3]) * wq, (m[4] * 0.0 + m[5] * 0.0 + m[6] * 0.0 + m[7]) * wq,
              (m[8] * 0.0 + m[9] * 0.0 + m[10] * 0.0 + m[11]) * wq)
    per_node = bool(getattr(engine, "per_node_materials", False)) or os.environ.get("RSX_HYBRID_PER_NODE", "0") == "1"
    if hasattr(world, "_spheres_cached"):
        world._spheres_cached()
    workers = getattr(engine, "host_workers", None)
    if workers is None:
        workers = int(os.environ.get("RSX_HOST_WORKERS", "0")) or min(usable_cores(), 16)
    if not python_materials(world, per_node):
        workers = 1                                         # (array forms only: nothing for more processes to do)
    for piece in pieces:
        block = piece["tasks"] if "tasks" in piece else list(RectTasks(*piece["rect"]))
        t = np.array(block, dtype=np.int64).reshape(-1, 2)
        n = len(t)
        ix, iy = np.repeat(t[:, 0], spp), np.repeat(t[:, 1], spp)
        s = np.tile(np.arange(spp, dtype=np.int64), n)
        if engine.rng == "stream":
            u = rsrandom.uniform_block(2 * n * spp)
            u1, u2 = u[0::2], u[1::2]
        else:
            u1, u2 = P.philox2_array(key, (ix * ny + iy).astype(np.uint64), (offset + s).astype(np.uint64))
        # PinholeCamera._generate_rays (pinhole.pyx:169-204), the operations of k_render_trace
        pixel_x = camera.image_start_x - delta * (ix + 0.5)
        pixel_y = camera.image_start_y - delta * (iy + 0.5)
        dx, dy, dz = (u2 * delta - half) + pixel_x, (u1 * delta - half) + pixel_y, np.full(n * spp, 0.0 + 1.0)
        norm = dx * dx + dy * dy + dz * dz
        norm = 1.0 / np.sqrt(norm)
        dx, dy, dz = dx * norm, dy * norm, dz * norm
        weight = dz
        wx, wy, wz = m[0] * dx + m[1] * dy + m[2] * dz, m[4] * dx + m[5] * dy + m[6] * dz, m[8] * dx + m[9] * dy + m[10] * dz
        spectra, traced = run_block(world, scene, key, template, per_node, np.broadcast_to(np.array(origin), (n * spp, 3)),
                                    np.stack((wx, wy, wz), axis=1), (ix * ny + iy).astype(np.uint64), (offset + s).astype(np.uint64), workers)
        spectra = spectra.reshape(n, spp, sl.bins) * weight.reshape(n, spp, 1)      # projection weight (observer.pyx:408)
        camera.stats["rays"] = camera.stats.get("rays", 0) + traced
        for pipe in camera._pipelines:
            if isinstance(pipe, RGBPipeline2D):             # XYZPixelProcessor.add_sample (rgb.pyx:534-562)
                curves, d_wl = pipe._resampled[slice_id], pipe._deltas[slice_id]
                xyz = np.zeros((n, spp, 3))
                for b in range(sl.bins):
                    xyz += d_wl * spectra[:, :, b:b + 1] * curves[b].reshape(1, 1, 3)
                mean, var = welford(xyz * camera._sensitivity)
                pipe.update_block(t[:, 0], t[:, 1], mean, var)
                continue
            mean, var = welford(spectra * camera._sensitivity if pipe.power else spectra)
            f = pipe.frame
            f._sync_host()
            z = slice(sl.offset, sl.offset + sl.bins)
            xs, ys = t[:, 0], t[:, 1]
            cm, cv, cn = combine_arrays(f._host[0][xs, ys, z], f._host[1][xs, ys, z], f._host[2][xs, ys, z], mean, np.maximum(var, 0.0),
                                        np.full(mean.shape, spp))
            f._host[0][xs, ys, z], f._host[1][xs, ys, z], f._host[2][xs, ys, z] = cm, cv, cn
            f._host_written()
