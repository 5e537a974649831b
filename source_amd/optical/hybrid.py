"""
Host-callback render path: observe() for scenes whose materials have no device lowering (user-written Material subclasses).

The reference calls Material.evaluate_surface / evaluate_volume once per hit from inside Ray.trace (raysect/optical/ray.pyx:338-455,
material/material.pxd:36-47), and a material obtains incoming light by calling daughter.trace(world) recursively. That plugin API is
kept as it is; what changes is who traces the rays. Every ray — primary or daughter — becomes a node of its path's call tree and is
traced on the MI355X together with all other rays that are ready (one rsx_hit_batch + one rsx_contains_batch per wave of rays),
never alone:

  * a node whose hit is known is *evaluated*: its material's evaluate_surface runs on the host, then the volume pass, then the
    Russian-roulette normalisation — the body of Ray.trace;
  * when the material calls daughter.trace(world) the scheduler looks the daughter up among the node's children: a finished
    daughter returns its spectrum; a new one is registered (roulette is decided on the spot, ray.pyx:380-388), queued for the next
    wave, and the evaluation is abandoned by raising _Pending;
  * when a daughter finishes, its parent is evaluated again from the start. Evaluation is deterministic — each node draws from
    its own counter-based random stream, rewound at every evaluation — so the second run takes the same decisions, finds its
    daughter finished and completes. A material that traces k daughters is evaluated k + 1 times; nothing else is repeated.

Random numbers follow librsx's Philox convention (include/rsx.h): for the depth-d ray of sample s of pixel p, draw 2d decides
roulette and draw 2d + 1 feeds the scattering; the host forms of Lambert and Dielectric (material.py) consume them exactly like the
device kernels, so a scene rendered through this path gives the same frame as the device path, bit for bit — that is how the path is
tested (tests/test_gpu_parity.py::test_host_callback_path_*). Sibling daughters (a material that traces several rays) get
decorrelated streams.

Per-pixel statistics use the same Welford recurrence in sample order and the same combine_samples merge as the device kernels.
"""
import numpy as np

from ..core import random as rsrandom
from ..core.math import Point3D, Vector3D
from . import _portable as P
from . import ray as ray_module


class _Pending(BaseException):
    """Raised through a material's evaluate_surface when it asks for a daughter ray that has not been traced yet."""


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


class _Node:
    """One ray of a path's call tree. path = (pixel word, sample counter) of the Philox counters; ordinal = position among the
    parent's daughters (the primary ray keeps its (pixel, sample, row) key there); mix = stream decorrelation word (0 on the chain
    the device numbers the same way)."""
    __slots__ = ("path", "parent", "ordinal", "depth", "mix", "ray", "norm", "hit", "inside", "result", "children", "stream")

    def __init__(self, path, parent, ordinal, depth, mix, ray):
        self.path, self.parent, self.ordinal, self.depth, self.mix, self.ray = path, parent, ordinal, depth, mix, ray
        self.norm, self.hit, self.inside, self.result, self.children, self.stream = 1.0, None, (), None, None, None


class WaveScheduler:
    """Traces the paths of one block of pixels of one spectral slice. seed / sample counters as in rsx_render_desc."""

    def __init__(self, world, scene, seed):
        self.world, self.scene, self.seed = world, scene, int(seed)
        self.need_hit = []
        self.current, self.ordinal = None, 0
        self.rays = 0
        self.results = {}

    # -- called by Ray.trace of a daughter, from inside a material --------------------------------------------------------
    def trace(self, ray, world, keep_alive):
        cur = self.current
        if cur is None:
            raise RuntimeError("Ray.trace() inside a host-callback render must be called from a material's evaluate_surface / evaluate_volume")
        ordinal = self.ordinal
        self.ordinal += 1
        if cur.children is None:
            cur.children = {}
        child = cur.children.get(ordinal)
        if child is None:
            child = self._spawn(cur, ordinal, ray, keep_alive)
            cur.children[ordinal] = child
        if child.result is None:
            raise _Pending()
        return child.result.copy()                          # the caller scales it in place

    def _stream(self, node, draw):
        path = node.path
        return _Stream(self.seed, path[0] | (node.mix << 40), path[1], draw)

    def _spawn(self, parent, ordinal, ray, keep_alive):
        # stream identity: the first daughter of a chain keeps mix = 0 (the device's numbering); siblings, and same-depth daughters
        # of a node that consumed random numbers itself, move to decorrelated counters
        mix = parent.mix
        if ordinal > 0 or (ray.depth == parent.depth and parent.stream is not None and parent.stream.pos > 0):
            mix = (mix * 0x9E3779B1 + ordinal + 1 + 0x7F4A7C15 * (parent.depth + 1)) & 0xFFFFF or 1
        node = _Node(parent.path, parent, ordinal, ray.depth, mix, ray)
        self.rays += 1
        if not (keep_alive or ray.depth < ray.extinction_min_depth):        # ray.pyx:380-388
            if ray.depth >= ray.max_depth or self._stream(node, 2 * ray.depth).next() < ray.extinction_prob:
                node.result = ray.new_spectrum()
                return node
            node.norm = 1 / (1 - ray.extinction_prob)
        self.need_hit.append(node)
        return node

    # -- one node = the body of Ray.trace after the roulette ----------------------------------------------------------------
    def _evaluate(self, node):
        ray = node.ray
        if node.hit is None:
            node.result = ray.new_spectrum()                # a ray that leaves the scene: zero spectrum, no volume pass (ray.pyx:391-393)
            return True
        if node.stream is None:
            node.stream = self._stream(node, 2 * node.depth + 1)
        node.stream.pos = 0
        self.current, self.ordinal = node, 0
        previous = rsrandom.set_stream(node.stream)
        try:
            spectrum = ray._sample_surface(node.hit, self.world)
            spectrum = ray._sample_volumes(spectrum, node.hit, node.inside, self.world)
            spectrum.mul_scalar(node.norm)
        except _Pending:
            return False
        finally:
            rsrandom.set_stream(previous)
            self.current = None
        node.result = spectrum
        node.children = None                                # the daughters' spectra are not needed any more
        return True

    def run(self, primaries):
        """primaries: list of (key, Ray): traces every path to its end; self.results[key] = Spectrum."""
        scene, flat = self.scene, self.scene.flat
        for key, ray in primaries:
            node = _Node((key[0], key[1]), None, 0, 0, 0, ray)
            node.ordinal = key
            self.need_hit.append(node)
            self.rays += 1
        saved = ray_module._scheduler
        ray_module._scheduler = self
        if hasattr(self.world, "_spheres_cached"):
            self.world._spheres_cached()                    # (validated once; frozen while materials are being evaluated)
            self.world._important_frozen = True
        try:
            while self.need_hit:
                batch, self.need_hit = self.need_hit, []
                o = np.array([(n.ray.origin.x, n.ray.origin.y, n.ray.origin.z) for n in batch], dtype=np.float64)
                d = np.array([(n.ray.direction.x, n.ray.direction.y, n.ray.direction.z) for n in batch], dtype=np.float64)
                m = np.array([n.ray.max_distance for n in batch], dtype=np.float64)
                hits = scene.hit_batch(o, d, m, geometry=True)
                hit_rows = np.nonzero(hits["prim"] >= 0)[0]
                inside = scene.contains_batch(o[hit_rows]) if len(hit_rows) else None
                for j, i in enumerate(hit_rows):
                    n = batch[i]
                    obj = flat.records[int(hits["prim"][i])]["obj"]
                    n.hit = scene._intersection(n.ray, obj, hits["t"][i], hits["exiting"][i], hits["tri"][i], hits["uvw"][i], hits["geom"][i])
                    flags = inside[j]
                    if flags.any():                         # world.contains(origin) in the world tree's leaf order (world.pyx:149-168)
                        n.inside = [self.world._primitives[k] for k in flat.contains_order(n.ray.origin) if flags[k]]
                stack = list(batch)
                while stack:
                    n = stack.pop()
                    if n.result is None and not self._evaluate(n):
                        continue
                    if n.parent is None:
                        self.results[n.ordinal] = n.result
                    elif n.parent.result is None:
                        stack.append(n.parent)
        finally:
            ray_module._scheduler = saved
            self.world._important_frozen = False
        return self.results


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
        sched = WaveScheduler(world, scene, key)
        primaries = []
        for r in range(n * spp):
            ray = template.copy(Point3D(*origin), Vector3D(float(wx[r]), float(wy[r]), float(wz[r])))
            primaries.append(((int(ix[r]) * ny + int(iy[r]), int(offset + s[r]), r), ray))
        results = sched.run(primaries)
        spectra = np.zeros((n, spp, sl.bins))
        for (_, _, r), spectrum in results.items():
            spectra[r // spp, r % spp, :] = spectrum.samples
        spectra *= weight.reshape(n, spp, 1)                # projection weight (observer.pyx:408)
        camera.stats["rays"] = camera.stats.get("rays", 0) + sched.rays
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
