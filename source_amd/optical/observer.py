"""
Observer.observe() — the drop-in top of the hot path.

The structure of the reference's driver is kept (raysect/optical/observer/base/observer.pyx:265-446):
slice the spectrum, build one ray template per slice, initialise the pipelines, ask the frame sampler for
tasks, then hand each slice's full task list to ``render_engine.run(...)`` — the reference's plug-point #2
(raysect/core/workflow.py:78-91). Where the reference's engines farm pixels over CPU processes, HipEngine
submits the whole slice to the MI355X: rsx_render_pinhole_frame traces every (pixel, sample) and merges the
per-pixel statistics straight into the pipeline's device-resident frame (combine_samples law).

Mirrors: observer/base/observer.pyx, base/slice.pyx, sampler2d.pyx:42-102, imaging/pinhole.pyx:42-207,
pipeline/spectral/power.pyx:335-486, radiance.pyx:182-263, core/math/statsarray.pyx:513-565, core/workflow.py.
"""
import ctypes as C
import math
import os
import weakref
import random as pyrandom

import numpy as np

from .. import _lib
from ..core import random as rsrandom
from ..core.scenegraph import Observer, World
from .ray import Ray


class SpectralSlice:
    """base/slice.pyx:32-69"""

    def __init__(self, min_wavelength, max_wavelength, total_bins, bins, offset):
        self.total_bins, self.bins, self.offset = total_bins, bins, offset
        delta = (max_wavelength - min_wavelength) / total_bins
        self.min_wavelength = min_wavelength + delta * offset
        self.max_wavelength = min_wavelength + delta * (offset + bins)


# ---------------------------------------------------------------------------------------------------
# render engines — raysect/core/workflow.py
# ---------------------------------------------------------------------------------------------------
class RenderEngine:
    """workflow.py:35-97"""

    def run(self, tasks, render, update, render_args=(), render_kwargs={}, update_args=(), update_kwargs={}):
        raise NotImplementedError("Virtual method must be implemented in sub-class.")

    def worker_count(self):
        raise NotImplementedError("Virtual method must be implemented in sub-class.")


class HipEngine(RenderEngine):
    """
    MI355X render engine. ``rng="philox"`` (default): on-device counter-based RNG keyed by (seed; pixel, sample),
    independent of task order and of how pixels are sharded over GPUs. ``rng="stream"``: the jitter samples are
    drawn from the host MT19937-64 stream (source_amd.core.random, same generator and consumption order as the
    reference's SerialEngine: 2 uniforms per sample in task order) — bit-parity mode for primary-ray scenes.
    ``fused=True`` merges results into the pipelines' device frames; ``fused=False`` follows the RenderEngine
    contract literally (update(packed_result) per task), which is what a stock raysect pipeline needs.

    Philox sample counters: pass p of an observer draws the counters ``sample_offset + p * pixel_samples ...`` — consecutive
    observe() calls of one observer never reuse a (pixel, sample) counter, like the reference's engines, which draw fresh random
    numbers every pass. Assigning ``engine.sample_offset`` between passes restarts the count from the assigned value (that is how
    the multi-GPU shards place their samples, source_amd/distributed.py); a sample-sharded process that does NOT reassign it every
    pass must set ``sample_stride`` to the number of processes (otherwise pass p + 1 of rank r would reuse the counters of pass p of
    rank r + 1 and correlated samples would be merged as if independent).

    ``auto_batch`` (default on): small consecutive passes need no opt-in. An observe() whose pass is small (pixel_samples < 64, a
    rectangle of pixels, accumulating spectral pipelines, closed-form materials, this engine in its default Philox form) returns at
    once; up to 64 / pixel_samples such calls in a row are submitted as ONE library call — exactly the call passes_per_call would
    make, so the frames are those of the separate passes, bit for bit — when the batch is full, when the device runs idle, or the moment
    anything reads or replaces a frame, synchronises the context, changes the scenegraph, or changes what the next pass would render. The reference's
    usual loop (`while not camera.render_complete: camera.observe()` with a display or a save per pass) reads the frame every pass
    and sees no difference; a loop that only accumulates runs several times faster (configs[1]: 2.5 -> ~9 G rays/s).

    ``passes_per_call=K``: one observe() renders K consecutive passes of ``pixel_samples`` samples each as ONE library call per spectral
    slice (rsx_render_desc.passes) and leaves the frames that K observe() calls would — bit for bit, the K merges included. For
    accumulating spectral pipelines on the fused Philox path, scenes without scattering or volume materials; anything else raises.
    A one-sample pass of a megapixel frame is a fraction of a millisecond of device work: K of them in one launch cost neither K launch
    tails nor waves whose 64 rays cross 64 pixels.
    """

    def __init__(self, rng="philox", seed=0, fused=True, timing=False, sample_offset=0, host_materials=False, sample_stride=1, slice_range=None,
                 passes_per_call=1, auto_batch=None, per_node_materials=False, host_workers=None):
        if rng not in ("philox", "stream"):
            raise ValueError("rng must be 'philox' or 'stream'")
        # multi-process renders (source_amd/distributed.py). sample_stride = N with sample_offset = rank * pixel_samples: pass p of this
        # process draws the counters (p * N + rank) * pixel_samples ..., so N sample-sharded processes never share a (pixel, sample)
        # counter however many passes they run; slice_range = (first, last): render those spectral slices only (slice sharding).
        self.sample_stride = int(sample_stride)
        self.slice_range = None if slice_range is None else (int(slice_range[0]), int(slice_range[1]))
        self.rng, self.seed, self.fused = rng, int(seed), bool(fused)
        self.host_materials = bool(host_materials)    # True: evaluate every material on the host (source_amd/optical/hybrid.py) even when
                                                      # all of them have device lowerings — scenes with a user-written material always do
        # host-callback path only (hybrid.py). per_node_materials: call every material per node through the plugin API instead of the
        # library's array forms (what a user-written evaluate_surface gets anyway; a test aid). host_workers: processes that evaluate
        # Python materials (None: min(cores, 16); 1: in this process) — see hybrid.render_slice
        self.per_node_materials = bool(per_node_materials)
        self.host_workers = None if host_workers is None else max(1, int(host_workers))
        self.timing = bool(timing)          # True: read back HIP-event kernel times after each library call (one stream sync per call:
                                            # a tuning aid, it defeats the pipelined render lanes)
        self.sample_offset = int(sample_offset)
        self.passes_per_call = int(passes_per_call)
        if self.passes_per_call < 1:
            raise ValueError("passes_per_call must be at least 1")
        # auto_batch (default on; RSX_AUTO_BATCH=0 or auto_batch=False turns it off): consecutive observe() calls whose passes are
        # small (pixel_samples < 64) are accepted at once and submitted together — as the ONE library call passes_per_call would have
        # made — when the batch fills a 64-ray unit per pixel, or as soon as anything looks at the frames (see PinholeCamera._lazy_pass)
        self.auto_batch = (os.environ.get("RSX_AUTO_BATCH", "1") != "0") if auto_batch is None else bool(auto_batch)
        # (a partial batch is submitted as soon as the device is idle — rsx_idle — instead of waiting to be full; RSX_EAGER_BATCH=0: only full
        # batches and reads submit. How many passes a library call carries never shows in the frames.)
        self.eager_batch = os.environ.get("RSX_EAGER_BATCH", "1") != "0"
        # samples per pixel a partial batch must hold to go out early. Swept on configs[1] (tools/r5_observe_times.py; rays/s at 20 / 64 / 200
        # observe() calls): never 3.9 / 6.4 / 8.5e9, from 16 on 4.7 / 7.5 / 8.5e9, from 4 or 8 on 5.2 / 5.5 / 2.0e9 — small batches are served by
        # launches that cost four times as much per pass, and new batch sizes meet first-use costs (80 ms stalls in the timed loop)
        self.eager_min = int(os.environ.get("RSX_EAGER_MIN", "16"))
        self.last_kernel_ms = None

    def worker_count(self):
        return 1

    def run(self, tasks, render, update, render_args=(), render_kwargs={}, update_args=(), update_kwargs={}):
        observer = render.__self__
        slice_id, template = render_args
        observer._render_slice_device(tasks, slice_id, template, self, update, update_args, update_kwargs)


class SerialEngine(HipEngine):
    """Drop-in for raysect.core.SerialEngine (workflow.py:100-146): same constructor, and — for the closed-form materials this
    round covers — the same frames bit for bit, because the jitter is drawn from the reference's MT19937-64 stream in task order."""

    def __init__(self):
        super().__init__(rng="stream")


class MulticoreEngine(HipEngine):
    """Drop-in for raysect.core.MulticoreEngine (workflow.py:149-326). The reference farms tasks to `processes` workers, each
    re-seeded from os.urandom (workflow.py:302-305), so it defines no reproducible stream; here the work goes to the GPU with the
    counter-based generator. `processes`, `tasks_per_job` and `start_method` are accepted and ignored."""

    def __init__(self, processes=None, tasks_per_job=None, start_method="fork", seed=None):
        import os as _os
        super().__init__(rng="philox", seed=int.from_bytes(_os.urandom(8), "little") if seed is None else seed)
        self.processes, self.tasks_per_job, self.start_method = processes, tasks_per_job, start_method


# ---------------------------------------------------------------------------------------------------
# frame sampler / pipelines
# ---------------------------------------------------------------------------------------------------
class FrameSampler2D:
    def generate_tasks(self, pixels):
        raise NotImplementedError


class RectTasks:
    """A rectangular block of pixels [x0,x1) x [y0,y1) in natural order, handed to the device as four integers instead
    of a Python list of tuples (1024^2 tuples cost more host time than the GPU needs to render them)."""

    def __init__(self, x0, y0, x1, y1):
        self.rect = (int(x0), int(y0), int(x1), int(y1))

    def __len__(self):
        return max(0, self.rect[2] - self.rect[0]) * max(0, self.rect[3] - self.rect[1])

    def __iter__(self):
        x0, y0, x1, y1 = self.rect
        return ((ix, iy) for iy in range(y0, y1) for ix in range(x0, x1))


class RectFrameSampler2D(FrameSampler2D):
    """Whole frame (or one tile of it: the multi-GPU tile shard) as a single RectTasks block; needs rng='philox'."""

    def __init__(self, rect=None):
        self.rect = rect

    def generate_tasks(self, pixels):
        return RectTasks(*(self.rect or (0, 0, pixels[0], pixels[1])))


class FullFrameSampler2D(FrameSampler2D):
    """sampler2d.pyx:42-102: every unmasked pixel, iy-outer / ix-inner, then Python random.shuffle."""

    def __init__(self, mask=None):
        self.mask = None if mask is None else np.asarray(mask, dtype=bool)

    def generate_tasks(self, pixels):
        nx, ny = pixels
        if self.mask is None or (self.mask.shape != tuple(pixels) and self.mask.all()):
            self.mask = np.ones(pixels, dtype=bool)
        elif self.mask.shape != tuple(pixels):
            raise ValueError("The pixel geometry passed to the frame sampler is inconsistent with the mask shape.")
        tasks = [(ix, iy) for iy in range(ny) for ix in range(nx) if self.mask[ix, iy]]
        pyrandom.shuffle(tasks)
        return tasks


LIBRARY_CALLS = [0]        # rsx_render_pinhole_frame calls made by this process (bench.py: launches per step when small passes are batched)


class StatsArray3D:
    """
    core/math/statsarray.pyx:513-565 — (mean f64, variance f64, samples i32)[nx, ny, nz], x-major. The arrays
    live in HBM while rendering; .mean/.variance/.samples download (and cache) host copies on access.
    """

    def __init__(self, nx, ny, nz):
        self.nx, self.ny, self.nz = int(nx), int(ny), int(nz)
        self._host = [np.zeros((nx, ny, nz)), np.zeros((nx, ny, nz)), np.zeros((nx, ny, nz), dtype=np.int32)]
        self._dev = None          # (ctx wrapper, mean ptr, var ptr, n ptr)
        self._dev_dirty = False

    @property
    def shape(self):
        return (self.nx, self.ny, self.nz)

    @property
    def length(self):
        return self.nx * self.ny * self.nz

    def _settle(self):
        """Passes an observer has accepted but not yet submitted (PinholeCamera: consecutive small passes are batched into one
        library call) are rendered before anybody looks at — or replaces — the frame."""
        owners = getattr(self, "_lazy_owners", None)
        if owners:
            for owner in list(owners):                      # (several observers may accumulate into one pipeline's frame)
                owner._flush_lazy()

    def _device(self, context, settle=True):
        """Device pointers of the frame (allocated and uploaded on first use)."""
        if settle:
            self._settle()
        if self._dev is None or self._dev[0] is not context:
            self._sync_host()
            ptrs = [context.alloc(a.nbytes) for a in self._host]
            for p, a in zip(ptrs, self._host):
                context.upload(p, a)
            self._dev = (context, *ptrs)
        return self._dev[1:]

    def bind_device(self, context, mean_ptr, var_ptr, n_ptr):
        """Use caller-owned device memory (e.g. torch tensors' data_ptr()) as the frame storage. The caller zeroes it."""
        self._settle()
        self._dev = (context, C.c_void_p(mean_ptr), C.c_void_p(var_ptr), C.c_void_p(n_ptr))
        self._external = True
        self._dev_dirty = True

    def _sync_host(self):
        self._settle()
        if self._dev is not None and self._dev_dirty:
            ctx = self._dev[0]
            for p, a in zip(self._dev[1:], self._host):
                ctx.download(a, p)
            self._dev_dirty = False

    def _mark_device_written(self):
        self._dev_dirty = True

    def _host_written(self):
        """The host arrays were modified (host-callback render path): bring the device copy, if there is one, up to date."""
        self._settle()
        if self._dev is not None:
            ctx = self._dev[0]
            for p, a in zip(self._dev[1:], self._host):
                ctx.upload(p, a)
            self._dev_dirty = False

    @property
    def mean(self):
        self._sync_host()
        return self._host[0]

    @property
    def variance(self):
        self._sync_host()
        return self._host[1]

    @property
    def samples(self):
        self._sync_host()
        return self._host[2]

    def error(self, x, y, z):                               # statsarray.pyx:728-739
        n, v = self.samples[x, y, z], self.variance[x, y, z]
        return 0.0 if n <= 0 or v <= 0 else math.sqrt(v / n)

    def errors(self):
        n, v = self.samples, self.variance
        with np.errstate(divide="ignore", invalid="ignore"):
            return np.where((n > 0) & (v > 0), np.sqrt(v / np.maximum(n, 1)), 0.0)

    def release(self):
        if self._dev is not None and not getattr(self, "_external", False):
            self._sync_host()
            ctx = self._dev[0]
            for p in self._dev[1:]:
                ctx.free(p)
            self._dev = None

    def __del__(self):
        try:
            if self._dev is not None and not getattr(self, "_external", False):
                ctx = self._dev[0]
                for p in self._dev[1:]:
                    ctx.free(p)
        except Exception:
            pass


class Pipeline2D:
    """base/pipeline.pyx:187-266"""
    power = False

    def initialise(self, pixels, pixel_samples, min_wavelength, max_wavelength, spectral_bins, spectral_slices, quiet):
        raise NotImplementedError

    def update(self, x, y, slice_id, packed_result):
        raise NotImplementedError

    def finalise(self):
        raise NotImplementedError


class SpectralPowerPipeline2D(Pipeline2D):
    """pipeline/spectral/power.pyx:335-437 — spectral power per pixel (W/nm): samples are scaled by pixel sensitivity."""
    power = True

    def __init__(self, accumulate=True, name=None):
        self.name = name or "Spectral Pipeline 2D"
        self.accumulate = accumulate
        self.frame = None
        self._pixels = None
        self._samples = 0
        self._spectral_slices = None
        self.min_wavelength = self.max_wavelength = self.delta_wavelength = 0
        self.bins = 0
        self.wavelengths = None

    def initialise(self, pixels, pixel_samples, min_wavelength, max_wavelength, spectral_bins, spectral_slices, quiet):
        nx, ny = pixels
        self._pixels, self._samples, self._spectral_slices = pixels, pixel_samples, spectral_slices
        self.min_wavelength, self.max_wavelength = min_wavelength, max_wavelength
        self.delta_wavelength = (max_wavelength - min_wavelength) / spectral_bins
        self.bins = spectral_bins
        self.wavelengths = np.array([min_wavelength + (0.5 + i) * self.delta_wavelength for i in range(spectral_bins)])
        if not self.accumulate or self.frame is None or self.frame.shape != (nx, ny, spectral_bins):
            if self.frame is not None:
                self.frame.release()
            self.frame = StatsArray3D(nx, ny, spectral_bins)

    def update(self, x, y, slice_id, packed_result):        # power.pyx:424-437 (host path of the RenderEngine contract)
        mean, variance = packed_result
        sl = self._spectral_slices[slice_id]
        f = self.frame
        f._sync_host()
        from ..device import combine_scalar
        for i in range(sl.bins):
            z = sl.offset + i
            m, v, n = combine_scalar(f._host[0][x, y, z], f._host[1][x, y, z], int(f._host[2][x, y, z]),
                                     float(mean[i]), max(0.0, float(variance[i])), int(self._samples))
            f._host[0][x, y, z], f._host[1][x, y, z], f._host[2][x, y, z] = m, v, n

    def finalise(self):
        pass


class SpectralRadiancePipeline2D(SpectralPowerPipeline2D):
    """pipeline/spectral/radiance.pyx:182-263 — spectral radiance per pixel (W/str/m^2/nm)."""
    power = False


class RGBPipeline2D(Pipeline2D):
    """pipeline/rgb.pyx:48-289 — CIE XYZ per pixel (the sRGB image is a view of it). Per spectral slice every sample's spectrum is
    projected on the slice's resampled XYZ curves and the three channels go through a Welford accumulator (XYZPixelProcessor, on the
    device: rsx_render_pinhole_xyz); update() sums the slices' means and variances into working frames, finalise() merges the working
    frame into xyz_frame with combine_samples. The display / auto-exposure machinery of the reference is not mirrored; the
    display_* keyword arguments are accepted so that existing scripts construct the pipeline unchanged."""
    power = True

    def __init__(self, display_progress=False, display_update_time=15, accumulate=True, display_auto_exposure=True,
                 display_sensitivity=1.0, display_unsaturated_fraction=1.0, name=None):
        self.name = name or "RGB Pipeline 2D"
        self.accumulate = accumulate
        self.display_sensitivity = display_sensitivity
        self.xyz_frame = None
        self._pixels, self._samples = None, 0
        self._resampled = []
        self._working_mean = self._working_variance = self._working_touched = None

    def initialise(self, pixels, pixel_samples, min_wavelength, max_wavelength, spectral_bins, spectral_slices, quiet):
        from .colour import resample_ciexyz
        nx, ny = pixels
        self._pixels, self._samples = pixels, pixel_samples
        if not self.accumulate or self.xyz_frame is None or self.xyz_frame.shape != (nx, ny, 3):
            self.xyz_frame = StatsArray3D(nx, ny, 3)
        self._working_mean = np.zeros((nx, ny, 3))
        self._working_variance = np.zeros((nx, ny, 3))
        self._working_touched = np.zeros((nx, ny), dtype=np.int8)
        self._resampled = [np.ascontiguousarray(resample_ciexyz(sl.min_wavelength, sl.max_wavelength, sl.bins)) for sl in spectral_slices]
        self._deltas = [(sl.max_wavelength - sl.min_wavelength) / sl.bins for sl in spectral_slices]

    def update(self, x, y, slice_id, packed_result):        # rgb.pyx:249-271
        mean, variance = packed_result
        for c in range(3):
            self._working_mean[x, y, c] += mean[c]
            self._working_variance[x, y, c] += variance[c]
        self._working_touched[x, y] = 1

    def update_block(self, xs, ys, mean, variance):
        """update() for a whole block of pixels of one slice (pixels of a block are distinct)."""
        self._working_mean[xs, ys, :] += mean
        self._working_variance[xs, ys, :] += variance
        self._working_touched[xs, ys] = 1

    def finalise(self):                                     # rgb.pyx:276-289
        from ..device import combine_arrays
        f = self.xyz_frame
        f._sync_host()
        t = self._working_touched == 1
        if t.any():
            m, v, n = combine_arrays(f._host[0][t], f._host[1][t], f._host[2][t], self._working_mean[t],
                                     np.maximum(self._working_variance[t], 0.0), np.full(self._working_mean[t].shape, self._samples))
            f._host[0][t], f._host[1][t], f._host[2][t] = m, v, n

    @property
    def rgb_frame(self):
        """[nx, ny, 3] sRGB image of xyz_frame.mean * display_sensitivity (colour.pyx:235-266)."""
        from .colour import ciexyz_to_srgb
        xyz = self.xyz_frame.mean * self.display_sensitivity
        out = np.zeros_like(xyz)
        for x in range(xyz.shape[0]):
            for y in range(xyz.shape[1]):
                out[x, y] = ciexyz_to_srgb(*xyz[x, y])
        return out


    def save(self, filename):
        """rgb.pyx:517-531 — writes the sRGB image as an 8-bit PNG (row = image y, as the reference's transposed imsave). The
        reference's auto-exposure is not mirrored: the image is xyz_frame.mean * display_sensitivity."""
        import struct
        import zlib
        img = np.clip(np.transpose(self.rgb_frame, (1, 0, 2)) * 255.0 + 0.5, 0, 255).astype(np.uint8)
        h, w, _ = img.shape
        raw = b"".join(b"\x00" + img[y].tobytes() for y in range(h))

        def chunk(tag, data):
            return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
        with open(filename, "wb") as f:
            f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
                    chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


class RGBAdaptiveSampler2D(FrameSampler2D):
    """sampler2d.pyx:697-896 — re-samples the pixels whose normalised standard error (worst of X, Y, Z) lies in the top `fraction`
    of the image or above `cutoff`, and every pixel that has fewer than max(min_samples, max_samples / ratio) samples."""

    def __init__(self, pipeline, fraction=0.2, ratio=10.0, min_samples=1000, cutoff=0.0, mask=None):
        if not isinstance(pipeline, RGBPipeline2D):
            raise TypeError("Sampler only compatible with RGBPipeline2D pipeline.")
        if fraction <= 0 or fraction > 1.0:
            raise ValueError("Attribute 'fraction' must be in the range (0, 1].")
        if ratio < 1.0:
            raise ValueError("Attribute 'ratio' must be >= 1.")
        if min_samples < 1:
            raise ValueError("Attribute 'min_samples' must be >= 1.")
        if cutoff < 0 or cutoff > 1.0:
            raise ValueError("Attribute 'cutoff' must be in the range [0, 1].")
        self.pipeline, self.fraction, self.ratio, self.min_samples, self.cutoff = pipeline, float(fraction), float(ratio), int(min_samples), float(cutoff)
        if mask is not None and np.asarray(mask).ndim != 2:
            raise ValueError("Mask must be a 2D array.")
        self.mask = None if mask is None else np.asarray(mask).astype(bool)

    def generate_tasks(self, pixels):
        pixels = tuple(pixels)
        if self.mask is None:
            self.mask = np.ones(pixels, dtype=bool)
        if pixels != self.mask.shape:
            if self.mask.all():
                self.mask = np.ones(pixels, dtype=bool)
            else:
                raise ValueError("The pixel geometry passed to the frame sampler is inconsistent with the mask shape.")
        frame = self.pipeline.xyz_frame
        if frame is None:
            return self._full_frame(pixels)
        if (pixels[0], pixels[1], 3) != frame.shape:
            raise ValueError("The number of pixels passed to the frame sampler are inconsistent with the pipeline frame size.")
        mask = self.mask
        samples, mean = frame.samples, frame.mean
        min_samples = max(self.min_samples, int(samples[mask].max() / self.ratio))
        error = frame.errors()
        with np.errstate(divide="ignore", invalid="ignore"):
            per_channel = np.where(mean > 0, error / mean, 0.0)
        normalised = np.where(mask, per_channel.max(axis=2), 0.0)
        percentile_error = np.percentile(normalised[mask], (1 - self.fraction) * 100)
        cutoff = max(self.cutoff, percentile_error)
        need = mask & ((samples.min(axis=2) < min_samples) | (normalised > cutoff))
        xs, ys = np.nonzero(need)                           # x outer, y inner
        tasks = list(zip(xs.tolist(), ys.tolist()))
        pyrandom.shuffle(tasks)
        return tasks

    def _full_frame(self, pixels):
        nx, ny = pixels
        tasks = [(x, y) for x in range(nx) for y in range(ny) if self.mask[x, y]]
        pyrandom.shuffle(tasks)
        return tasks


# ---------------------------------------------------------------------------------------------------
# observers
# ---------------------------------------------------------------------------------------------------
class _ObserverBase(Observer):
    """observer/base/observer.pyx:45-511"""

    def __init__(self, parent=None, transform=None, name=None, render_engine=None, spectral_rays=None, spectral_bins=None,
                 min_wavelength=None, max_wavelength=None, ray_extinction_prob=None, ray_extinction_min_depth=None,
                 ray_max_depth=None, ray_importance_sampling=None, ray_important_path_weight=None, quiet=None):
        super().__init__(parent, transform, name)
        self.render_engine = render_engine or HipEngine()       # reference default: MulticoreEngine()
        # `x or default`, as the reference
        self._min_wavelength = min_wavelength or 375.0
        self._max_wavelength = max_wavelength or 740.0
        self.spectral_bins = spectral_bins or 15
        self.spectral_rays = spectral_rays or 1
        self.ray_extinction_prob = ray_extinction_prob or 0.01
        self.ray_extinction_min_depth = ray_extinction_min_depth or 3
        self.ray_max_depth = ray_max_depth or 500
        self.ray_importance_sampling = ray_importance_sampling or True
        self.ray_important_path_weight = ray_important_path_weight or 0.2
        self.quiet = quiet or False
        self.render_complete = False
        self.stats = {}
        self._auto_key, self._auto_offset = None, 0         # Philox counters handed out by earlier passes (see pass_sample_offset)

    @property
    def min_wavelength(self):
        return self._min_wavelength

    @min_wavelength.setter
    def min_wavelength(self, value):
        if value <= 0:
            raise ValueError("The minimum wavelength must be greater than 0.")
        if value >= self._max_wavelength:
            raise ValueError("The minimum wavelength must be less than the maximum wavelength.")
        self._min_wavelength = value

    @property
    def max_wavelength(self):
        return self._max_wavelength

    @max_wavelength.setter
    def max_wavelength(self, value):
        if value <= 0:
            raise ValueError("The maximum wavelength must be greater than 0.")
        if self._min_wavelength >= value:
            raise ValueError("The maximum wavelength must be greater than the minimum wavelength.")
        self._max_wavelength = value

    def observe(self):                                      # observer.pyx:265-309
        self.render_complete = False
        if not isinstance(self.root, World):
            raise TypeError("Observer is not connected to a scene graph containing a World object.")
        slices = self._slice_spectrum()
        templates = self._generate_templates(slices)
        self._initialise_pipelines(self._min_wavelength, self._max_wavelength, self.spectral_bins, slices, self.quiet)
        tasks = self._generate_tasks()
        if not len(tasks):
            self.render_complete = True
            return
        self._slices = slices
        self.stats = {"rays": 0, "kernel_ms": 0.0}
        self._check_counter_layout(self.render_engine)
        self._pass_offset = self.pass_sample_offset(self.render_engine)
        self._initialise_statistics(tasks)
        # (an observer that reports — quiet False — renders its pass now: the closing line states this pass's time and rays)
        if self.quiet and self._lazy_pass(tasks, templates):  # a small pass: accepted, submitted with its successors (HipEngine.auto_batch)
            self._auto_offset += self._samples_per_pass()
            self._finalise_pipelines()
            return
        # The slices fill disjoint bins of the frames, so on the device their passes need not wait for one another: a HipEngine
        # render of several slices defers the end-of-pass checks of path-traced scenes (librsx: rsx_defer_path_checks) and the
        # tail of one slice — a few paths bouncing on for hundreds of segments — drains under the bulk of the next.
        deferring = self._begin_deferred_slices(len(templates))
        try:
            # (slice sharding, SURVEY.md 8e: an engine with a slice_range renders those spectral slices only — another GPU the rest)
            first, last = getattr(self.render_engine, "slice_range", None) or (0, len(templates))
            for slice_id, template in enumerate(templates):
                if not first <= slice_id < last:
                    continue
                self.render_engine.run(tasks, self._render_pixel, self._update_state,
                                       render_args=(slice_id, template), update_args=(slice_id,))
                self._update_statistics(len(tasks))
        finally:
            if deferring:
                self._end_deferred_slices()
        # (sample_stride: a process that renders 1 / N of a sample-sharded job leaves the counters between its passes to the others)
        self._auto_offset += (self._samples_per_pass() * max(1, int(getattr(self.render_engine, "sample_stride", 1)))
                              * max(1, int(getattr(self.render_engine, "passes_per_call", 1))))
        self._finalise_pipelines()                           # render_complete stays False: only a pass without tasks completes a render
        self._finalise_statistics()

    # The observer's own report (observer.pyx:463-511): the reference counts a task per pixel and spectral ray as its workers hand them
    # back and prints a progress line at most once a second and a closing line. Here a slice comes back whole, so progress moves a slice
    # at a time; the closing line waits for the device (passes are asynchronous) so that its time is the render's and its rays are all
    # of them. `quiet` switches both off, as there.
    def _initialise_statistics(self, tasks):
        if self.quiet:
            return
        import time
        self._stats_ray_count = 0
        self._stats_start_time = self._stats_progress_timer = time.time()
        self._stats_total_tasks = len(tasks) * self.spectral_rays
        self._stats_completed_tasks = 0

    def _update_statistics(self, n_tasks):
        if self.quiet:
            return
        import time
        self._stats_completed_tasks += n_tasks
        if (time.time() - self._stats_progress_timer) > 1.0:
            rays = self.stats.get("rays", 0)
            print("Render time: {:0.3f}s ({:0.2f}% complete, {:0.1f}k rays)".format(
                time.time() - self._stats_start_time, 100 * (self._stats_completed_tasks / self._stats_total_tasks), (rays - self._stats_ray_count) / 1000))
            self._stats_ray_count = rays
            self._stats_progress_timer = time.time()

    def _finalise_statistics(self):
        if self.quiet:
            return
        import time
        from ..device import get_context
        try:
            get_context().synchronize()
        except Exception:
            pass
        elapsed_time = max(time.time() - self._stats_start_time, 1e-9)
        print("Render complete - time elapsed {:0.3f}s - {:0.1f}k rays/s".format(elapsed_time, self.stats.get("rays", 0) / elapsed_time / 1000))

    def _lazy_pass(self, tasks, templates):
        """Observers that can batch small passes override this (PinholeCamera); False: render the pass now."""
        return False

    def _flush_lazy(self):
        pass

    def _needs_host_materials(self, world, engine):
        """True when the slice must go through the host-callback path: a material without a device lowering (a user-written
        Material subclass), or an engine that asks for it."""
        if getattr(engine, "host_materials", False):
            return True
        from .material import has_device_lowering
        return not all(has_device_lowering(p.material) for p in world._primitives)

    def _begin_deferred_slices(self, n_slices):
        engine = self.render_engine
        self._deferred = None
        if n_slices < 2 or not isinstance(engine, HipEngine) or not engine.fused or engine.timing or engine.rng != "philox":
            return False
        if any(isinstance(p, RGBPipeline2D) for p in self._pipelines) or self._needs_host_materials(self.root, engine):
            return False
        from ..device import get_context
        self._deferred = []                                 # one re-issue closure per deferred library call, in call order
        get_context().defer_path_checks(True)
        return True

    def _end_deferred_slices(self):
        from ..device import get_context
        ctx = get_context()
        calls, self._deferred = self._deferred, None
        try:
            failed, rays = ctx.collect_path_checks(max(16, len(calls)))
        finally:
            ctx.defer_path_checks(False)
        self.stats["rays"] = self.stats.get("rays", 0) + rays
        for k in failed:                                    # (term arena ran out, too many volumes at a point: the ordinary retry path)
            calls[k]()

    @staticmethod
    def _check_counter_layout(engine):
        """sample_stride = N interleaves the ranks' passes (rank r, pass p: counters (p * N + r) * pixel_samples ...); a call of K passes
        draws K * pixel_samples consecutive counters and would run into the next rank's. The two do not compose: a sample-sharded
        process that wants K passes per call places every call itself (engine.sample_offset =
        distributed.rank_sample_offset(call, rank, N, K * pixel_samples), sample_stride left at 1)."""
        if int(getattr(engine, "sample_stride", 1)) > 1 and int(getattr(engine, "passes_per_call", 1)) > 1:
            raise ValueError("sample_stride > 1 with passes_per_call > 1: the K passes of one call would reuse the Philox counters of the next "
                             "rank's pass; assign engine.sample_offset per call (distributed.rank_sample_offset with spp = K * pixel_samples) instead")

    def pass_sample_offset(self, engine):
        """First Philox sample counter of the pass about to be rendered: the engine's sample_offset plus the samples this observer's
        earlier passes drew with that same setting. The reference's engines draw fresh random numbers every pass (one running
        MT19937-64 stream, core/math/random.pyx); with counter-based numbers that means: never hand out a counter twice. Assigning
        engine.sample_offset (or another engine) restarts the count at the assigned value."""
        key = (id(engine), getattr(engine, "sample_offset", 0), getattr(engine, "seed", 0))
        if key != self._auto_key:
            self._auto_key, self._auto_offset = key, 0
        return getattr(engine, "sample_offset", 0) + self._auto_offset

    def _samples_per_pass(self):
        return 0

    def _slice_spectrum(self):                              # observer.pyx:311-340
        current, start, ranges = 0, 0, []
        while start < self.spectral_bins:
            current += self.spectral_bins / self.spectral_rays
            end = round(current)
            ranges.append((start, end))
            start = end
        return [SpectralSlice(self._min_wavelength, self._max_wavelength, self.spectral_bins, end - start, start) for start, end in ranges]

    def _generate_templates(self, slices):                  # observer.pyx:342-355
        return [Ray(min_wavelength=s.min_wavelength, max_wavelength=s.max_wavelength, bins=s.bins,
                    extinction_prob=self.ray_extinction_prob, extinction_min_depth=self.ray_extinction_min_depth,
                    max_depth=self.ray_max_depth, importance_sampling=self.ray_importance_sampling,
                    important_path_weight=self.ray_important_path_weight) for s in slices]

    def _render_pixel(self, task, slice_id, template):
        raise RuntimeError("source_amd observers render whole slices on the device (HipEngine); the per-pixel CPU "
                           "worker of the reference (observer.pyx:363-419) is intentionally not provided.")

    def _update_state(self, packed_result, slice_id):       # observer.pyx:425-446
        task, results, ray_count = packed_result
        self._update_pipelines(task, results, slice_id)
        self.stats["rays"] = self.stats.get("rays", 0) + ray_count


class Observer2D(_ObserverBase):
    """observer/base/observer.pyx:896-1079"""

    def __init__(self, pixels, frame_sampler, pipelines, parent=None, transform=None, name=None, render_engine=None,
                 pixel_samples=None, **kw):
        self.pixel_samples = pixel_samples or 100
        self.pixels = pixels
        self.frame_sampler = frame_sampler
        self.pipelines = pipelines
        super().__init__(parent, transform, name, render_engine, **kw)

    @property
    def pixel_samples(self):
        return self._pixel_samples

    @pixel_samples.setter
    def pixel_samples(self, value):
        if value <= 0:
            raise ValueError("The number of pixel samples must be greater than 0.")
        self._pixel_samples = int(value)

    @property
    def pixels(self):
        return self._pixels

    @pixels.setter
    def pixels(self, value):
        pixels = tuple(value)
        if len(pixels) != 2:
            raise ValueError("Pixels must be a 2 element tuple defining the x and y resolution.")
        if pixels[0] <= 0:
            raise ValueError("Number of x pixels must be greater than 0.")
        if pixels[1] <= 0:
            raise ValueError("Number of y pixels must be greater than 0.")
        self._pixels = pixels

    @property
    def frame_sampler(self):
        return self._frame_sampler

    @frame_sampler.setter
    def frame_sampler(self, value):
        if not isinstance(value, FrameSampler2D):
            raise TypeError("The frame sampler for a 2d observer must be a subclass of FrameSampler2D.")
        self._frame_sampler = value

    @property
    def pipelines(self):
        return self._pipelines

    @pipelines.setter
    def pipelines(self, value):
        pipelines = tuple(value)
        if len(pipelines) < 1:
            raise ValueError("At least one processing pipeline must be provided.")
        for p in pipelines:
            if not isinstance(p, Pipeline2D):
                raise TypeError("Processing pipelines for a 2d observer must be a subclass of Pipeline2D.")
        self._pipelines = pipelines

    def _generate_tasks(self):
        return self._frame_sampler.generate_tasks(self._pixels)

    def _samples_per_pass(self):
        return self._pixel_samples

    def _initialise_pipelines(self, min_wavelength, max_wavelength, spectral_bins, slices, quiet):
        for p in self._pipelines:
            p.initialise(self._pixels, self._pixel_samples, min_wavelength, max_wavelength, spectral_bins, slices, quiet)

    def _update_pipelines(self, task, results, slice_id):
        x, y = task
        for result, p in zip(results, self._pipelines):
            p.update(x, y, slice_id, result)

    def _finalise_pipelines(self):
        for p in self._pipelines:
            p.finalise()


class PinholeCamera(Observer2D):
    """optical/observer/imaging/pinhole.pyx:42-207"""

    MAX_RAYS_PER_CALL = 1 << 29       # rays per librsx render call (12.9 GB of sample records); larger slices are cut, see _render_slice_device

    def __init__(self, pixels, fov=None, sensitivity=None, frame_sampler=None, pipelines=None, parent=None, transform=None, name=None):
        if not pipelines and not frame_sampler:                 # pinhole.pyx:76-83: an adaptively sampled RGB pipeline by default
            rgb = RGBPipeline2D()
            pipelines = [rgb]
            frame_sampler = RGBAdaptiveSampler2D(rgb)
        else:
            pipelines = pipelines or [RGBPipeline2D()]
            frame_sampler = frame_sampler or FullFrameSampler2D()
        self._fov = 45
        super().__init__(pixels, frame_sampler, pipelines, parent=parent, transform=transform, name=name)
        self.fov = fov or 45
        self.sensitivity = sensitivity or 1.0

    @property
    def fov(self):
        return self._fov

    @fov.setter
    def fov(self, value):
        if value <= 0 or value >= 180:
            raise ValueError("The field-of-view angle must lie in the range (0, 180).")
        self._fov = value
        self._update_image_geometry()

    @property
    def pixels(self):
        return self._pixels

    @pixels.setter
    def pixels(self, value):
        Observer2D.pixels.fset(self, value)
        self._update_image_geometry()

    @property
    def sensitivity(self):
        return self._sensitivity

    @sensitivity.setter
    def sensitivity(self, value):
        if value <= 0:
            raise ValueError("Sensitivity must be greater than zero.")
        self._sensitivity = value

    def _update_image_geometry(self):                       # pinhole.pyx:148-167
        max_pixels = max(self._pixels)
        if max_pixels > 1:
            image_max_width = 2 * math.tan(math.pi / 180 * 0.5 * self._fov)
            self.image_delta = image_max_width / max_pixels
            self.image_start_x = 0.5 * self._pixels[0] * self.image_delta
            self.image_start_y = 0.5 * self._pixels[1] * self.image_delta
        else:
            raise RuntimeError("Number of Pinhole camera Pixels must be > 1.")

    def _pixel_sensitivity(self, x, y):
        return self._sensitivity

    def device_camera(self):
        cam = _lib.Camera()
        cam.nx, cam.ny = self._pixels
        cam.image_delta, cam.image_start_x, cam.image_start_y = self.image_delta, self.image_start_x, self.image_start_y
        for i, v in enumerate(self.to_root().m):
            cam.to_root[i] = v
        cam.sensitivity = float(self._sensitivity)
        return cam

    # -- small passes, batched (HipEngine.auto_batch) -----------------------------------------------
    def _lazy_signature(self, tasks, engine, world):
        """What must stay the same for the next observe() to be one more pass of the pending call — or None when this pass has to be
        rendered by itself (anything the K-passes-per-call form of librsx does not cover)."""
        if not isinstance(engine, HipEngine) or not getattr(engine, "auto_batch", False):
            return None
        if (not engine.fused or engine.rng != "philox" or engine.timing or engine.host_materials or engine.passes_per_call != 1
                or engine.sample_stride != 1):
            return None
        spp = self._pixel_samples
        if spp >= 64 or 64 % spp:
            return None
        for pipe in self._pipelines:
            if isinstance(pipe, RGBPipeline2D) or not isinstance(pipe, SpectralPowerPipeline2D) or not pipe.accumulate or pipe.frame is None:
                return None
        if not isinstance(tasks, RectTasks):
            tasks = self._coherent_tasks(tasks)
            if not isinstance(tasks, RectTasks):
                return None                                 # (an adaptive sampler's pick: it reads the frame every pass anyway)
        from .material import NullSurface, Lambert, Dielectric, UniformVolumeEmitter
        if self._needs_host_materials(world, engine) or any(isinstance(p.material, (NullSurface, Lambert, Dielectric, UniformVolumeEmitter)) for p in world._primitives):
            return None                                     # (path passes may have to be rendered again by themselves)
        return (id(engine), engine.seed, engine.slice_range, id(world), tuple(tasks.rect), self._pixels, spp, self.spectral_bins, self.spectral_rays,
                self._min_wavelength, self._max_wavelength, self._fov, self._sensitivity, tuple(self.to_root().m),
                tuple((id(pipe), id(pipe.frame), pipe.power) for pipe in self._pipelines))

    def _lazy_pass(self, tasks, templates):
        engine, world = self.render_engine, self.root
        sig = self._lazy_signature(tasks, engine, world)
        pend = getattr(self, "_lazy", None)
        offset, spp = self._pass_offset, self._pixel_samples
        if pend is not None and (sig is None or pend["sig"] != sig or offset != pend["first"] + pend["count"] * spp):
            self._flush_lazy()
            pend = None
        if sig is None:
            return False
        if pend is not None:
            # material parameters changed in place (no scenegraph notification: `emitter.scale = 2`) since the batch began? The pending
            # calls hold the materials and spectral tables of their own moment; a pass that would send other bytes starts a new batch
            # (every slice the batch renders: a spectral function edited only at the wavelengths of a later slice must be seen too)
            if self._all_material_bytes(world, engine, len(templates)) != pend["materials"]:
                self._flush_lazy()
                pend = None
        if pend is None:
            # the library calls of this pass, built now — materials, tables, camera, Philox keys are those of THIS moment — and kept;
            # later passes of the batch only raise their `passes` count
            if not isinstance(tasks, RectTasks):
                tasks = self._coherent_tasks(tasks)
            scene = world.build_accelerator()
            first, last = getattr(engine, "slice_range", None) or (0, len(templates))
            limit = self.MAX_RAYS_PER_CALL // (64 // spp)          # a full batch is 64 / spp passes of spp samples
            calls = []
            for slice_id in range(len(templates)):
                if not first <= slice_id < last:
                    continue
                sl = self._slices[slice_id]
                for piece in self._pieces(tasks, world, limit * max(1, int(getattr(engine, "passes_per_call", 1)))):
                    for pipe in self._pipelines:
                        keep = []
                        desc = self.render_desc(world, None, sl, engine, keep, rect=piece["rect"], sample_offset=offset)
                        desc.power = 1 if pipe.power else 0
                        calls.append(dict(desc=desc, keep=keep, frame=pipe.frame, offset=sl.offset))
            pend = self._lazy = dict(sig=sig, first=offset, count=0, calls=calls, scene=scene, rays=sum(c["desc"].n_tasks for c in calls) * spp // max(1, len(self._pipelines)),
                                     materials=self._all_material_bytes(world, engine, len(templates)))
            for pipe in self._pipelines:
                owners = getattr(pipe.frame, "_lazy_owners", None)
                if owners is None:
                    owners = pipe.frame._lazy_owners = weakref.WeakSet()
                owners.add(self)
            world._lazy_observers.add(self)
            from .. import device
            device.pending_observers.add(self)
        pend["count"] += 1
        self.stats = {"rays": pend["rays"], "kernel_ms": 0.0}
        # A full batch goes out at once; a partial one goes out the moment the device has nothing to do (the first passes of a loop, a
        # loop of fewer passes than a batch holds: 20 batched passes used to be a 16-pass launch after sixteen observe() calls of host
        # time and a 4-pass launch at the read — three times their kernel time). While the device is busy the batch keeps growing.
        # (only power-of-two batches of at least eager_min samples per pixel: anything smaller is cut into launches of 1, 2, 4 passes that
        # cost 0.4 ms per pass where a 16-pass launch takes 0.1 — profiles/r05a_c2_kernel_stats.csv)
        count = pend["count"]
        if count * spp >= 64 or (getattr(engine, "eager_batch", True) and count * spp >= getattr(engine, "eager_min", 16) and count & (count - 1) == 0 and pend["scene"].context.idle()):
            self._flush_lazy()
        return True

    def _all_material_bytes(self, world, engine, n_slices):
        """The material records and spectral tables of every slice the engine renders, as one byte string (the batch's staleness probe)."""
        first, last = getattr(engine, "slice_range", None) or (0, n_slices)
        parts = []
        for slice_id in range(first, last):
            # (what render_desc sends of the materials, without the rest of a render description — camera, important spheres with their
            # bounding boxes, Philox keys: this probe runs in every observe() of a lazy batch)
            sl = self._slices[slice_id]
            tables = []
            mats = [p.material.device_material(tables, sl.min_wavelength, sl.max_wavelength, sl.bins) for p in world._primitives]
            parts.append(bytes((_lib.Material * max(1, len(mats)))(*mats)))
            if tables:
                parts.append(np.ascontiguousarray(np.array(tables, dtype=np.float64).reshape(len(tables), sl.bins)).tobytes())
        return b"".join(parts)

    @staticmethod
    def _material_bytes(desc, keep):
        """The material records and spectral tables a render call sends (render_desc puts both at the head of `keep`)."""
        mat_arr, tab = keep[0], keep[1]
        return bytes(mat_arr) + tab.tobytes()

    def _flush_lazy(self):
        """Submits the pending passes: one library call per (slice, block of pixels, pipeline) with passes = the number accepted."""
        pend = getattr(self, "_lazy", None)
        if pend is None:
            return
        self._lazy = None
        from .. import device
        device.pending_observers.discard(self)
        L = _lib.lib()
        scene = pend["scene"]
        for call in pend["calls"]:
            desc, frame = call["desc"], call["frame"]
            desc.passes = pend["count"]
            fm, fv, fn = frame._device(scene.context, settle=False)
            rays = C.c_uint64(0)
            _lib.check(L.rsx_render_pinhole_frame(scene.handle, C.byref(desc), fm, fv, fn, frame.nz, call["offset"], C.byref(rays)))
            LIBRARY_CALLS[0] += 1
            frame._mark_device_written()

    # -- the device path --------------------------------------------------------------------------
    def render_desc(self, world, tasks, slice_, engine, keep, rect=None, sample_offset=None):
        """Builds the rsx_render_desc for one spectral slice. ``keep`` collects arrays that must outlive the call.
        ``sample_offset``: first Philox sample counter (default: the engine's; observe() passes pass_sample_offset())."""
        tables = []
        mats = [p.material.device_material(tables, slice_.min_wavelength, slice_.max_wavelength, slice_.bins) for p in world._primitives]
        desc = _lib.RenderDesc()
        desc.camera = self.device_camera()
        mat_arr = (_lib.Material * max(1, len(mats)))(*mats)
        tab = np.ascontiguousarray(np.array(tables, dtype=np.float64).reshape(len(tables), slice_.bins)) if tables else np.zeros((0, slice_.bins))
        keep.extend([mat_arr, tab])
        desc.materials, desc.n_materials = mat_arr, len(mats)
        desc.tables, desc.n_tables = _lib.ptr(tab) if len(tables) else None, len(tables)
        desc.bins, desc.spp = slice_.bins, self._pixel_samples
        desc.ray_max_depth, desc.ray_extinction_min_depth = int(self.ray_max_depth), int(self.ray_extinction_min_depth)
        desc.ray_extinction_prob = float(self.ray_extinction_prob)
        # multiple importance sampling (material.pyx:327: ray.importance_sampling and world.has_important_primitives())
        spheres = world.important_spheres() if self.ray_importance_sampling else []
        desc.n_important, desc.important_path_weight = len(spheres), float(self.ray_important_path_weight)
        if spheres:
            arr = (_lib.ImportantSphere * len(spheres))()
            for rec, (centre, radius, cdf, weight) in zip(arr, spheres):
                rec.centre[0], rec.centre[1], rec.centre[2] = centre
                rec.radius, rec.cdf, rec.weight = radius, cdf, weight
            keep.append(arr)
            desc.important = arr
        if rect is not None:
            desc.tasks = None
            for i in range(4):
                desc.rect[i] = rect[i]
            desc.n_tasks = (rect[2] - rect[0]) * (rect[3] - rect[1])
        else:
            t = np.ascontiguousarray(np.array(tasks, dtype=np.int32).reshape(-1, 2))
            keep.append(t)
            desc.tasks, desc.n_tasks = _lib.ptr(t), len(t)
        if engine.rng == "stream":
            if rect is not None:
                raise ValueError("rng='stream' needs an explicit task list (the MT stream is consumed in task order)")
            u = rsrandom.uniform_block(2 * desc.n_tasks * self._pixel_samples)
            keep.append(u)
            desc.uniforms, desc.rng_mode = _lib.ptr(u), _lib.RNG_STREAM
        else:
            # one independent Philox stream per spectral slice (the reference draws fresh jitter for every slice of a pixel,
            # observer.pyx:299-305): the slice offset is folded into the key; slice 0 keeps the engine's seed
            key = (engine.seed + slice_.offset * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
            desc.uniforms, desc.rng_mode, desc.seed = None, _lib.RNG_PHILOX, key
            desc.sample_offset = engine.sample_offset if sample_offset is None else int(sample_offset)
        return desc

    def _render_xyz(self, scene, desc, pipe, slice_id, tasks, rect, rays):
        """One slice of one block of pixels for an RGBPipeline2D: XYZPixelProcessor results from the device, summed into the
        pipeline's working frames (RGBPipeline2D.update)."""
        n = desc.n_tasks
        mean, var = np.zeros((n, 3)), np.zeros((n, 3))
        _lib.check(_lib.lib().rsx_render_pinhole_xyz(scene.handle, C.byref(desc), _lib.ptr(pipe._resampled[slice_id]), pipe._deltas[slice_id],
                                                     _lib.ptr(mean), _lib.ptr(var), C.byref(rays)))
        if rect is not None:                                # row-major over the rect: k = ly * w + lx
            x0, y0, x1, y1 = rect
            xs, ys = np.meshgrid(np.arange(x0, x1), np.arange(y0, y1), indexing="xy")
            xs, ys = xs.reshape(-1), ys.reshape(-1)
        else:
            t = np.array(tasks, dtype=np.int64).reshape(-1, 2)
            xs, ys = t[:, 0], t[:, 1]
        pipe.update_block(xs, ys, mean, var)

    HOST_RAYS_PER_PIECE = 1 << 16     # primary rays per scheduler run of the host-callback path (bounds the Python objects alive at once)

    def _pieces(self, tasks, world, limit=None):
        """One library call renders at most MAX_RAYS_PER_CALL rays (its sample-record buffer is 24 B per ray): larger slices go
        band by band (rect) or run by run (task list, which keeps the MT stream's consumption order). Pixels are independent
        and the Philox counters are per (pixel, sample), so the frame does not depend on how a slice is cut."""
        if limit is None:
            limit = self.MAX_RAYS_PER_CALL
        from .material import NullSurface, Lambert, Dielectric
        if any(isinstance(p.material, (NullSurface, Lambert, Dielectric)) for p in world._primitives):
            limit = min(limit, 1 << 24)                     # the volume path also keeps 768 B of emission terms per ray
        per_call = max(1, limit // (self._pixel_samples * max(1, int(getattr(self.render_engine, "passes_per_call", 1)))))
        if isinstance(tasks, RectTasks):
            x0, y0, x1, y1 = tasks.rect
            band = max(1, per_call // max(1, y1 - y0))
            return [dict(rect=(xa, y0, min(xa + band, x1), y1)) for xa in range(x0, x1, band)]
        return [dict(tasks=tasks[a:a + per_call]) for a in range(0, len(tasks), per_call)]

    def _coherent_tasks(self, tasks):
        """Philox samples are keyed by (pixel, sample), so the order in which a pass visits its pixels never shows in the frames: a task
        list — FullFrameSampler2D shuffles its pixels like the reference's (sampler2d.pyx:42-102), an adaptive sampler picks them — is
        rendered as a rectangle when it is one (every pixel of its bounding box, once), else in 8 x 8 tile order, so that the 64 rays of
        a unit lie side by side instead of all over the frame. (The MT stream of rng="stream" is consumed in task order: untouched.)
        One conversion per observe(): the spectral slices of a pass share the list."""
        cached = getattr(self, "_coherent_cache", None)
        if cached is not None and cached[0] is tasks:
            return cached[1]
        t = np.asarray(tasks, dtype=np.int64).reshape(-1, 2)
        out = tasks
        if len(t):
            x0, y0 = (int(v) for v in t.min(axis=0))
            x1, y1 = (int(v) + 1 for v in t.max(axis=0))
            if len(t) == (x1 - x0) * (y1 - y0) and len(np.unique(t[:, 0] * (y1 + 1) + t[:, 1])) == len(t):
                out = RectTasks(x0, y0, x1, y1)
            else:
                order = np.lexsort((t[:, 0] & 7, t[:, 1] & 7, t[:, 0] >> 3, t[:, 1] >> 3))
                out = np.ascontiguousarray(t[order].astype(np.int32))
        self._coherent_cache = (tasks, out)
        return out

    def _render_slice_device(self, tasks, slice_id, template, engine, update, update_args, update_kwargs):
        world = self.root
        self._flush_lazy()                                  # (passes accepted earlier come first)
        if self._needs_host_materials(world, engine):
            if int(getattr(engine, "passes_per_call", 1)) > 1:
                raise ValueError("passes_per_call > 1 needs device lowerings for every material (host-evaluated materials render pass by pass)")
            from . import hybrid
            hybrid.render_slice(self, tasks, slice_id, template, engine, self._pieces(tasks, world, self.HOST_RAYS_PER_PIECE))
            return
        scene = world.build_accelerator()
        sl = self._slices[slice_id]
        keep = []
        L = _lib.lib()
        rays = C.c_uint64(0)
        offset = getattr(self, "_pass_offset", None)
        if engine.fused and engine.rng == "philox" and not isinstance(tasks, RectTasks):
            tasks = self._coherent_tasks(tasks)
        passes = max(1, int(getattr(engine, "passes_per_call", 1)))
        if passes > 1:
            if not engine.fused or engine.rng != "philox":
                raise ValueError("passes_per_call > 1 needs the fused Philox path (HipEngine(rng='philox', fused=True))")
            if any(isinstance(pipe, RGBPipeline2D) or not getattr(pipe, "accumulate", False) for pipe in self._pipelines):
                raise ValueError("passes_per_call > 1 is for accumulating spectral pipelines (accumulate=True): the passes of one call are merged into the frame")
        for piece in self._pieces(tasks, world):
            desc = self.render_desc(world, piece.get("tasks"), sl, engine, keep, rect=piece.get("rect"), sample_offset=offset)
            desc.passes = passes
            if engine.fused:
                for pipe in self._pipelines:
                    if isinstance(pipe, RGBPipeline2D):
                        self._render_xyz(scene, desc, pipe, slice_id, piece.get("tasks"), piece.get("rect"), rays)
                        continue
                    desc.power = 1 if pipe.power else 0
                    fm, fv, fn = pipe.frame._device(scene.context)
                    _lib.check(L.rsx_render_pinhole_frame(scene.handle, C.byref(desc), fm, fv, fn, pipe.frame.nz, sl.offset, C.byref(rays)))
                    LIBRARY_CALLS[0] += 1
                    pipe.frame._mark_device_written()
                    if rays.value == 0xFFFFFFFFFFFFFFFF:        # a deferred path pass: counted, and if need be issued again, at the end of observe()
                        rays.value = 0
                        def again(desc=desc, keep=keep, power=desc.power, fm=fm, fv=fv, fn=fn, nz=pipe.frame.nz, offset=sl.offset, handle=scene.handle):
                            desc.power = power
                            count = C.c_uint64(0)
                            _lib.check(L.rsx_render_pinhole_frame(handle, C.byref(desc), fm, fv, fn, nz, offset, C.byref(count)))
                            self.stats["rays"] = self.stats.get("rays", 0) + count.value
                        self._deferred.append(again)
                    if engine.timing:
                        tr, ac = scene.context.last_render_ms()
                        engine.last_kernel_ms = tr
                        self.stats["kernel_ms"] = self.stats.get("kernel_ms", 0.0) + tr
                        self.stats["accumulate_ms"] = self.stats.get("accumulate_ms", 0.0) + ac
                self.stats["rays"] = self.stats.get("rays", 0) + rays.value
                continue
            # RenderEngine contract taken literally (workflow.py:78-91): per-task (mean, variance) blocks, update() once per task
            n = desc.n_tasks
            results = []
            for pipe in self._pipelines:
                if isinstance(pipe, RGBPipeline2D):
                    mean, var = np.zeros((n, 3)), np.zeros((n, 3))
                    _lib.check(L.rsx_render_pinhole_xyz(scene.handle, C.byref(desc), _lib.ptr(pipe._resampled[slice_id]), pipe._deltas[slice_id],
                                                        _lib.ptr(mean), _lib.ptr(var), C.byref(rays)))
                    results.append((mean, var))
                    continue
                desc.power = 1 if pipe.power else 0
                mean, var = np.zeros((n, sl.bins)), np.zeros((n, sl.bins))
                _lib.check(L.rsx_render_pinhole(scene.handle, C.byref(desc), _lib.ptr(mean), _lib.ptr(var), C.byref(rays)))
                results.append((mean, var))
            piece_tasks = piece["tasks"] if "tasks" in piece else RectTasks(*piece["rect"])    # rect blocks come back row-major, like RectTasks iterates
            ray_share, ray_rest = divmod(int(rays.value), max(1, n))
            for k, task in enumerate(piece_tasks):
                packed = (tuple(task), [(m[k], v[k]) for m, v in results], ray_share + (1 if k < ray_rest else 0))
                update(packed, *update_args, **update_kwargs)
