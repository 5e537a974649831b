"""
Optical Ray: the geometric ray plus its spectral configuration (raysect/optical/ray.pyx:44-330). trace() keeps the
reference's control flow (ray.pyx:338-455): Russian roulette, world.hit(), the material's evaluate_surface(), the volume pass over
world.contains(origin), the roulette normalisation. A single Ray.trace() runs world.hit()/contains() as one-ray device calls and the
material plugin on the host. Inside the host-callback render path (source_amd/optical/hybrid.py) the same method is what a
material's daughter.trace(world) reaches: there it hands the ray to the wave scheduler instead of tracing it alone.
Bulk rendering of device-lowered materials never goes through this class (see observer.py).
"""
from ..core import random as rsrandom
from ..core.scenegraph import Ray as CoreRay
from .spectral import Spectrum

_scheduler = None           # hybrid.WaveScheduler while a host-callback render is running


class Ray(CoreRay):
    __slots__ = ("min_wavelength", "max_wavelength", "bins", "extinction_prob", "extinction_min_depth", "max_depth",
                 "importance_sampling", "important_path_weight", "depth", "ray_count", "_primary_ray", "_node")

    def __init__(self, origin=None, direction=None, min_wavelength=375, max_wavelength=785, bins=40, max_distance=float("inf"),
                 extinction_prob=0.1, extinction_min_depth=3, max_depth=100, importance_sampling=True, important_path_weight=0.25):
        super().__init__(origin, direction, max_distance)
        if bins < 1:
            raise ValueError("Number of bins cannot be less than 1.")
        if min_wavelength <= 0.0 or max_wavelength <= 0.0 or min_wavelength >= max_wavelength:
            raise ValueError("Invalid wavelength range.")
        if extinction_prob < 0.0 or extinction_prob > 1.0:
            raise ValueError("The extinction probability must lie in the range [0, 1].")             # ray.pyx:262
        if extinction_min_depth < 1:
            raise ValueError("The minimum extinction depth cannot be less than 1.")                  # ray.pyx:276
        if max_depth < extinction_min_depth:
            raise ValueError("The maximum depth cannot be less than the minimum extinction depth.")  # ray.pyx:290
        if important_path_weight < 0.0 or important_path_weight > 1.0:
            raise ValueError("Important path weight must be in the range [0, 1].")                   # ray.pyx:107
        self.min_wavelength, self.max_wavelength, self.bins = float(min_wavelength), float(max_wavelength), int(bins)
        self.extinction_prob, self.extinction_min_depth, self.max_depth = extinction_prob, extinction_min_depth, max_depth
        self.importance_sampling, self.important_path_weight = importance_sampling, important_path_weight
        self.depth = 0
        self.ray_count = 0
        self._primary_ray = None
        self._node = None

    def new_spectrum(self):
        return Spectrum(self.min_wavelength, self.max_wavelength, self.bins)

    def copy(self, origin=None, direction=None):            # ray.pyx:548-585
        return Ray(origin or self.origin.copy(), direction or self.direction.copy(), self.min_wavelength, self.max_wavelength, self.bins,
                   self.max_distance, self.extinction_prob, self.extinction_min_depth, self.max_depth, self.importance_sampling,
                   self.important_path_weight)

    def spawn_daughter(self, origin, direction):            # ray.pyx:506-545
        """A daughter ray: same spectral configuration, depth + 1; the primary ray counts every daughter spawned (ray_count)."""
        r = Ray.__new__(Ray)
        r.origin, r.direction, r.max_distance = origin, direction, self.max_distance
        r.min_wavelength, r.max_wavelength, r.bins = self.min_wavelength, self.max_wavelength, self.bins
        r.extinction_prob, r.extinction_min_depth, r.max_depth = self.extinction_prob, self.extinction_min_depth, self.max_depth
        r.importance_sampling, r.important_path_weight = self.importance_sampling, self.important_path_weight
        r.depth = self.depth + 1
        r.ray_count = 0
        r._node = None
        primary = self if self._primary_ray is None else self._primary_ray
        primary.ray_count += 1
        r._primary_ray = primary
        return r

    def trace(self, world, keep_alive=False):               # ray.pyx:338-401
        if _scheduler is not None:
            return _scheduler.trace(self, world, keep_alive)        # a daughter spawned inside a host-callback render: traced with its wave
        if self._primary_ray is None:
            self.ray_count = 1
        # Russian roulette, with the normalisation that keeps the estimate unbiased (ray.pyx:380-388)
        if keep_alive or self.depth < self.extinction_min_depth:
            normalisation = 1.0
        else:
            if self.depth >= self.max_depth or rsrandom.probability(self.extinction_prob):
                return self.new_spectrum()
            normalisation = 1 / (1 - self.extinction_prob)
        intersection = world.hit(self)
        if intersection is None:
            return self.new_spectrum()
        spectrum = self._sample_surface(intersection, world)
        spectrum = self._sample_volumes(spectrum, intersection, world.contains(self.origin), world)
        spectrum.mul_scalar(normalisation)
        return spectrum

    def _sample_surface(self, intersection, world):         # ray.pyx:403-420
        material = intersection.primitive.material
        return material.evaluate_surface(world, self, intersection.primitive, intersection.hit_point, intersection.exiting,
                                         intersection.inside_point, intersection.outside_point, intersection.normal,
                                         intersection.world_to_primitive, intersection.primitive_to_world, intersection)

    def _sample_volumes(self, spectrum, intersection, primitives, world):    # ray.pyx:422-455
        if primitives:
            start_point = intersection.hit_point.transform(intersection.primitive_to_world)
            end_point = self.origin
            for primitive in primitives:
                spectrum = primitive.material.evaluate_volume(spectrum, world, self, primitive, start_point, end_point,
                                                              primitive.to_local(), primitive.to_root())
        return spectrum

    def sample(self, world, count):                         # ray.pyx:457-503
        if count < 1:
            raise ValueError("Samples must be >= 1.")
        spectrum = self.new_spectrum()
        normalisation = 1 / float(count)
        while count:
            spectrum.mad_scalar(normalisation, self.trace(world).samples)
            count -= 1
        return spectrum
