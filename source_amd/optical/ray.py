"""
Optical Ray: the geometric ray plus its spectral configuration (raysect/optical/ray.pyx:44-330). trace() keeps the
reference's control flow (ray.pyx:338-401) for single rays — world.hit() on the device, the material plugin on the
host — and is what a user-written material sees; bulk rendering never goes through it (see observer.py).
"""
from ..core.scenegraph import Ray as CoreRay
from .spectral import Spectrum


class Ray(CoreRay):
    __slots__ = ("min_wavelength", "max_wavelength", "bins", "extinction_prob", "extinction_min_depth", "max_depth",
                 "importance_sampling", "important_path_weight", "depth", "ray_count")

    def __init__(self, origin=None, direction=None, min_wavelength=375, max_wavelength=785, bins=40, max_distance=float("inf"),
                 extinction_prob=0.1, extinction_min_depth=3, max_depth=100, importance_sampling=True, important_path_weight=0.25):
        super().__init__(origin, direction, max_distance)
        if bins < 1:
            raise ValueError("Number of bins cannot be less than 1.")
        if min_wavelength <= 0.0 or max_wavelength <= 0.0 or min_wavelength >= max_wavelength:
            raise ValueError("Invalid wavelength range.")
        self.min_wavelength, self.max_wavelength, self.bins = float(min_wavelength), float(max_wavelength), int(bins)
        self.extinction_prob, self.extinction_min_depth, self.max_depth = extinction_prob, extinction_min_depth, max_depth
        self.importance_sampling, self.important_path_weight = importance_sampling, important_path_weight
        self.depth = 0
        self.ray_count = 0

    def new_spectrum(self):
        return Spectrum(self.min_wavelength, self.max_wavelength, self.bins)

    def copy(self, origin=None, direction=None):
        r = Ray(origin or self.origin.copy(), direction or self.direction.copy(), self.min_wavelength, self.max_wavelength, self.bins,
                self.max_distance, self.extinction_prob, self.extinction_min_depth, self.max_depth, self.importance_sampling,
                self.important_path_weight)
        return r

    def spawn_daughter(self, origin, direction):            # ray.pyx:506-545
        """A daughter ray: same spectral configuration, depth + 1."""
        r = Ray(origin, direction, self.min_wavelength, self.max_wavelength, self.bins, self.max_distance, self.extinction_prob,
                self.extinction_min_depth, self.max_depth, self.importance_sampling, self.important_path_weight)
        r.depth = self.depth + 1
        return r

    def trace(self, world, keep_alive=False):               # ray.pyx:338-401
        if self.depth == 0:
            self.ray_count = 1
        if not (keep_alive or self.depth < self.extinction_min_depth):
            # Russian roulette (ray.pyx:380-386) draws from the MT stream: only stochastic materials reach this depth, and they have
            # no lowering in this build
            raise NotImplementedError("Ray.trace(): Russian roulette at depth %d is part of the stochastic-material scope row" % self.depth)
        intersection = world.hit(self)
        if intersection is None:
            return self.new_spectrum()
        material = intersection.primitive.material
        spectrum = material.evaluate_surface(world, self, intersection.primitive, intersection.hit_point, intersection.exiting,
                                             intersection.inside_point, intersection.outside_point, intersection.normal,
                                             intersection.world_to_primitive, intersection.primitive_to_world, intersection)
        for primitive in world.contains(self.origin):       # _sample_volumes, ray.pyx:422-455
            spectrum = primitive.material.evaluate_volume(spectrum, world, self, primitive,
                                                          intersection.hit_point.transform(intersection.primitive_to_world), self.origin,
                                                          primitive.to_local(), primitive.to_root())
        return spectrum
