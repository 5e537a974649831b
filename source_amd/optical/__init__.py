"""source_amd.optical — mirrors the hot-path names of raysect.optical."""
from ..core import (AffineMatrix3D, Normal3D, Point3D, Vector3D, rotate, rotate_vector, rotate_x, rotate_y, rotate_z, translate,
                    Node, Primitive, Intersection, BoundingBox3D)
from ..core.scenegraph import World as _CoreWorld
from .spectral import ConstantSF, InterpolatedSF, SpectralFunction, Spectrum
from .ray import Ray
from . import material
from .material import AbsorbingSurface, Light, Material, NullVolume, UniformSurfaceEmitter


class World(_CoreWorld):
    """raysect/optical/scenegraph/world.pyx — the optical world. important_spheres() is the host half of its ImportanceManager
    (world.pyx:47-128): bounding spheres, cumulative selection probabilities and selection weights of the primitives whose material
    has importance > 0; the sampling itself (world.pyx:150-230) runs on the device."""

    def important_spheres(self):
        import math
        import numpy as np
        from ..primitive import Sphere
        spheres, total = [], 0
        for p in self._primitives:                                          # ImportanceManager._process_primitives
            importance = getattr(p.material, "importance", 0.0)
            if importance > 0:
                if isinstance(p, Sphere):                                   # sphere.pyx:232-234
                    c = Point3D(0, 0, 0).transform(p.to_root())
                    centre, radius = (c.x, c.y, c.z), p.radius * 1.000000001
                else:                                                       # primitive.pyx:166-186, boundingbox.pyx:440-457
                    box = p.bounding_box()
                    centre = (0.5 * (box.lower.x + box.upper.x), 0.5 * (box.lower.y + box.upper.y), 0.5 * (box.lower.z + box.upper.z))
                    x, y, z = centre[0] - box.lower.x, centre[1] - box.lower.y, centre[2] - box.lower.z
                    radius = math.sqrt(x * x + y * y + z * z) * 1.000001
                total += importance
                spheres.append((centre, radius, importance))
        if not spheres:
            return []
        cdf = np.zeros(len(spheres), dtype=np.float64)                      # _calculate_cdf
        for index, (_, _, importance) in enumerate(spheres):
            cdf[index] = importance if index == 0 else cdf[index - 1] + importance
        cdf /= total
        return [(centre, radius, float(cdf[i]), importance / total) for i, (centre, radius, importance) in enumerate(spheres)]
