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

    # -- the sampling half of the ImportanceManager, host form (world.pyx:130-230): what a material's evaluate_surface calls ------
    def _spheres_cached(self):
        if getattr(self, "_important_frozen", False):       # (inside one host-callback render: the scene cannot change between two hits)
            return self._important
        key = (id(self._device_scene), self._rebuild_accelerator, tuple(id(p.material) for p in self._primitives),
               tuple(getattr(p.material, "importance", 0.0) for p in self._primitives))
        if getattr(self, "_important_key", None) != key:
            self._important_key, self._important = key, self.important_spheres()
        return self._important

    def has_important_primitives(self):                                     # world.pyx:333-339
        return len(self._spheres_cached()) > 0

    def important_direction_sample(self, origin):                           # world.pyx:150-188
        """A direction from `origin` towards the bounding sphere of an important primitive picked by importance weight. Draws, in
        the reference's order: the selection uniform, then the direction pair (vector_sphere / vector_cone_uniform)."""
        import math
        from ..core import random as rsrandom
        from . import _portable as P
        spheres = self._spheres_cached()
        if not spheres:
            raise ValueError("Attempted to sample important direction when no important primitives have been specified.")
        pick = rsrandom.uniform()
        index = 0
        while index < len(spheres) - 1 and not (pick < spheres[index][2]):      # find_index(cdf, u) + 1
            index += 1
        align = getattr(rsrandom._override, "align", None)
        if align:
            align()                                                             # per-path Philox streams: the direction pair is its own counter
        ua, ub = rsrandom.uniform(), rsrandom.uniform()
        centre, radius = spheres[index][0], spheres[index][1]
        dx, dy, dz = centre[0] - origin.x, centre[1] - origin.y, centre[2] - origin.z
        distance = math.sqrt(dx * dx + dy * dy + dz * dz)
        portable = rsrandom._override is not None
        sincos = P.sincos if portable else (lambda a: (math.sin(a), math.cos(a)))
        if distance == 0 or distance < radius:                                  # vector_sphere, random.pyx:373-387
            z = 1.0 - 2.0 * ua
            r2 = 1.0 - z * z
            r = math.sqrt(r2 if r2 > 0 else 0)
            sn, cs = sincos(2.0 * math.pi * ub)
            return Vector3D(r * cs, r * sn, z)
        angular_radius = P.asin(radius / distance) if portable else math.asin(radius / distance)
        theta = angular_radius * 180 / math.pi                                  # vector_cone_uniform(degrees), random.pyx:425-446
        theta *= 0.017453292519943295
        phi = 2.0 * math.pi * ua
        cos_theta = sincos(theta)[1]
        z = ub * (1 - cos_theta) + cos_theta
        r2 = 1.0 - z * z
        r = math.sqrt(r2 if r2 > 0 else 0)
        sn, cs = sincos(phi)
        sx, sy, sz = r * cs, r * sn, z
        d = Vector3D(dx, dy, dz).normalise()
        up = d.orthogonal()
        right = up.cross(d)                                                     # the cimported rotate_basis: up.cross(forward), no re-normalisation
        return Vector3D(right.x * sx + up.x * sy + d.x * sz, right.y * sx + up.y * sy + d.y * sz, right.z * sx + up.z * sy + d.z * sz)

    def important_direction_pdf(self, origin, direction):                   # world.pyx:190-230
        import math
        pdf_all = 0
        for centre, radius, _, weight in self._spheres_cached():
            ax, ay, az = centre[0] - origin.x, centre[1] - origin.y, centre[2] - origin.z
            distance = math.sqrt(ax * ax + ay * ay + az * az)
            if distance == 0 or distance < radius:
                solid_angle = 4 * math.pi
            else:
                t = radius / distance
                angular_radius_cos = math.sqrt(1 - t * t)
                k = ax * ax + ay * ay + az * az
                k = 1.0 / math.sqrt(k)
                ax, ay, az = ax * k, ay * k, az * k
                if direction.x * ax + direction.y * ay + direction.z * az < angular_radius_cos:
                    continue
                solid_angle = 2 * math.pi * (1 - angular_radius_cos)
            pdf_sphere = 1 / solid_angle
            pdf_all += weight * pdf_sphere
        return pdf_all
