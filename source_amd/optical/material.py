"""
Materials. The material plugin API (evaluate_surface / evaluate_volume, raysect/optical/material/material.pxd:36-47)
is kept as the host-side interface. The closed-form materials are lowered to the device render kernel: AbsorbingSurface,
UniformSurfaceEmitter and the debug Light (one world.hit() plus a bins-wide multiply) and — first slice of the
§8(f) "next" rows — the deterministic transparent ones, NullMaterial and UniformVolumeEmitter (null surfaces continue the ray,
every segment integrates the emission of the volumes it starts in) and Lambert (stochastic secondary rays, Philox-keyed). Any
other material raises when an observer tries to render it
on the device (there is no CPU fallback).

Mirrors raysect/optical/material/{material,absorber,debug}.pyx and emitter/uniform.pyx.
"""
from ..core.scenegraph import Material as CoreMaterial
from .. import _lib
from .spectral import ConstantSF, NumericallyIntegratedSF


class Material(CoreMaterial):
    """raysect/optical/material/material.pyx — base of the surface/volume plugin API."""

    def __init__(self):
        super().__init__()
        self.importance = 0.0

    def evaluate_surface(self, world, ray, primitive, hit_point, exiting, inside_point, outside_point, normal,
                         world_to_primitive, primitive_to_world, intersection):
        raise NotImplementedError("Material virtual method evaluate_surface() has not been implemented.")

    def evaluate_volume(self, spectrum, world, ray, primitive, start_point, end_point, world_to_primitive, primitive_to_world):
        raise NotImplementedError("Material virtual method evaluate_volume() has not been implemented.")

    def device_material(self, tables, min_wavelength, max_wavelength, bins):
        """Returns the rsx_material record for the device kernel, appending any spectral table it needs to
        ``tables`` (list of f64[bins] arrays). Materials that cannot run on the device raise."""
        raise NotImplementedError(
            "%s has no device lowering: only AbsorbingSurface, UniformSurfaceEmitter and debug Light render on the "
            "MI355X path in this version (secondary-ray materials are the next scope row, SURVEY.md §8f)." % type(self).__name__)


class NullVolume(Material):
    """material.pyx:150-165 — volume that contributes nothing."""

    def evaluate_volume(self, spectrum, world, ray, primitive, start_point, end_point, world_to_primitive, primitive_to_world):
        return spectrum


def _record(kind, table, scale, light=(0.0, 0.0, 0.0)):
    m = _lib.Material()
    m.type, m.table, m.scale = kind, table, float(scale)
    m.light_dir[0], m.light_dir[1], m.light_dir[2] = light
    return m


class AbsorbingSurface(NullVolume):
    """absorber.pyx:37-55 — zero spectrum."""

    def evaluate_surface(self, world, ray, primitive, hit_point, exiting, inside_point, outside_point, normal,
                         world_to_primitive, primitive_to_world, intersection):
        return ray.new_spectrum()

    def device_material(self, tables, min_wavelength, max_wavelength, bins):
        return _record(_lib.MAT_ABSORBER, 0, 0.0)


class UniformSurfaceEmitter(NullVolume):
    """emitter/uniform.pyx:36-88 — emission_spectrum.sample(bins) * scale."""

    def __init__(self, emission_spectrum, scale=1.0):
        super().__init__()
        self.emission_spectrum = emission_spectrum
        self.scale = float(scale)
        self.importance = 1.0

    def evaluate_surface(self, world, ray, primitive, hit_point, exiting, inside_point, outside_point, normal,
                         world_to_primitive, primitive_to_world, intersection):
        spectrum = ray.new_spectrum()
        emission = self.emission_spectrum.sample(spectrum.min_wavelength, spectrum.max_wavelength, spectrum.bins)
        spectrum.samples[:] = emission * self.scale
        return spectrum

    def device_material(self, tables, min_wavelength, max_wavelength, bins):
        tables.append(self.emission_spectrum.sample(min_wavelength, max_wavelength, bins))
        return _record(_lib.MAT_UNIFORM_EMITTER, len(tables) - 1, self.scale)


class Light(NullVolume):
    """debug.pyx:41-79 — Lambertian surface lit by a distant light: intensity * max(0, -L_local . n) * spectrum."""

    def __init__(self, light_direction, intensity=1.0, spectrum=None):
        super().__init__()
        self.light_direction = light_direction.normalise()
        self.intensity = max(0, intensity)
        if spectrum is None:
            raise ValueError("source_amd's debug Light needs an explicit spectrum (the reference defaults to its "
                             "d65_white library table, which is not part of this hot-path build).")
        self.spectrum = spectrum

    def evaluate_surface(self, world, ray, primitive, hit_point, exiting, inside_point, outside_point, normal,
                         world_to_primitive, primitive_to_world, intersection):
        spectrum = ray.new_spectrum()
        if self.intensity != 0.0:
            diffuse = self.intensity * max(0, -(self.light_direction.transform(world_to_primitive).dot(normal)))
            spectrum.samples[:] = diffuse * self.spectrum.sample(ray.min_wavelength, ray.max_wavelength, ray.bins)
        return spectrum

    def device_material(self, tables, min_wavelength, max_wavelength, bins):
        tables.append(self.spectrum.sample(min_wavelength, max_wavelength, bins))
        d = self.light_direction
        return _record(_lib.MAT_DEBUG_LIGHT, len(tables) - 1, float(self.intensity), (d.x, d.y, d.z))


class NullSurface(Material):
    """material.pyx:104-147 — a surface the ray passes straight through: the daughter ray starts on the far side of the boundary
    with the same direction, its depth is not increased and Russian roulette is disabled, so the continuation is deterministic."""

    def evaluate_surface(self, world, ray, primitive, hit_point, exiting, inside_point, outside_point, normal,
                         world_to_primitive, primitive_to_world, intersection):
        origin = (outside_point if exiting else inside_point).transform(primitive_to_world)
        daughter = ray.spawn_daughter(origin, ray.direction)
        daughter.depth -= 1
        return daughter.trace(world, keep_alive=True)


class NullMaterial(NullSurface):
    """material.pyx:166-200 — perfectly transparent: null surface and no volume contribution."""

    def evaluate_volume(self, spectrum, world, ray, primitive, start_point, end_point, world_to_primitive, primitive_to_world):
        return spectrum

    def device_material(self, tables, min_wavelength, max_wavelength, bins):
        return _record(_lib.MAT_NULL, 0, 0.0)


class UniformVolumeEmitter(NullSurface):
    """emitter/uniform.pyx:91-131 over HomogeneousVolumeEmitter (emitter/homogeneous.pyx:40-102): a transparent boundary whose
    interior emits emission_spectrum * scale (W/m^3/str/nm) — a path segment that starts inside adds emission * segment length."""

    def __init__(self, emission_spectrum, scale=1.0):
        super().__init__()
        self.emission_spectrum = emission_spectrum
        self.scale = float(scale)
        self.importance = 1.0

    def evaluate_volume(self, spectrum, world, ray, primitive, start_point, end_point, world_to_primitive, primitive_to_world):
        start, end = start_point.transform(world_to_primitive), end_point.transform(world_to_primitive)
        length = end.vector_to(start).length
        if length == 0:
            return spectrum
        emission = self.emission_spectrum.sample(spectrum.min_wavelength, spectrum.max_wavelength, spectrum.bins) * self.scale
        spectrum.samples[:] = spectrum.samples + emission * length
        return spectrum

    def device_material(self, tables, min_wavelength, max_wavelength, bins):
        tables.append(self.emission_spectrum.sample(min_wavelength, max_wavelength, bins))
        return _record(_lib.MAT_UNIFORM_VOLUME_EMITTER, len(tables) - 1, self.scale)


class Lambert(NullVolume):
    """lambert.pyx:40-112 under ContinuousBSDF.evaluate_surface (material.pyx:286-361) — ideal diffuse reflector: one cosine-weighted
    daughter ray per hit, spectrum = trace(daughter) * reflectivity * pdf / pdf. Rendered on the device with Philox-keyed scattering
    (RSX_MAT_LAMBERT); in a world with important primitives and ray_importance_sampling on, the outgoing direction is the reference's
    important-path / BSDF mixture (material.pyx:327-352), also on the device."""

    def __init__(self, reflectivity=None):
        super().__init__()
        self.reflectivity = ConstantSF(0.5) if reflectivity is None else reflectivity

    def evaluate_surface(self, *args, **kwargs):
        raise NotImplementedError("Lambert is path traced on the device by the observers (observe()); a host-side evaluate_surface for "
                                  "single Ray.trace() calls is not part of this build (there is no CPU rendering path)")

    def device_material(self, tables, min_wavelength, max_wavelength, bins):
        tables.append(self.reflectivity.sample(min_wavelength, max_wavelength, bins))
        return _record(_lib.MAT_LAMBERT, len(tables) - 1, 1.0)


class Sellmeier(NumericallyIntegratedSF):
    """dielectric.pyx:40-122 — three-term Sellmeier refractive index; wavelength in nm (the equation's is in micrometres)."""

    def __init__(self, b1, b2, b3, c1, c2, c3, sample_resolution=10):
        super().__init__(sample_resolution)
        self.b1, self.b2, self.b3, self.c1, self.c2, self.c3 = float(b1), float(b2), float(b3), float(c1), float(c2), float(c3)

    def function(self, wavelength):
        import math
        w2 = wavelength * wavelength * 1e-6
        return math.sqrt(1 + (self.b1 * w2) / (w2 - self.c1) + (self.b2 * w2) / (w2 - self.c2) + (self.b3 * w2) / (w2 - self.c3))


class Dielectric(Material):
    """dielectric.pyx:125-328 — ideal dielectric: Fresnel-weighted stochastic choice between the refracted and the reflected ray
    (total internal reflection handled), refractive indices averaged over the ray's spectral range (one index per spectral slice:
    dispersion comes from spectral_rays), Beer-Lambert attenuation transmission ** path_length inside the primitive.
    Rendered on the device (RSX_MAT_DIELECTRIC) with Philox-keyed paths."""

    def __init__(self, index, transmission, external_index=None, transmission_only=False):
        super().__init__()
        self.index = index
        self.transmission = transmission
        self.transmission_only = bool(transmission_only)
        self.external_index = ConstantSF(1.0) if external_index is None else external_index
        self.importance = 1.0

    def evaluate_surface(self, *args, **kwargs):
        raise NotImplementedError("Dielectric is path traced on the device by the observers (observe()); a host-side evaluate_surface "
                                  "for single Ray.trace() calls is not part of this build (there is no CPU rendering path)")

    def device_material(self, tables, min_wavelength, max_wavelength, bins):
        tables.append(self.transmission.sample(min_wavelength, max_wavelength, bins))
        internal = self.index.average(min_wavelength, max_wavelength)
        external = self.external_index.average(min_wavelength, max_wavelength)
        return _record(_lib.MAT_DIELECTRIC, len(tables) - 1, internal, (external, 1.0 if self.transmission_only else 0.0, 0.0))


# convenience used by scene builders
def default_white():
    return ConstantSF(1.0)
