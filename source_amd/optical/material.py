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
import math

from ..core import random as rsrandom
from ..core.math import AffineMatrix3D, Vector3D
from ..core.scenegraph import Material as CoreMaterial
from .. import _lib
from . import _portable
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
        """Returns the rsx_material record for the device render kernels, appending any spectral table it needs to
        ``tables`` (list of f64[bins] arrays). A material without a lowering returns None: observe() then renders the scene through
        the batched host-callback path (source_amd/optical/hybrid.py — rays traced on the device wave by wave, evaluate_surface /
        evaluate_volume of every hit called on the host), which is what makes user-written Material subclasses work unchanged."""
        return None


_PLUGIN_HOOKS = ("evaluate_surface", "evaluate_volume", "evaluate_shading", "sample", "pdf")


def has_device_lowering(material):
    """True when `material` renders in the device kernels: its class (or a base) defines device_material(), and no class further
    down the hierarchy re-defines one of the plugin methods without re-defining the lowering (a user subclass of Lambert that
    overrides evaluate_shading is a new material: it goes through the host-callback path)."""
    mro = type(material).__mro__
    owner = next((c for c in mro if "device_material" in vars(c)), None)
    if owner is None or owner is Material:
        return False
    below = mro[:mro.index(owner)]
    return not any(hook in vars(c) for c in below for hook in _PLUGIN_HOOKS)


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


def _generate_surface_transforms(normal):                   # material.pyx:393-422
    tangent = normal.orthogonal()
    bitangent = normal.cross(tangent)
    primitive_to_surface = AffineMatrix3D._new(tangent.x, tangent.y, tangent.z, 0.0, bitangent.x, bitangent.y, bitangent.z, 0.0,
                                               normal.x, normal.y, normal.z, 0.0, 0.0, 0.0, 0.0, 1.0)
    surface_to_primitive = AffineMatrix3D._new(tangent.x, bitangent.x, normal.x, 0.0, tangent.y, bitangent.y, normal.y, 0.0,
                                               tangent.z, bitangent.z, normal.z, 0.0, 0.0, 0.0, 0.0, 1.0)
    return primitive_to_surface, surface_to_primitive


def _surface_frame(exiting, inside_point, outside_point, normal, world_to_primitive, primitive_to_world):
    """Launch points and surface-space transforms shared by the BSDF base classes (material.pyx:222-246, 304-325)."""
    if exiting:                                             # ray incident on the back face
        w_reflection_origin = inside_point.transform(primitive_to_world)
        w_transmission_origin = outside_point.transform(primitive_to_world)
        normal = normal.neg()
    else:
        w_reflection_origin = outside_point.transform(primitive_to_world)
        w_transmission_origin = inside_point.transform(primitive_to_world)
    primitive_to_surface, surface_to_primitive = _generate_surface_transforms(normal)
    world_to_surface = primitive_to_surface.mul(world_to_primitive)
    surface_to_world = primitive_to_world.mul(surface_to_primitive)
    return w_reflection_origin, w_transmission_origin, world_to_surface, surface_to_world


class DiscreteBSDF(Material):
    """material.pyx:219-259 — base class of materials with a discrete BSDF (mirrors, ideal interfaces): evaluate_shading() gets the
    incident direction in surface space and the two launch points."""

    def evaluate_surface(self, world, ray, primitive, hit_point, exiting, inside_point, outside_point, normal,
                         world_to_primitive, primitive_to_world, intersection):
        w_refl, w_trans, world_to_surface, surface_to_world = _surface_frame(exiting, inside_point, outside_point, normal, world_to_primitive, primitive_to_world)
        s_incoming = ray.direction.transform(world_to_surface).neg()
        return self.evaluate_shading(world, ray, s_incoming, w_refl, w_trans, exiting, world_to_surface, surface_to_world, intersection)

    def evaluate_shading(self, world, ray, s_incoming, w_reflection_origin, w_transmission_origin, back_face, world_to_surface, surface_to_world, intersection):
        raise NotImplementedError("Virtual method evaluate_shading() has not been implemented.")


class ContinuousBSDF(Material):
    """material.pyx:278-391 — base class of materials with a continuous BSDF: sample() / pdf() / evaluate_shading(); in a world with
    important primitives the outgoing direction is drawn from the important-path / BSDF mixture (multiple importance sampling)."""

    def evaluate_surface(self, world, ray, primitive, hit_point, exiting, inside_point, outside_point, normal,
                         world_to_primitive, primitive_to_world, intersection):
        w_refl, w_trans, world_to_surface, surface_to_world = _surface_frame(exiting, inside_point, outside_point, normal, world_to_primitive, primitive_to_world)
        s_incoming = ray.direction.transform(world_to_surface).neg()
        if ray.importance_sampling and world.has_important_primitives():
            w_hit_point = hit_point.transform(primitive_to_world)
            align = getattr(rsrandom._override, "align", None)      # per-path Philox streams (hybrid.py): the direction pair has its own counter
            if rsrandom.probability(ray.important_path_weight):
                w_outgoing = world.important_direction_sample(w_hit_point)
                s_outgoing = w_outgoing.transform(world_to_surface)
            else:
                if align:
                    align()
                s_outgoing = self.sample(s_incoming, exiting)
                w_outgoing = s_outgoing.transform(surface_to_world)
            pdf_important = world.important_direction_pdf(w_hit_point, w_outgoing)
            pdf_bsdf = self.pdf(s_incoming, s_outgoing, exiting)
            pdf = ray.important_path_weight * pdf_important + (1 - ray.important_path_weight) * pdf_bsdf
            spectrum = self.evaluate_shading(world, ray, s_incoming, s_outgoing, w_refl, w_trans, exiting, world_to_surface, surface_to_world, intersection)
            spectrum.div_scalar(pdf)
            return spectrum
        s_outgoing = self.sample(s_incoming, exiting)
        spectrum = self.evaluate_shading(world, ray, s_incoming, s_outgoing, w_refl, w_trans, exiting, world_to_surface, surface_to_world, intersection)
        pdf = self.pdf(s_incoming, s_outgoing, exiting)
        spectrum.div_scalar(pdf)
        return spectrum

    def pdf(self, s_incoming, s_outgoing, back_face):
        raise NotImplementedError("Virtual method pdf() has not been implemented.")

    def sample(self, s_incoming, back_face):
        raise NotImplementedError("Virtual method sample() has not been implemented.")

    def evaluate_shading(self, world, ray, s_incoming, s_outgoing, w_reflection_origin, w_transmission_origin, back_face,
                         world_to_surface, surface_to_world, intersection):
        raise NotImplementedError("Virtual method evaluate_shading() has not been implemented.")

    def bsdf(self, s_incident, s_reflected, wavelength):
        raise NotImplementedError("This ContinuousBSDF material has not implemented the bsdf() method.")

    def evaluate_volume(self, spectrum, world, ray, primitive, start_point, end_point, world_to_primitive, primitive_to_world):
        return spectrum


def hemisphere_cosine_sample():
    """HemisphereCosineSampler.sample (core/math/sampler/solidangle.pyx:228-233). With a per-path Philox stream active the sine and
    cosine are the portable pair the device uses, so host and device paths take the same turns."""
    r = math.sqrt(rsrandom.uniform())
    phi = 2.0 * math.pi * rsrandom.uniform()
    if rsrandom._override is not None:
        sn, cs = _portable.sincos(phi)
    else:
        sn, cs = math.sin(phi), math.cos(phi)
    x, y = r * cs, r * sn
    z2 = 1.0 - x * x - y * y
    return Vector3D(x, y, math.sqrt(z2 if z2 > 0 else 0))


def hemisphere_cosine_pdf(sample):                          # solidangle.pyx:223-226
    return (1.0 / math.pi) * sample.z if sample.z >= 0.0 else 0.0


class Lambert(ContinuousBSDF):
    """lambert.pyx:40-112 under ContinuousBSDF.evaluate_surface (material.pyx:286-361) — ideal diffuse reflector: one cosine-weighted
    daughter ray per hit, spectrum = trace(daughter) * reflectivity * pdf / pdf. Rendered on the device with Philox-keyed scattering
    (RSX_MAT_LAMBERT); in a world with important primitives and ray_importance_sampling on, the outgoing direction is the reference's
    important-path / BSDF mixture (material.pyx:327-352), also on the device."""

    def __init__(self, reflectivity=None):
        super().__init__()
        self.reflectivity = ConstantSF(0.5) if reflectivity is None else reflectivity

    # host form (lambert.pyx:71-104): what a single Ray.trace() and the host-callback render path evaluate
    def pdf(self, s_incoming, s_outgoing, back_face):
        return hemisphere_cosine_pdf(s_outgoing)

    def sample(self, s_incoming, back_face):
        return hemisphere_cosine_sample()

    def evaluate_shading(self, world, ray, s_incoming, s_outgoing, w_reflection_origin, w_transmission_origin, back_face,
                         world_to_surface, surface_to_world, intersection):
        pdf = hemisphere_cosine_pdf(s_outgoing)
        if pdf == 0.0:
            return ray.new_spectrum()
        reflected = ray.spawn_daughter(w_reflection_origin, s_outgoing.transform(surface_to_world))
        spectrum = reflected.trace(world)
        spectrum.mul_array(self.reflectivity.sample(spectrum.min_wavelength, spectrum.max_wavelength, spectrum.bins))
        spectrum.mul_scalar(pdf)
        return spectrum

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

    # host form (dielectric.pyx:159-328): what a single Ray.trace() and the host-callback render path evaluate
    def evaluate_surface(self, world, ray, primitive, hit_point, exiting, inside_point, outside_point, normal,
                         world_to_primitive, primitive_to_world, intersection):
        incident = ray.direction.transform(world_to_primitive).normalise()
        normal = normal.normalise()
        c1 = -normal.dot(incident)
        internal_index = self.index.average(ray.min_wavelength, ray.max_wavelength)
        external_index = self.external_index.average(ray.min_wavelength, ray.max_wavelength)
        n1, n2 = (internal_index, external_index) if c1 < 0.0 else (external_index, internal_index)
        gamma = n1 / n2
        c2s = 1 - (gamma * gamma) * (1 - c1 * c1)
        if c2s <= 0:                                        # total internal reflection
            if self.transmission_only:
                return ray.new_spectrum()
            temp = 2 * c1
            reflected = Vector3D(incident.x + temp * normal.x, incident.y + temp * normal.y, incident.z + temp * normal.z).transform(primitive_to_world)
            origin = (inside_point if c1 < 0.0 else outside_point).transform(primitive_to_world)
            return ray.spawn_daughter(origin, reflected).trace(world)
        temp = gamma * c1 + math.sqrt(c2s) if c1 < 0.0 else gamma * c1 - math.sqrt(c2s)
        transmitted = Vector3D(gamma * incident.x + temp * normal.x, gamma * incident.y + temp * normal.y, gamma * incident.z + temp * normal.z)
        ci, ct = c1, -normal.dot(transmitted)
        ra, rb = (n1 * ci - n2 * ct) / (n1 * ci + n2 * ct), (n1 * ct - n2 * ci) / (n1 * ct + n2 * ci)
        reflectivity = 0.5 * (ra * ra + rb * rb)
        transmission = 1 - reflectivity
        if self.transmission_only or rsrandom.probability(transmission):
            origin = (outside_point if c1 < 0.0 else inside_point).transform(primitive_to_world)
            return ray.spawn_daughter(origin, transmitted.transform(primitive_to_world)).trace(world)
        temp = 2 * c1
        reflected = Vector3D(incident.x + temp * normal.x, incident.y + temp * normal.y, incident.z + temp * normal.z).transform(primitive_to_world)
        origin = (inside_point if c1 < 0.0 else outside_point).transform(primitive_to_world)
        return ray.spawn_daughter(origin, reflected).trace(world)

    def evaluate_volume(self, spectrum, world, ray, primitive, start_point, end_point, world_to_primitive, primitive_to_world):
        length = start_point.vector_to(end_point).length
        transmission = self.transmission.sample(spectrum.min_wavelength, spectrum.max_wavelength, spectrum.bins)
        if rsrandom._override is not None:                  # host-callback render: the device's portable pow, bit for bit
            for i in range(spectrum.bins):
                if transmission[i] != 1.0:
                    spectrum.samples[i] = spectrum.samples[i] * _portable.pow(float(transmission[i]), length)
        else:
            spectrum.samples[:] = spectrum.samples * transmission ** length
        return spectrum

    def device_material(self, tables, min_wavelength, max_wavelength, bins):
        tables.append(self.transmission.sample(min_wavelength, max_wavelength, bins))
        internal = self.index.average(min_wavelength, max_wavelength)
        external = self.external_index.average(min_wavelength, max_wavelength)
        return _record(_lib.MAT_DIELECTRIC, len(tables) - 1, internal, (external, 1.0 if self.transmission_only else 0.0, 0.0))


# convenience used by scene builders
def default_white():
    return ConstantSF(1.0)
