"""source_amd.optical — mirrors the hot-path names of raysect.optical."""
from ..core import (AffineMatrix3D, Normal3D, Point3D, Vector3D, rotate, rotate_vector, rotate_x, rotate_y, rotate_z, translate,
                    Node, Primitive, Intersection, BoundingBox3D)
from ..core.scenegraph import World as _CoreWorld
from .spectral import ConstantSF, InterpolatedSF, SpectralFunction, Spectrum
from .ray import Ray
from . import material
from .material import AbsorbingSurface, Light, Material, NullVolume, UniformSurfaceEmitter


class World(_CoreWorld):
    """raysect/optical/scenegraph/world.pyx — the optical world (the importance-sampling manager, SURVEY.md §8f row 2, is not built yet)."""
