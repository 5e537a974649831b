"""source_amd.core — mirrors the names exported by raysect.core that sit on the hot path."""
from .math import (AffineMatrix3D, Normal3D, Point3D, Vector3D, rotate, rotate_vector, rotate_x, rotate_y, rotate_z, translate)
from .scenegraph import (BoundingBox3D, GEOMETRY, Intersection, MATERIAL, Material, MeshIntersection, Node, Observer, Primitive,
                         Ray, World)
from . import random  # noqa: F401
