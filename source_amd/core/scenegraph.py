"""
Scenegraph bookkeeping (host-only Python): Node / Primitive / Observer / World with Raysect's
constructor signatures and change-notification behaviour.

Mirrors raysect/core/scenegraph/{_nodebase,node,primitive,observer,world}.pyx. The tree lives on the
host exactly as in the reference; what changes is World.hit()/contains(): instead of walking Python
objects they flatten the scenegraph once (source_amd/_flatten.py), upload it to HBM and run the HIP
traversal kernels through librsx (include/rsx.h). A GEOMETRY change invalidates the device scene the same
way it invalidates the reference's accelerator (world.pyx:220-238).
"""
import weakref

from .math import AffineMatrix3D, Point3D, Vector3D


class Ray:
    """raysect/core/ray.pyx:38-145 — geometric ray: origin, direction, max_distance."""
    __slots__ = ("origin", "direction", "max_distance")

    def __init__(self, origin=None, direction=None, max_distance=float("inf")):
        self.origin = origin if origin is not None else Point3D(0, 0, 0)
        self.direction = direction if direction is not None else Vector3D(0, 0, 1)
        self.max_distance = float(max_distance)

    def __repr__(self):
        return "Ray(%r, %r, %r)" % (self.origin, self.direction, self.max_distance)

    def point_on(self, t):
        return Point3D(self.origin.x + t * self.direction.x, self.origin.y + t * self.direction.y, self.origin.z + t * self.direction.z)

    def copy(self, origin=None, direction=None):
        return Ray(origin or self.origin.copy(), direction or self.direction.copy(), self.max_distance)


class Intersection:
    """raysect/core/intersection.pyx:36-106 — points and normal are in primitive-local space."""
    __slots__ = ("ray", "ray_distance", "primitive", "hit_point", "inside_point", "outside_point", "normal",
                 "exiting", "world_to_primitive", "primitive_to_world")

    def __init__(self, ray, ray_distance, primitive, hit_point, inside_point, outside_point, normal, exiting,
                 world_to_primitive, primitive_to_world):
        self.ray = ray
        self.ray_distance = ray_distance
        self.primitive = primitive
        self.hit_point = hit_point
        self.inside_point = inside_point
        self.outside_point = outside_point
        self.normal = normal
        self.exiting = exiting
        self.world_to_primitive = world_to_primitive
        self.primitive_to_world = primitive_to_world

    def __repr__(self):
        return "Intersection(%r, %r, %r, exiting=%r)" % (self.ray, self.ray_distance, self.primitive, self.exiting)


class MeshIntersection(Intersection):
    """raysect/primitive/mesh/mesh.pyx:85-135 — adds triangle index and barycentrics."""
    __slots__ = ("triangle", "u", "v", "w")


class BoundingBox3D:
    """raysect/core/boundingbox.pyx — host-side subset used to build the scene description."""
    __slots__ = ("lower", "upper")

    def __init__(self, lower=None, upper=None):
        inf = float("inf")
        if lower is None or upper is None:
            self.lower, self.upper = Point3D(inf, inf, inf), Point3D(-inf, -inf, -inf)
        else:
            if lower.x > upper.x or lower.y > upper.y or lower.z > upper.z:
                raise ValueError("The lower point coordinates must be less than or equal to the upper point coordinates.")
            self.lower, self.upper = lower, upper

    def __repr__(self):
        return "BoundingBox3D(%r, %r)" % (self.lower, self.upper)

    def union(self, box):                                   # boundingbox.pyx:265-281
        lo, hi = self.lower, self.upper
        lo.x, lo.y, lo.z = min(lo.x, box.lower.x), min(lo.y, box.lower.y), min(lo.z, box.lower.z)
        hi.x, hi.y, hi.z = max(hi.x, box.upper.x), max(hi.y, box.upper.y), max(hi.z, box.upper.z)

    def extend(self, point, padding=0.0):                   # boundingbox.pyx:283-300
        lo, hi = self.lower, self.upper
        lo.x, lo.y, lo.z = min(lo.x, point.x - padding), min(lo.y, point.y - padding), min(lo.z, point.z - padding)
        hi.x, hi.y, hi.z = max(hi.x, point.x + padding), max(hi.y, point.y + padding), max(hi.z, point.z + padding)

    def vertices(self):                                     # boundingbox.pyx:326-343
        lo, hi = self.lower, self.upper
        return [Point3D(x, y, z) for x in (lo.x, hi.x) for y in (lo.y, hi.y) for z in (lo.z, hi.z)]

    def contains(self, p):
        lo, hi = self.lower, self.upper
        return not (p.x < lo.x or p.x > hi.x or p.y < lo.y or p.y > hi.y or p.z < lo.z or p.z > hi.z)

    def as_list(self):
        return [self.lower.x, self.lower.y, self.lower.z, self.upper.x, self.upper.y, self.upper.z]


GEOMETRY = "GEOMETRY"
MATERIAL = "MATERIAL"


class Node:
    """raysect/core/scenegraph/_nodebase.pyx + node.pyx:56-189."""

    def __init__(self, parent=None, transform=None, name=None):
        self._name = name
        self._parent = None
        self.children = []
        self.root = self
        self._transform = transform if transform is not None else AffineMatrix3D()
        self._root_transform = AffineMatrix3D()
        self._root_transform_inverse = AffineMatrix3D()
        self.meta = {}
        self.parent = parent

    def __repr__(self):
        return "<%s %s at 0x%x>" % (type(self).__name__, self._name or "", id(self))

    @property
    def name(self):
        return self._name

    @name.setter
    def name(self, value):
        self._name = value

    @property
    def parent(self):
        return self._parent

    @parent.setter
    def parent(self, value):
        if self._parent is value:
            return
        if value is not None:
            if not isinstance(value, Node):
                raise TypeError("The specified parent is not a scene-graph node or None (unparented).")
            value._check_parent(self)
        if self._parent is not None:
            self._parent.children.remove(self)
        self._parent = value
        if value is not None:
            value.children.append(self)
        self._update()

    def _check_parent(self, candidate_child):
        # a node may not become a descendant of itself
        node = self
        while node is not None:
            if node is candidate_child:
                raise ValueError("A node cannot be parented to itself or one of it's descendants.")
            node = node._parent

    @property
    def transform(self):
        return self._transform

    @transform.setter
    def transform(self, value):
        self._transform = value
        self._update()

    def _update(self):                                      # _nodebase.pyx:83-134
        if self._parent is None:
            if self.root is not self:
                self.root._deregister(self)
                self.root = self
            self._root_transform = AffineMatrix3D()
            self._root_transform_inverse = AffineMatrix3D()
        else:
            if self.root is not self._parent.root:
                self.root._deregister(self)
                self.root = self._parent.root
                self._parent.root._register(self)
            self._root_transform = self._parent._root_transform.mul(self._transform)
            self._root_transform_inverse = self._root_transform.inverse()
        self._modified()
        self.root._change(self, GEOMETRY)
        for child in self.children:
            child._update()

    def _modified(self):
        pass

    def _register(self, node):
        pass

    def _deregister(self, node):
        pass

    def _change(self, node, change):
        pass

    def to(self, node):                                     # node.pyx:135-171
        if self.root is node.root:
            return node._root_transform_inverse.mul(self._root_transform)
        raise ValueError("The target node must be in the same scene-graph.")

    def to_local(self):                                     # node.pyx:173-180
        return self._root_transform_inverse

    def to_root(self):                                      # node.pyx:182-189
        return self._root_transform


class Material:
    """raysect/core/material.pyx stub; optical materials derive from it."""

    def __init__(self):
        self.primitives = []

    def notify_material_change(self):
        for p in self.primitives:
            p.notify_material_change()


class Primitive(Node):
    """raysect/core/scenegraph/primitive.pyx:35-224."""

    def __init__(self, parent=None, transform=None, material=None, name=None):
        self._material = material if material is not None else Material()
        self._material.primitives.append(self)              # primitive.pyx:62-63: the material knows the primitives it coats
        self._geometry_version = 0
        super().__init__(parent, transform, name)

    @property
    def material(self):
        return self._material

    @material.setter
    def material(self, value):                              # primitive.pyx:84-96
        if value is None:
            value = Material()
        if not isinstance(value, Material):
            raise TypeError("The material must be a Material object or None.")
        if self in self._material.primitives:
            self._material.primitives.remove(self)
        self._material = value
        value.primitives.append(self)
        self.notify_material_change()

    def get_material(self):
        return self._material

    def hit(self, ray):
        raise NotImplementedError("Primitive surface has not been defined. Virtual method hit() has not been implemented.")

    def next_intersection(self):
        raise NotImplementedError("Primitive surface has not been defined. Virtual method next_intersection() has not been implemented.")

    def contains(self, point):
        raise NotImplementedError("Primitive surface has not been defined. Virtual method contains() has not been implemented.")

    def bounding_box(self):
        raise NotImplementedError("Primitive surface has not been defined. Virtual method bounding_box() has not been implemented.")

    def instance(self, parent=None, transform=None, material=None, name=None):
        raise NotImplementedError("Primitive surface has not been defined. Virtual method instance() has not been implemented.")

    def notify_geometry_change(self):                       # primitive.pyx:201-211
        self._geometry_version = getattr(self, "_geometry_version", 0) + 1     # also read by device.scene_for_primitive (unregistered primitives)
        self.root._change(self, GEOMETRY)

    def notify_material_change(self):
        self.root._change(self, MATERIAL)


class Observer(Node):
    """raysect/core/scenegraph/observer.pyx:35-49."""

    def observe(self):
        raise NotImplementedError("Observer is a virtual scene-graph object and cannot be used to observe the scene.")


class Accelerator:
    """core/acceleration/accelerator.pyx:35-68 — the plug-point World.accelerator accepts."""

    def build(self, primitives):
        raise NotImplementedError("Accelerator virtual method build() has not been implemented.")

    def hit(self, ray):
        raise NotImplementedError("Accelerator virtual method hit() has not been implemented.")

    def contains(self, point):
        raise NotImplementedError("Accelerator virtual method contains() has not been implemented.")


class HipAccelerator(Accelerator):
    """The default accelerator of a source_amd World (the reference's default is KDTree, world.pyx:52): the world KD-tree and every
    mesh KD-tree are built on the host exactly as the reference builds them (core/acceleration/kdtree.pyx:43, kdtree3d.pyx:126-486),
    uploaded to HBM and traversed by librsx (rsx_hit_batch / rsx_contains_batch). The object is a view of its World's device scene."""

    def __init__(self, world):
        self._world = world

    def build(self, primitives=None):
        return self._world.build_accelerator(force=True)

    def hit(self, ray):
        return self._world.build_accelerator().hit_single(ray)

    def contains(self, point):
        return World.contains(self._world, point)


class World(Node):
    """
    raysect/core/scenegraph/world.pyx:40-239. hit()/contains() keep their single-ray signatures; the
    batched forms hit_batch()/contains_batch() are what the render engine uses.
    """

    def __init__(self, name=None):
        self._primitives = []
        self._observers = []
        self._rebuild_accelerator = True
        self._device_scene = None
        self._accelerator = HipAccelerator(self)
        self._lazy_observers = weakref.WeakSet()            # observers holding accepted, not yet submitted passes (_settle_observers)
        super().__init__(None, None, name)

    @property
    def parent(self):
        return None

    @parent.setter
    def parent(self, value):
        if value is not None:
            raise RuntimeError("The World object cannot be parented to another scene-graph Node.")

    @property
    def accelerator(self):                                  # world.pyx:59-70
        """The acceleration structure in use. Default: HipAccelerator — the scenegraph flattened into HBM and traversed by librsx.
        Any object with the reference's Accelerator interface (build(primitives), hit(ray), contains(point);
        core/acceleration/accelerator.pxd:37-41) may be assigned instead; World.hit()/contains() then dispatch to it."""
        return self._accelerator

    @accelerator.setter
    def accelerator(self, value):
        for method in ("build", "hit", "contains"):
            if not callable(getattr(value, method, None)):
                raise TypeError("The accelerator must implement the Accelerator interface (build, hit, contains).")
        self._accelerator = value
        self._rebuild_accelerator = True

    @property
    def primitives(self):
        return list(self._primitives)

    @property
    def observers(self):
        return list(self._observers)

    def _register(self, node):                              # world.pyx:196-207
        self._settle_observers()
        if isinstance(node, Primitive):
            self._primitives.append(node)
            self._rebuild_accelerator = True
        if isinstance(node, Observer):
            self._observers.append(node)

    def _deregister(self, node):                            # world.pyx:209-218
        self._settle_observers()
        if isinstance(node, Primitive):
            self._primitives.remove(node)
            self._rebuild_accelerator = True
        if isinstance(node, Observer):
            self._observers.remove(node)

    def _change(self, node, change):                        # world.pyx:220-238
        self._settle_observers()
        if change is GEOMETRY:
            self._rebuild_accelerator = True

    def _settle_observers(self):
        """Passes that observers have accepted but not yet submitted (PinholeCamera batches small passes) are rendered before the
        scenegraph changes under them."""
        for obs in list(self._lazy_observers):
            self._lazy_observers.discard(obs)
            obs._flush_lazy()

    # -- device scene management ---------------------------------------------------------------
    def flatten(self):
        """Host-side flattening only (no GPU needed): returns source_amd._flatten.FlatScene."""
        from .._flatten import flatten_world
        return flatten_world(self)

    def build_accelerator(self, force=False):               # world.pyx:170-194
        """Rebuilds the acceleration structure if the scenegraph changed; returns the device scene the render path uses."""
        if self._rebuild_accelerator or force or self._device_scene is None:
            from ..device import DeviceScene
            if self._device_scene is not None:
                self._device_scene.close()
            self._device_scene = DeviceScene(self.flatten())
            if not isinstance(self._accelerator, HipAccelerator):
                self._accelerator.build(list(self._primitives))
            self._rebuild_accelerator = False
        return self._device_scene

    def hit(self, ray):                                     # world.pyx:125-146
        scene = self.build_accelerator()
        if not isinstance(self._accelerator, HipAccelerator):
            return self._accelerator.hit(ray)
        return scene.hit_single(ray)

    def hit_batch(self, origin, direction, max_distance=None, geometry=False):
        """Vector form of hit(): numpy [n,3] origins/directions -> dict of result arrays (prim == -1 on miss)."""
        return self.build_accelerator().hit_batch(origin, direction, max_distance, geometry)

    def contains(self, point):                              # world.pyx:149-168
        scene = self.build_accelerator()
        if not isinstance(self._accelerator, HipAccelerator):
            return self._accelerator.contains(point)
        flags = scene.contains_single(point)
        order = scene.flat.contains_order(point)
        return [self._primitives[i] for i in order if flags[i]]

    def contains_batch(self, points):
        return self.build_accelerator().contains_batch(points)
