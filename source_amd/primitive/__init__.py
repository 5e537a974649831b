"""
Primitives with Raysect's constructor signatures: Sphere, Box, Cylinder, Mesh and
the CSG operators Union / Intersect / Subtract. The Python objects only hold parameters and compute
bounding boxes (host f64, same operation order as the reference); intersection itself runs on the GPU:
``hit()`` / ``next_intersection()`` / ``contains()`` route through librsx (rsx_roots_batch /
rsx_contains_batch) on a device scene flattened from the primitive's scenegraph root.

Mirrors raysect/primitive/{sphere,box,cylinder,csg}.pyx and raysect/primitive/mesh/mesh.pyx.
"""
import io
import struct

import numpy as np

from ..core.math import AffineMatrix3D, Normal3D, Point3D, Vector3D
from ..core.scenegraph import BoundingBox3D, Intersection, MeshIntersection, Node, Primitive, World
from .. import _lib

BOX_PADDING = 1e-9          # sphere.pyx / box.pyx / cylinder.pyx / csg.pyx
MESH_BOX_PADDING = 1e-6     # mesh.pyx:42


class _DevicePrimitive(Primitive):
    """hit()/next_intersection()/contains() through the device for any primitive type."""

    _roots_cache = None

    def _scene_and_index(self):
        from ..device import scene_for_primitive
        return scene_for_primitive(self)

    def hit(self, ray):
        scene, idx = self._scene_and_index()
        self._roots_cache = scene.roots_single(idx, ray, self)
        return self.next_intersection()

    def next_intersection(self):
        if not self._roots_cache:
            return None
        return self._roots_cache.pop(0)

    def contains(self, point):
        scene, idx = self._scene_and_index()
        return bool(scene.prim_contains(idx, point))

    def _world_box(self, points):
        box = BoundingBox3D()
        m = self.to_root()
        for p in points:
            box.extend(p.transform(m), BOX_PADDING)
        return box


class Sphere(_DevicePrimitive):
    """raysect/primitive/sphere.pyx:73-234"""

    def __init__(self, radius=0.5, parent=None, transform=None, material=None, name=None):
        if radius < 0.0:
            raise ValueError("Sphere radius cannot be less than zero.")
        self._radius = float(radius)
        super().__init__(parent, transform, material, name)

    @property
    def radius(self):
        return self._radius

    @radius.setter
    def radius(self, value):
        if value == self._radius:
            return
        if value < 0.0:
            raise ValueError("Sphere radius cannot be less than zero.")
        self._radius = float(value)
        self.notify_geometry_change()

    def bounding_box(self):                                 # sphere.pyx:216-229
        o = Point3D(0, 0, 0).transform(self.to_root())
        e = self._radius + BOX_PADDING
        return BoundingBox3D(Point3D(o.x - e, o.y - e, o.z - e), Point3D(o.x + e, o.y + e, o.z + e))

    def instance(self, parent=None, transform=None, material=None, name=None):
        return Sphere(self._radius, parent, transform, material, name)


class Box(_DevicePrimitive):
    """raysect/primitive/box.pyx:84-437"""

    def __init__(self, lower=None, upper=None, parent=None, transform=None, material=None, name=None):
        if lower is not None and upper is not None:
            if lower.x > upper.x or lower.y > upper.y or lower.z > upper.z:
                raise ValueError("The lower point coordinates must be less than or equal to the upper point coordinates.")
            self._lower, self._upper = lower, upper
        elif lower is None and upper is None:
            self._lower, self._upper = Point3D(-0.5, -0.5, -0.5), Point3D(0.5, 0.5, 0.5)
        else:
            raise ValueError("Lower and upper points must both be defined.")
        super().__init__(parent, transform, material, name)

    @property
    def lower(self):
        return self._lower

    @lower.setter
    def lower(self, value):
        if value.x > self._upper.x or value.y > self._upper.y or value.z > self._upper.z:
            raise ValueError("The lower point coordinates must be less than or equal to the upper point coordinates.")
        self._lower = value
        self.notify_geometry_change()

    @property
    def upper(self):
        return self._upper

    @upper.setter
    def upper(self, value):
        if self._lower.x > value.x or self._lower.y > value.y or self._lower.z > value.z:
            raise ValueError("The upper point coordinates must be greater than or equal to the lower point coordinates.")
        self._upper = value
        self.notify_geometry_change()

    def bounding_box(self):                                 # box.pyx:405-432
        return self._world_box(BoundingBox3D(self._lower.copy(), self._upper.copy()).vertices())

    def instance(self, parent=None, transform=None, material=None, name=None):
        return Box(self._lower.copy(), self._upper.copy(), parent, transform, material, name)


class Cylinder(_DevicePrimitive):
    """raysect/primitive/cylinder.pyx:86-447 — radius r, height h along +z from z=0."""

    def __init__(self, radius=0.5, height=1.0, parent=None, transform=None, material=None, name=None):
        if radius < 0.0:
            raise ValueError("Cylinder radius cannot be less than zero.")
        if height < 0.0:
            raise ValueError("Cylinder height cannot be less than zero.")
        self._radius, self._height = float(radius), float(height)
        super().__init__(parent, transform, material, name)

    @property
    def radius(self):
        return self._radius

    @radius.setter
    def radius(self, value):
        if value < 0.0:
            raise ValueError("Cylinder radius cannot be less than zero.")
        self._radius = float(value)
        self.notify_geometry_change()

    @property
    def height(self):
        return self._height

    @height.setter
    def height(self, value):
        if value < 0.0:
            raise ValueError("Cylinder height cannot be less than zero.")
        self._height = float(value)
        self.notify_geometry_change()

    def bounding_box(self):                                 # cylinder.pyx:402-425
        r, h = self._radius, self._height
        return self._world_box(BoundingBox3D(Point3D(-r, -r, 0.0), Point3D(r, r, h)).vertices())

    def instance(self, parent=None, transform=None, material=None, name=None):
        return Cylinder(self._radius, self._height, parent, transform, material, name)


# ------------------------------------------------------------------------------------------------
# CSG — raysect/primitive/csg.pyx
# ------------------------------------------------------------------------------------------------
class NullPrimitive(Primitive):
    """csg.pyx:236-247 — placeholder operand with an empty bounding box."""

    def bounding_box(self):
        return BoundingBox3D()


class CSGRoot(Node):
    """csg.pyx:250-277 — private scenegraph root of a CSG node's operands."""

    def __init__(self, csg_primitive):
        self.csg_primitive = csg_primitive
        super().__init__()

    def _change(self, node, change):
        self.csg_primitive.root._change(node, change)


class CSGPrimitive(_DevicePrimitive):
    """csg.pyx:40-234"""

    def __init__(self, primitive_a=None, primitive_b=None, parent=None, transform=None, material=None, name=None):
        self._primitive_a = primitive_a or NullPrimitive()
        self._primitive_b = primitive_b or NullPrimitive()
        self._csgroot = None
        super().__init__(parent, transform, material, name)
        self._csgroot = CSGRoot(self)
        self._primitive_a.parent = self._csgroot
        self._primitive_b.parent = self._csgroot

    @property
    def primitive_a(self):
        return self._primitive_a

    @primitive_a.setter
    def primitive_a(self, primitive):
        self._primitive_a.parent = None
        self._primitive_a = primitive
        primitive.parent = self._csgroot
        self.notify_geometry_change()

    @property
    def primitive_b(self):
        return self._primitive_b

    @primitive_b.setter
    def primitive_b(self, primitive):
        self._primitive_b.parent = None
        self._primitive_b = primitive
        primitive.parent = self._csgroot
        self.notify_geometry_change()

    def _local_box(self):
        raise NotImplementedError

    def bounding_box(self):
        return self._world_box(self._local_box().vertices())

    def instance(self, parent=None, transform=None, material=None, name=None):
        return type(self)(self._primitive_a.instance(), self._primitive_b.instance(), parent, transform, material, name)


class Union(CSGPrimitive):
    """csg.pyx:280-383"""

    def _local_box(self):                                   # :355-375
        box = BoundingBox3D()
        box.union(self._primitive_a.bounding_box())
        box.union(self._primitive_b.bounding_box())
        return box


class Intersect(CSGPrimitive):
    """csg.pyx:386-468"""

    def _local_box(self):                                   # :453-468 (may be inverted when the operands' boxes are disjoint)
        a, b = self._primitive_a.bounding_box(), self._primitive_b.bounding_box()
        box = BoundingBox3D()
        box.lower = Point3D(max(a.lower.x, b.lower.x), max(a.lower.y, b.lower.y), max(a.lower.z, b.lower.z))
        box.upper = Point3D(min(a.upper.x, b.upper.x), min(a.upper.y, b.upper.y), min(a.upper.z, b.upper.z))
        return box


class Subtract(CSGPrimitive):
    """csg.pyx:471-599"""

    def _local_box(self):                                   # :575-592 (A's box)
        return self._primitive_a.bounding_box()


# ------------------------------------------------------------------------------------------------
# Mesh — raysect/primitive/mesh/mesh.pyx
# ------------------------------------------------------------------------------------------------
class MeshData:
    """
    mesh.pyx:142-1045 MeshData: f32 vertices / normals, i32 triangles, f32 face normals and the SAH KD-tree
    over padded triangle boxes. Built by librsx's host builders (C++), shared by all instances of a mesh and
    uploaded to HBM once per device scene.
    """

    def __init__(self, vertices, triangles, normals=None, smoothing=True, closed=True, tolerant=True, flip_normals=False,
                 max_depth=0, min_items=1, hit_cost=20.0, empty_bonus=0.2):
        L = _lib.lib()
        self.smoothing = bool(smoothing)
        self.closed = bool(closed)
        vertices = np.array(vertices, dtype=np.float32)
        triangles = np.array(triangles, dtype=np.int32)
        vertex_normals = np.array(normals, dtype=np.float32) if normals is not None else None
        if vertices.ndim != 2 or vertices.shape[1] != 3:
            raise ValueError("The vertex array must have dimensions Nx3.")
        if vertex_normals is not None:
            if vertex_normals.ndim != 2 or vertex_normals.shape[1] != 3:
                raise ValueError("The normal array must have dimensions Nx3.")
            if triangles.ndim != 2 or triangles.shape[1] != 6:
                raise ValueError("The triangle array must have dimensions Nx6.")
        elif triangles.ndim != 2 or triangles.shape[1] != 3:
            raise ValueError("The triangle array must have dimensions Nx3.")
        if ((triangles[:, 0:3] < 0) | (triangles[:, 0:3] >= vertices.shape[0])).any():
            raise ValueError("The triangle array references non-existent vertices.")
        if vertex_normals is not None and ((triangles[:, 3:6] < 0) | (triangles[:, 3:6] >= vertex_normals.shape[0])).any():
            raise ValueError("The triangle array references non-existent normals.")
        vertices = np.ascontiguousarray(vertices)
        triangles = np.ascontiguousarray(triangles)
        stride = triangles.shape[1]
        self._all_triangles = triangles                        # property keeps the unfiltered length
        n = triangles.shape[0]
        if tolerant:                                           # _filter_triangles, mesh.pyx:363-399
            n = L.rsx_mesh_filter_triangles(_lib.ptr(vertices), _lib.ptr(triangles), n, stride)
            if n < 0:
                _lib.check(n)
        tris = triangles[:n]
        if flip_normals:                                       # _flip_normals, mesh.pyx:401-426
            tris[:, [0, 2]] = tris[:, [2, 0]]
            if vertex_normals is not None:
                tris[:, [3, 5]] = tris[:, [5, 3]]
                vertex_normals = -vertex_normals
        self._vertices = vertices
        self._vertex_normals = None if vertex_normals is None else np.ascontiguousarray(vertex_normals)
        self._triangles = tris
        self._face_normals = np.zeros((n, 3), dtype=np.float32)
        _lib.check(L.rsx_mesh_face_normals(_lib.ptr(vertices), _lib.ptr(tris), n, stride, _lib.ptr(self._face_normals)))
        boxes = np.zeros((n, 6), dtype=np.float64)
        _lib.check(L.rsx_mesh_triangle_aabbs(_lib.ptr(vertices), _lib.ptr(tris), n, stride, _lib.ptr(boxes)))
        self._kd_params = (max(0, int(max_depth)), max(1, int(min_items)), max(1.0, float(hit_cost)), float(empty_bonus))
        self.kd = KDTreeHost.build(boxes, max_depth, min_items, hit_cost, empty_bonus)

    # -- accessors (copies, as the reference) ----------------------------------------------------
    @property
    def vertices(self):
        return self._vertices.copy()

    @property
    def triangles(self):
        return self._all_triangles.copy()

    @property
    def vertex_normals(self):
        return None if self._vertex_normals is None else self._vertex_normals.copy()

    @property
    def face_normals(self):
        return self._face_normals.copy()

    def bounding_box(self, to_world):                       # mesh.pyx:835-859
        out = np.zeros(6)
        m = np.array(to_world.m, dtype=np.float64)
        _lib.check(_lib.lib().rsx_mesh_world_bbox(_lib.ptr(self._vertices), self._vertices.shape[0], _lib.ptr(m), _lib.ptr(out)))
        return BoundingBox3D(Point3D(*out[:3]), Point3D(*out[3:]))

    @classmethod
    def from_file(cls, file):
        """RSM v1.0 reader (MeshData.load / from_file, mesh.pyx:933-1045): vertices, normals and triangles as stored, the KD-tree
        taken from the file unchanged, face normals recomputed (mesh.pyx:1012-1013)."""
        close = False
        if isinstance(file, str):
            file = open(file, "rb")
            close = True
        try:
            blob = file.read()
        finally:
            if close:
                file.close()
        if blob[:3] != b"RSM":
            raise ValueError("The specified mesh file is not a valid RSM mesh file.")
        major, minor = struct.unpack_from("<BB", blob, 3)
        if (major, minor) != (1, 0):
            raise ValueError("Unsupported Raysect mesh version (v%d.%d)." % (major, minor))
        smoothing, closed, has_tree = struct.unpack_from("<???", blob, 5)
        nv, nn, nt = struct.unpack_from("<iii", blob, 8)
        pos = 20
        self = cls.__new__(cls)
        self.smoothing, self.closed = bool(smoothing), bool(closed)
        self._vertices = np.frombuffer(blob, dtype="<f4", count=3 * nv, offset=pos).reshape(nv, 3).astype(np.float32)
        pos += 12 * nv
        self._vertex_normals = None
        if nn:
            self._vertex_normals = np.frombuffer(blob, dtype="<f4", count=3 * nn, offset=pos).reshape(nn, 3).astype(np.float32)
            pos += 12 * nn
        stride = 6 if nn else 3
        self._triangles = np.frombuffer(blob, dtype="<i4", count=stride * nt, offset=pos).reshape(nt, stride).astype(np.int32)
        self._all_triangles = self._triangles
        pos += 4 * stride * nt
        L = _lib.lib()
        self._face_normals = np.zeros((nt, 3), dtype=np.float32)
        _lib.check(L.rsx_mesh_face_normals(_lib.ptr(self._vertices), _lib.ptr(self._triangles), nt, stride, _lib.ptr(self._face_normals)))
        if has_tree:
            self.kd, used = KDTreeHost.from_blob(memoryview(blob)[pos:])
            self._kd_params = (self.kd.max_depth, self.kd._min_items, self.kd._hit_cost, self.kd._empty_bonus)
        else:                                                # mesh.pyx:1017-1019: rebuild with the default settings
            boxes = np.zeros((nt, 6), dtype=np.float64)
            _lib.check(L.rsx_mesh_triangle_aabbs(_lib.ptr(self._vertices), _lib.ptr(self._triangles), nt, stride, _lib.ptr(boxes)))
            self._kd_params = (0, 1, 20.0, 0.2)
            self.kd = KDTreeHost.build(boxes, 0, 1, 20.0, 0.2)
        return self

    def save(self, file):
        """RSM v1.0 writer, byte-compatible with MeshData.save (mesh.pyx:864-931)."""
        close = False
        if isinstance(file, str):
            file = open(file, "wb")
            close = True
        nn = 0 if self._vertex_normals is None else self._vertex_normals.shape[0]
        file.write(b"RSM")
        file.write(struct.pack("<BB???", 1, 0, self.smoothing, self.closed, True))
        file.write(struct.pack("<iii", self._vertices.shape[0], nn, self._triangles.shape[0]))
        file.write(self._vertices.astype("<f4").tobytes())
        if nn:
            file.write(self._vertex_normals.astype("<f4").tobytes())
        file.write(np.ascontiguousarray(self._triangles).astype("<i4").tobytes())
        file.write(self.kd.serialise(*self._kd_params[1:]))
        if close:
            file.close()


class KDTreeHost:
    """Host copy of a flattened KD-tree (numpy), built by rsx_kd_build (kdtree3d.pyx:126-486 semantics)."""

    def __init__(self, nodes, items, lower, upper, max_depth):
        self.nodes, self.items = nodes, items
        self.lower, self.upper = np.array(lower, dtype=np.float64), np.array(upper, dtype=np.float64)
        self.max_depth = int(max_depth)

    @classmethod
    def build(cls, boxes, max_depth=0, min_items=1, hit_cost=20.0, empty_bonus=0.2):
        import ctypes as C
        L = _lib.lib()
        boxes = np.ascontiguousarray(boxes, dtype=np.float64).reshape(-1, 6)
        handle = C.c_void_p()
        _lib.check(L.rsx_kd_build(_lib.ptr(boxes), boxes.shape[0], int(max_depth), int(min_items), float(hit_cost), float(empty_bonus), C.byref(handle)))
        try:
            view = _lib.KDTree()
            _lib.check(L.rsx_kd_info(handle, C.byref(view)))
            nodes, items = _lib.kd_view_to_arrays(view)
            tree = cls(nodes, items, list(view.lower), list(view.upper), view.max_depth)
            tree._min_items, tree._hit_cost, tree._empty_bonus = max(1, int(min_items)), max(1.0, float(hit_cost)), float(empty_bonus)
            need = L.rsx_kd_serialise(handle, int(min_items), float(hit_cost), float(empty_bonus), None, 0)
            buf = np.zeros(need, dtype=np.uint8)
            L.rsx_kd_serialise(handle, int(min_items), float(hit_cost), float(empty_bonus), _lib.ptr(buf), need)
            tree._blob = buf.tobytes()
        finally:
            L.rsx_kd_free(handle)
        return tree

    def serialise(self, *_):
        """KDTree3DCore.save() byte layout (kdtree3d.pyx:864-912)."""
        return self._blob

    @classmethod
    def from_blob(cls, blob):
        """Rebuilds the flattened tree from a KDTree3DCore.save() blob (kdtree3d.pyx:914-990; layout in SURVEY.md Appendix A):
        a tree built by the reference is used as stored, node for node. Returns (tree, bytes consumed)."""
        max_depth, min_items, hit_cost, empty_bonus = struct.unpack_from("<iidd", blob, 0)
        bounds = struct.unpack_from("<6d", blob, 24)
        (n_nodes,) = struct.unpack_from("<i", blob, 72)
        if n_nodes < 0:
            raise ValueError("Corrupt KD-tree record: negative node count.")
        nodes = np.zeros(n_nodes, dtype=_lib.KDNODE_DTYPE)
        raw = nodes.view(np.int32).reshape(n_nodes, 4)
        items = []
        pos = 76
        n_items = 0
        for i in range(n_nodes):
            (kind,) = struct.unpack_from("<i", blob, pos)
            if kind == -1:                                   # leaf: i32 -1, i32 count, i32 items[count]
                (count,) = struct.unpack_from("<i", blob, pos + 4)
                items.append(np.frombuffer(blob, dtype="<i4", count=count, offset=pos + 8))
                raw[i, 0], raw[i, 1], raw[i, 2] = -1, count, n_items
                n_items += count
                pos += 8 + 4 * count
            else:                                            # branch: i32 axis, f64 split, i32 upper child (lower = id + 1)
                split, upper = struct.unpack_from("<di", blob, pos + 4)
                nodes["type"][i], nodes["count"][i], nodes["split"][i] = kind, upper, split
                pos += 16
        tree = cls(nodes, np.concatenate(items).astype(np.int32) if items else np.zeros(0, dtype=np.int32), bounds[:3], bounds[3:], max_depth)
        tree._min_items, tree._hit_cost, tree._empty_bonus = min_items, hit_cost, empty_bonus
        tree._blob = bytes(blob[:pos])
        return tree, pos

    def fill(self, view, keep):
        """Populates a ctypes KDTree struct pointing at this tree's arrays."""
        keep.extend([self.nodes, self.items])
        view.nodes = _lib.ptr(self.nodes)
        view.items = _lib.ptr(self.items)
        view.n_nodes, view.n_items, view.max_depth, view.pad = len(self.nodes), len(self.items), self.max_depth, 0
        for k in range(3):
            view.lower[k], view.upper[k] = self.lower[k], self.upper[k]

    def leaf_items_containing(self, p):
        """Item ids of the leaf containing point p, in leaf order (kdtree3d.pyx:736-792)."""
        if any(p[k] < self.lower[k] or p[k] > self.upper[k] for k in range(3)):
            return []
        i = 0
        while self.nodes["type"][i] >= 0:
            i = i + 1 if p[self.nodes["type"][i]] < self.nodes["split"][i] else int(self.nodes["count"][i])
        first = int(self.nodes[i:i + 1].view(np.int32)[2])
        return [int(v) for v in self.items[first:first + int(self.nodes["count"][i])]]


class Mesh(_DevicePrimitive):
    """raysect/primitive/mesh/mesh.pyx:1048-1397"""

    def __init__(self, vertices=None, triangles=None, normals=None, smoothing=True, closed=True, tolerant=True,
                 flip_normals=False, kdtree_max_depth=-1, kdtree_min_items=1, kdtree_hit_cost=5.0, kdtree_empty_bonus=0.25,
                 parent=None, transform=None, material=None, name=None):
        if vertices is None or triangles is None:
            raise ValueError("Vertices and triangle arrays must be supplied if the mesh is not configured to be an instance.")
        self.data = MeshData(vertices, triangles, normals, smoothing, closed, tolerant, flip_normals,
                             kdtree_max_depth, kdtree_min_items, kdtree_hit_cost, kdtree_empty_bonus)
        super().__init__(parent, transform, material, name)

    def instance(self, parent=None, transform=None, material=None, name=None):   # mesh.pyx:1162-1176
        mesh = Mesh.__new__(Mesh)
        mesh.data = self.data
        Primitive.__init__(mesh, parent, transform, material, name)
        return mesh

    def bounding_box(self):                                 # mesh.pyx:1299-1310
        return self.data.bounding_box(self.to_root())

    def save(self, file):
        self.data.save(file)

    def load(self, file):                                   # mesh.pyx:1320-1339
        """Replaces this mesh's data with the contents of an RSM file."""
        self.data = MeshData.from_file(file)
        self.notify_geometry_change()

    @classmethod
    def from_file(cls, file, parent=None, transform=None, material=None, name=None):   # mesh.pyx:1341-1369
        """Mesh.from_file(file, parent, transform, material, name): instance a mesh from an RSM file."""
        mesh = Mesh.__new__(Mesh)
        mesh.data = MeshData.from_file(file)
        Primitive.__init__(mesh, parent, transform, material, name)
        return mesh
