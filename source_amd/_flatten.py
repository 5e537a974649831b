"""
Scenegraph -> rsx_scene_desc (include/rsx.h): the host-side flattening that replaces the reference's
per-object acceleration wrappers (core/acceleration/kdtree.pyx:43-58 `_PrimitiveKDTree.__init__`,
boundprimitive.pyx:36-40). Pure host work, no GPU needed.

Layout: primitives[0..n_world) are World.primitives in registration order (the hit "primitive id");
CSG operands follow, depth-first, with transforms and boxes relative to their CSG node — exactly the spaces
the reference evaluates them in (csg.pyx:62-100: operands are re-parented to a private CSGRoot).
"""
import numpy as np

from . import _lib
from .primitive import (Box, CSGPrimitive, Cylinder, Intersect, KDTreeHost, Mesh, NullPrimitive, Sphere, Subtract, Union)

WORLD_KD = dict(max_depth=0, min_items=1, hit_cost=80.0, empty_bonus=0.2)   # kdtree.pyx:43


class FlatScene:
    def __init__(self, top_level):
        self.top_level = list(top_level)
        self.records = []          # python-side (object, dict) per flattened primitive
        self.mesh_datas = []
        self._keep = []
        for p in self.top_level:
            self.records.append(None)
        for i, p in enumerate(self.top_level):
            self.records[i] = self._record(p, material=i)
        n = len(self.records)
        self.n_world = len(self.top_level)
        boxes = np.array([r["box"] for r in self.records[:self.n_world]], dtype=np.float64).reshape(-1, 6)
        self.world_kd = KDTreeHost.build(boxes, **WORLD_KD)
        self.boxes = boxes

        # ctypes mirror
        prims = (_lib.Primitive * max(1, n))()
        for i, r in enumerate(self.records):
            c = prims[i]
            c.type, c.material, c.mesh, c.child_a, c.child_b, c.pad = r["type"], r["material"], r["mesh"], r["a"], r["b"], 0
            for k, v in enumerate(r["params"]):
                c.params[k] = v
            for k in range(16):
                c.to_local[k] = r["to_local"][k]
                c.to_root[k] = r["to_root"][k]
            for k in range(3):
                c.box_lower[k], c.box_upper[k] = r["box"][k], r["box"][3 + k]
        meshes = (_lib.MeshData * max(1, len(self.mesh_datas)))()
        for i, md in enumerate(self.mesh_datas):
            m = meshes[i]
            tris = np.ascontiguousarray(md._triangles)
            self._keep.extend([md._vertices, tris, md._face_normals, md._vertex_normals])
            m.vertices, m.triangles, m.face_normals = _lib.ptr(md._vertices), _lib.ptr(tris), _lib.ptr(md._face_normals)
            m.vertex_normals = _lib.ptr(md._vertex_normals)
            m.n_vertices, m.n_triangles = md._vertices.shape[0], tris.shape[0]
            m.n_normals = 0 if md._vertex_normals is None else md._vertex_normals.shape[0]
            m.tri_stride = tris.shape[1] if tris.ndim == 2 and tris.shape[0] else (6 if md._vertex_normals is not None else 3)
            m.smoothing, m.closed = int(md.smoothing), int(md.closed)
            md.kd.fill(m.kd, self._keep)
        desc = _lib.SceneDesc()
        desc.primitives, desc.meshes = prims, meshes
        desc.n_primitives, desc.n_world, desc.n_meshes, desc.pad = n, self.n_world, len(self.mesh_datas), 0
        self.world_kd.fill(desc.world_kd, self._keep)
        self._keep.extend([prims, meshes])
        self.desc = desc
        self.index_of = {id(r["obj"]): i for i, r in enumerate(self.records)}

    def _record(self, p, material=-1):
        box = p.bounding_box().as_list()
        r = dict(obj=p, material=material, mesh=-1, a=-1, b=-1, params=[0.0] * 6,
                 to_local=list(p.to_local().m), to_root=list(p.to_root().m), box=box)
        if isinstance(p, Sphere):
            r["type"], r["params"][0] = _lib.PRIM_SPHERE, p.radius
        elif isinstance(p, Box):
            r["type"] = _lib.PRIM_BOX
            r["params"] = [p.lower.x, p.lower.y, p.lower.z, p.upper.x, p.upper.y, p.upper.z]
        elif isinstance(p, Cylinder):
            r["type"], r["params"][0], r["params"][1] = _lib.PRIM_CYLINDER, p.radius, p.height
        elif isinstance(p, Mesh):
            r["type"] = _lib.PRIM_MESH
            for i, md in enumerate(self.mesh_datas):
                if md is p.data:
                    r["mesh"] = i
                    break
            else:
                self.mesh_datas.append(p.data)
                r["mesh"] = len(self.mesh_datas) - 1
        elif isinstance(p, CSGPrimitive):
            r["type"] = {Union: _lib.PRIM_UNION, Intersect: _lib.PRIM_INTERSECT, Subtract: _lib.PRIM_SUBTRACT}[type(p)]
            for key, child in (("a", p.primitive_a), ("b", p.primitive_b)):
                self.records.append(None)
                slot = len(self.records) - 1
                r[key] = slot
                self.records[slot] = self._record(child)
        elif isinstance(p, NullPrimitive):
            r["type"] = _lib.PRIM_NULL
        else:
            raise NotImplementedError("%s is not on the MI355X hot path (supported: Sphere, Box, Cylinder, Mesh, "
                                      "Union, Intersect, Subtract)" % type(p).__name__)
        return r

    def contains_order(self, point):
        """World KD leaf order for the point (world.contains returns primitives in that order)."""
        return self.world_kd.leaf_items_containing((point.x, point.y, point.z))


def flatten_world(world):
    return FlatScene(world._primitives)
