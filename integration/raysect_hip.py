"""
raysect_hip — the module a Raysect maintainer adds to run librsx under STOCK Raysect (plug-point #1).

It imports the real `raysect` package (not source_amd's API mirror) and subclasses `raysect.core.acceleration.Accelerator`
(accelerator.pxd:37-41: cpdef build / hit / contains), so that

    from raysect_hip import HipAccelerator
    world.accelerator = HipAccelerator()          # world.pyx:67-70

makes `World.hit(ray)` / `World.contains(point)` (world.pyx:125-168) answer through the C-ABI of include/rsx.h:
the scenegraph is flattened into an `rsx_scene_desc` (primitive records with Raysect's own to_local / to_root / bounding boxes, the
mesh KD-trees taken verbatim from `MeshData.save()`'s RSM blob, the world tree rebuilt by `rsx_kd_build` with kdtree.pyx:43's
parameters), single rays and points are answered by the host twin (`rsx_host_scene_create`, `rsx_hit_host_one`, `rsx_contains_host` —
SURVEY.md 8b: "World.hit n = 1 -> CPU lib", no GPU needed) and batches by the device scene when a gfx950 device is present
(`rsx_scene_create`, `rsx_hit_batch`).

This file is the code that RAN in the build container against the compiled reference: tests/golden/bind_reference.py installs it over
stock worlds and compares every Intersection field with the stock KDTree accelerator's (log: profiles/r06_bind_reference.txt).
Nothing here is imported by source_amd, the tests on the GPU box or bench.py (raysect is not installed there).
"""
import ctypes as C
import io
import os
import sys

import numpy as np

_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _REPO not in sys.path:
    sys.path.insert(0, _REPO)

from raysect.core import AffineMatrix3D, Intersection, Normal3D, Point3D          # noqa: E402  (stock Raysect)
from raysect.core.acceleration.accelerator import Accelerator                      # noqa: E402
from raysect.primitive import Box, Cylinder, Intersect, Mesh, Sphere, Subtract, Union   # noqa: E402
from raysect.primitive.mesh.mesh import MeshIntersection                          # noqa: E402

from source_amd import _lib                                                        # noqa: E402  (ctypes prototypes of include/rsx.h)
from source_amd.primitive import KDTreeHost, MeshData as RsxMeshData              # noqa: E402  (RSM reader + rsx_kd_build wrapper)

WORLD_KD = dict(max_depth=0, min_items=1, hit_cost=80.0, empty_bonus=0.2)          # kdtree.pyx:43


def _m16(m):
    return [m[i, j] for i in range(4) for j in range(4)]


class StockFlat:
    """rsx_scene_desc of a list of stock Raysect primitives. World primitives first — the index is the hit's
    primitive id — CSG operands behind them, in the spaces Raysect evaluates them in (operands hang under the CSG node's private root,
    csg.pyx:62-100: their to_local() / bounding_box() are already relative to the CSG node)."""

    def __init__(self, primitives):
        self.primitives = list(primitives)
        self.records = [None] * len(self.primitives)
        self.mesh_datas, self._mesh_keys, self._keep = [], [], []
        for i, p in enumerate(self.primitives):
            self.records[i] = self._record(p, i)
        n, self.n_world = len(self.records), len(self.primitives)
        boxes = np.array([r["box"] for r in self.records[:self.n_world]], dtype=np.float64).reshape(-1, 6)
        self.world_kd = KDTreeHost.build(boxes, **WORLD_KD)
        prims = (_lib.Primitive * max(1, n))()
        for i, r in enumerate(self.records):
            c = prims[i]
            c.type, c.material, c.mesh, c.child_a, c.child_b, c.pad = r["type"], r["material"], r["mesh"], r["a"], r["b"], 0
            for k, v in enumerate(r["params"]):
                c.params[k] = v
            for k in range(16):
                c.to_local[k], c.to_root[k] = r["to_local"][k], r["to_root"][k]
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

    def _record(self, p, material=-1):
        box = p.bounding_box()                                # BoundPrimitive.box, boundprimitive.pyx:36-40
        r = dict(obj=p, material=material, mesh=-1, a=-1, b=-1, params=[0.0] * 6, to_local=_m16(p.to_local()), to_root=_m16(p.to_root()),
                 box=[box.lower.x, box.lower.y, box.lower.z, box.upper.x, box.upper.y, box.upper.z])
        if isinstance(p, Sphere):
            r["type"], r["params"][0] = _lib.PRIM_SPHERE, p.radius
        elif isinstance(p, Box):
            r["type"], r["params"] = _lib.PRIM_BOX, [p.lower.x, p.lower.y, p.lower.z, p.upper.x, p.upper.y, p.upper.z]
        elif isinstance(p, Cylinder):
            r["type"], r["params"][0], r["params"][1] = _lib.PRIM_CYLINDER, p.radius, p.height
        elif isinstance(p, Mesh):
            r["type"] = _lib.PRIM_MESH
            for i, key in enumerate(self._mesh_keys):          # Mesh.instance() shares one MeshData: one tree, many records
                if key is p.data:
                    r["mesh"] = i
                    break
            else:
                f = io.BytesIO()
                p.data.save(f)                                 # MeshData.save (mesh.pyx:864-931): vertices, triangles AND Raysect's own KD-tree
                self._mesh_keys.append(p.data)
                self.mesh_datas.append(RsxMeshData.from_file(io.BytesIO(f.getvalue())))
                r["mesh"] = len(self.mesh_datas) - 1
        elif isinstance(p, (Union, Intersect, Subtract)):
            r["type"] = _lib.PRIM_UNION if isinstance(p, Union) else _lib.PRIM_INTERSECT if isinstance(p, Intersect) else _lib.PRIM_SUBTRACT
            for key, child in (("a", p.primitive_a), ("b", p.primitive_b)):
                self.records.append(None)
                slot = len(self.records) - 1
                r[key] = slot
                self.records[slot] = self._record(child)
        else:
            raise NotImplementedError("%s has no librsx lowering (Sphere, Box, Cylinder, Mesh, Union, Intersect, Subtract)" % type(p).__name__)
        return r


class HipAccelerator(Accelerator):
    """world.accelerator = HipAccelerator() — replaces raysect.core.acceleration.KDTree (kdtree.pyx:164-180)."""

    def __init__(self):
        self.flat = None
        self._host = C.c_void_p()
        self._scene = C.c_void_p()
        self._ctx = None

    # Accelerator.build(list primitives) — accelerator.pxd:39
    def build(self, primitives):
        L = _lib.lib()
        self._release()
        self.flat = StockFlat(primitives)
        _lib.check(L.rsx_host_scene_create(C.byref(self.flat.desc), C.byref(self._host)))
        self._io = ((C.c_double * 7)(), (C.c_double * 19)())
        ctx = C.c_void_p()
        if L.rsx_init(0, C.byref(ctx)) == 0:                   # a gfx950 device is present: the same description goes to HBM for batches
            self._ctx = ctx
            _lib.check(L.rsx_scene_create(ctx, C.byref(self.flat.desc), C.byref(self._scene)))

    # Accelerator.hit(Ray) -> Intersection | None — accelerator.pxd:40
    def hit(self, ray):
        i, o = self._io
        og, d = ray.origin, ray.direction
        i[0], i[1], i[2], i[3], i[4], i[5], i[6] = og.x, og.y, og.z, d.x, d.y, d.z, ray.max_distance
        _lib.check(_lib.lib().rsx_hit_host_one(self._host, i, o))
        if o[0] < 0:
            return None
        p = self.flat.primitives[int(o[0])]
        args = (ray, o[1], p, Point3D(o[7], o[8], o[9]), Point3D(o[10], o[11], o[12]), Point3D(o[13], o[14], o[15]),
                Normal3D(o[16], o[17], o[18]), bool(o[2]), p.to_local(), p.to_root())
        if o[3] >= 0:                                          # a mesh surface (also as a CSG operand): MeshIntersection, mesh.pyx:85-135
            return MeshIntersection(*args, int(o[3]), o[4], o[5], o[6])
        return Intersection(*args)

    # Accelerator.contains(Point3D) -> list[Primitive] — accelerator.pxd:41 (in the world tree's leaf order, kdtree.pyx:126-162)
    def contains(self, point):
        n = self.flat.n_world
        if n == 0:
            return []
        inside = (C.c_uint8 * n)()
        pt = (C.c_double * 3)(point.x, point.y, point.z)
        _lib.check(_lib.lib().rsx_contains_host(self._host, 1, pt, inside))
        order = self.flat.world_kd.leaf_items_containing((point.x, point.y, point.z))
        return [self.flat.primitives[j] for j in order if inside[j]]

    # batches (not part of Accelerator's interface): ids / t of many rays at once on the device
    def hit_batch(self, origin, direction, max_distance):
        if not self._scene:
            raise RuntimeError("no gfx950 device: batches need the device scene")
        o, d, m = (np.ascontiguousarray(a, dtype=np.float64) for a in (origin, direction, max_distance))
        n = o.shape[0]
        prim, t = np.empty(n, dtype=np.int32), np.empty(n)
        _lib.check(_lib.lib().rsx_hit_batch(self._scene, n, _lib.ptr(o), _lib.ptr(d), _lib.ptr(m), _lib.ptr(prim), _lib.ptr(t), None, None, None, None))
        return prim, t

    def _release(self):
        L = _lib.lib()
        if self._scene:
            L.rsx_scene_free(self._scene)
            self._scene = C.c_void_p()
        if self._host:
            L.rsx_host_scene_free(self._host)
            self._host = C.c_void_p()

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass
