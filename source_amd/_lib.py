"""
ctypes binding of librsx (include/rsx.h). The library is built in-tree by __graft_entry__.build()
(hipcc --offload-arch=gfx950) into source_amd/lib/librsx.so. There is NO fallback: if the shared library
is missing or fails to load, importing anything that needs it raises immediately.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RSX_LIB") or os.path.join(_HERE, "lib", "librsx.so")   # RSX_LIB: kernel-variant A/B runs (tools/)

f64p = C.POINTER(C.c_double)
f32p = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int32)
u8p = C.POINTER(C.c_uint8)
u64p = C.POINTER(C.c_uint64)


class KDNodeLeaf(C.Structure):
    _fields_ = [("first_item", C.c_int32), ("pad", C.c_int32)]


class KDNodeU(C.Union):
    _fields_ = [("split", C.c_double), ("leaf", KDNodeLeaf)]


class KDNode(C.Structure):
    _fields_ = [("type", C.c_int32), ("count", C.c_int32), ("u", KDNodeU)]


KDNODE_DTYPE = np.dtype([("type", "<i4"), ("count", "<i4"), ("split", "<f8")])   # leaf: first_item = low 4 bytes of split
assert KDNODE_DTYPE.itemsize == C.sizeof(KDNode) == 16


class KDTree(C.Structure):
    _fields_ = [("nodes", C.c_void_p), ("items", C.c_void_p), ("n_nodes", C.c_int32), ("n_items", C.c_int32),
                ("max_depth", C.c_int32), ("pad", C.c_int32), ("lower", C.c_double * 3), ("upper", C.c_double * 3)]


class MeshData(C.Structure):
    _fields_ = [("vertices", C.c_void_p), ("triangles", C.c_void_p), ("vertex_normals", C.c_void_p),
                ("face_normals", C.c_void_p), ("n_vertices", C.c_int32), ("n_triangles", C.c_int32),
                ("n_normals", C.c_int32), ("tri_stride", C.c_int32), ("smoothing", C.c_int32), ("closed", C.c_int32),
                ("kd", KDTree)]


class Primitive(C.Structure):
    _fields_ = [("type", C.c_int32), ("material", C.c_int32), ("mesh", C.c_int32), ("child_a", C.c_int32),
                ("child_b", C.c_int32), ("pad", C.c_int32), ("params", C.c_double * 6), ("to_local", C.c_double * 16),
                ("to_root", C.c_double * 16), ("box_lower", C.c_double * 3), ("box_upper", C.c_double * 3)]


class SceneDesc(C.Structure):
    _fields_ = [("primitives", C.POINTER(Primitive)), ("meshes", C.POINTER(MeshData)), ("n_primitives", C.c_int32),
                ("n_world", C.c_int32), ("n_meshes", C.c_int32), ("pad", C.c_int32), ("world_kd", KDTree)]


class Camera(C.Structure):
    _fields_ = [("nx", C.c_int32), ("ny", C.c_int32), ("image_delta", C.c_double), ("image_start_x", C.c_double),
                ("image_start_y", C.c_double), ("to_root", C.c_double * 16), ("sensitivity", C.c_double)]


class Material(C.Structure):
    _fields_ = [("type", C.c_int32), ("table", C.c_int32), ("scale", C.c_double), ("light_dir", C.c_double * 3)]


class ImportantSphere(C.Structure):
    _fields_ = [("centre", C.c_double * 3), ("radius", C.c_double), ("cdf", C.c_double), ("weight", C.c_double)]


class RenderDesc(C.Structure):
    _fields_ = [("camera", Camera), ("materials", C.POINTER(Material)), ("tables", C.c_void_p), ("tasks", C.c_void_p),
                ("uniforms", C.c_void_p), ("n_tasks", C.c_int64), ("rect", C.c_int32 * 4), ("n_materials", C.c_int32),
                ("n_tables", C.c_int32), ("bins", C.c_int32), ("spp", C.c_int32), ("power", C.c_int32),
                ("rng_mode", C.c_int32), ("seed", C.c_uint64), ("sample_offset", C.c_uint64),
                ("ray_max_depth", C.c_int32), ("ray_extinction_min_depth", C.c_int32), ("ray_extinction_prob", C.c_double),
                ("important", C.POINTER(ImportantSphere)), ("n_important", C.c_int32), ("passes", C.c_int32),
                ("important_path_weight", C.c_double)]


class MT(C.Structure):
    _fields_ = [("mt", C.c_uint64 * 312), ("mti", C.c_int32), ("pad", C.c_int32)]


PRIM_SPHERE, PRIM_BOX, PRIM_CYLINDER, PRIM_MESH, PRIM_UNION, PRIM_INTERSECT, PRIM_SUBTRACT, PRIM_NULL = range(8)
MAT_ABSORBER, MAT_UNIFORM_EMITTER, MAT_DEBUG_LIGHT, MAT_NULL, MAT_UNIFORM_VOLUME_EMITTER, MAT_LAMBERT, MAT_DIELECTRIC = range(7)
RNG_STREAM, RNG_PHILOX = 0, 1

# every symbol include/rsx.h declares: (name, restype, argtypes)
_vp = C.c_void_p
SYMBOLS = [
    ("rsx_init", C.c_int, [C.c_int, C.POINTER(_vp)]),
    ("rsx_free", None, [_vp]),
    ("rsx_last_error", C.c_char_p, []),
    ("rsx_version", C.c_char_p, []),
    ("rsx_set_stream", C.c_int, [_vp, _vp]),
    ("rsx_synchronize", C.c_int, [_vp]),
    ("rsx_idle", C.c_int, [_vp, C.POINTER(C.c_int32)]),
    ("rsx_last_kernel_ms", C.c_int, [_vp, C.POINTER(C.c_float)]),
    ("rsx_last_render_ms", C.c_int, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    ("rsx_render_history", C.c_int, [_vp, C.c_int32, _vp, _vp]),
    ("rsx_selftest_exact_division", C.c_int, [_vp, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]),
    ("rsx_debug_unit_times", C.c_int, [_vp, _vp]),
    ("rsx_render_timeline", C.c_int, [_vp, C.c_int32, _vp]),
    ("rsx_dev_alloc", C.c_int, [_vp, C.c_size_t, C.POINTER(_vp)]),
    ("rsx_dev_free", C.c_int, [_vp, _vp]),
    ("rsx_dev_upload", C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    ("rsx_dev_download", C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    ("rsx_dev_memset", C.c_int, [_vp, _vp, C.c_int, C.c_size_t]),
    ("rsx_kd_build", C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double, C.POINTER(_vp)]),
    ("rsx_kd_info", C.c_int, [_vp, C.POINTER(KDTree)]),
    ("rsx_host_team_size", C.c_int, []),
    ("rsx_kd_free", None, [_vp]),
    ("rsx_kd_serialise", C.c_int64, [_vp, C.c_int32, C.c_double, C.c_double, _vp, C.c_int64]),
    ("rsx_mesh_filter_triangles", C.c_int32, [_vp, _vp, C.c_int32, C.c_int32]),
    ("rsx_mesh_face_normals", C.c_int, [_vp, _vp, C.c_int32, C.c_int32, _vp]),
    ("rsx_mesh_triangle_aabbs", C.c_int, [_vp, _vp, C.c_int32, C.c_int32, _vp]),
    ("rsx_mesh_world_bbox", C.c_int, [_vp, C.c_int32, _vp, _vp]),
    ("rsx_mt_seed_words", None, [C.POINTER(MT), _vp, C.c_uint64]),
    ("rsx_mt_uniform", None, [C.POINTER(MT), C.c_int64, _vp]),
    ("rsx_scene_create", C.c_int, [_vp, C.POINTER(SceneDesc), C.POINTER(_vp)]),
    ("rsx_scene_free", None, [_vp]),
    ("rsx_hit_batch", C.c_int, [_vp, C.c_int64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("rsx_hit_batch_dev", C.c_int, [_vp, C.c_int64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("rsx_roots_batch", C.c_int, [_vp, C.c_int32, C.c_int64, _vp, _vp, _vp, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("rsx_contains_batch", C.c_int, [_vp, C.c_int64, _vp, _vp]),
    ("rsx_host_scene_create", C.c_int, [C.POINTER(SceneDesc), C.POINTER(_vp)]),
    ("rsx_host_scene_free", None, [_vp]),
    ("rsx_hit_host", C.c_int, [_vp, C.c_int64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("rsx_hit_host_one", C.c_int, [_vp, _vp, _vp]),
    ("rsx_contains_host", C.c_int, [_vp, C.c_int64, _vp, _vp]),
    ("rsx_render_pinhole", C.c_int, [_vp, C.POINTER(RenderDesc), _vp, _vp, C.POINTER(C.c_uint64)]),
    ("rsx_render_pinhole_frame", C.c_int, [_vp, C.POINTER(RenderDesc), _vp, _vp, _vp, C.c_int32, C.c_int32, C.POINTER(C.c_uint64)]),
    ("rsx_render_pinhole_xyz", C.c_int, [_vp, C.POINTER(RenderDesc), _vp, C.c_double, _vp, _vp, C.POINTER(C.c_uint64)]),
    ("rsx_frame_combine_dev", C.c_int, [_vp, C.c_int64, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("rsx_selftest_aabb", C.c_int, [_vp, C.c_int64, _vp, _vp, _vp, _vp, _vp, C.POINTER(C.c_uint64)]),
    ("rsx_selftest_camera_rays", C.c_int, [_vp, C.POINTER(RenderDesc), _vp]),
    ("rsx_selftest_welford", C.c_int, [_vp, C.c_int64, C.c_int32, _vp, _vp, _vp]),
    ("rsx_selftest_math", C.c_int, [_vp, C.c_int32, C.c_int64, _vp, _vp, _vp, _vp]),
    ("rsx_comm_unique_id", C.c_int, [_vp]),
    ("rsx_comm_create", C.c_int, [_vp, C.c_int32, C.c_int32, _vp, C.POINTER(_vp)]),
    ("rsx_comm_free", None, [_vp]),
    ("rsx_comm_barrier", C.c_int, [_vp]),
    ("rsx_comm_max_f64", C.c_int, [_vp, C.POINTER(C.c_double)]),
    ("rsx_set_path_stages", C.c_int, [_vp, C.c_int32, C.c_int64]),
    ("rsx_defer_path_checks", C.c_int, [_vp, C.c_int32]),
    ("rsx_collect_path_checks", C.c_int, [_vp, _vp, C.c_int32, _vp, _vp]),
    ("rsx_allgather_frame", C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    ("rsx_allreduce_frame", C.c_int, [_vp, _vp, _vp, _vp, C.c_int64]),
    ("rsx_frame_segment", C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ("rsx_allgather_bins", C.c_int, [_vp, _vp, _vp, _vp, C.c_int64, C.c_int32, _vp]),
    ("rsx_comm_size", C.c_int, [_vp, C.POINTER(C.c_int32)]),
]
# Entry points an A/B build of an earlier revision ($RSX_LIB, tools/ab.sh) may lack: callers test `has(name)` before using them.
# Any other missing symbol is ABI drift between include/rsx.h and the binary and stops the load, whichever library was named.
OPTIONAL_WITH_RSX_LIB = {"rsx_frame_segment", "rsx_allgather_bins", "rsx_comm_size", "rsx_set_path_stages", "rsx_idle", "rsx_host_scene_create", "rsx_host_scene_free", "rsx_hit_host", "rsx_hit_host_one", "rsx_contains_host"}

_lib = None


class RsxError(RuntimeError):
    pass


def lib():
    """Loads librsx.so (once). Raises if it is not built — there is no CPU fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RsxError("librsx.so not found at %s — run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950). source_amd has no CPU fallback." % LIB_PATH)
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")    # (the render lanes are streams that must run side by side: see rsx_init)
        handle = C.CDLL(LIB_PATH)
        for name, restype, argtypes in SYMBOLS:
            if not hasattr(handle, name):
                if os.environ.get("RSX_LIB") and name in OPTIONAL_WITH_RSX_LIB:
                    continue
                raise RsxError("%s does not export %s: the binary and include/rsx.h have drifted apart (rebuild librsx)" % (LIB_PATH, name))
            fn = getattr(handle, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
    return _lib


def has(name):
    """Is an optional entry point (OPTIONAL_WITH_RSX_LIB) present in the loaded library?"""
    return hasattr(lib(), name)


def check(code):
    if code != 0:
        msg = lib().rsx_last_error()
        raise RsxError("librsx error %d: %s" % (code, msg.decode() if msg else "?"))


def ptr(a):
    """numpy array -> void* (None -> NULL). The caller keeps the array alive."""
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def kd_view_to_arrays(view):
    """Copies the nodes/items a KDTree view points at into numpy arrays."""
    nodes = np.ctypeslib.as_array(C.cast(view.nodes, C.POINTER(C.c_uint8)), shape=(view.n_nodes * 16,)).copy().view(KDNODE_DTYPE)
    if view.n_items:
        items = np.ctypeslib.as_array(C.cast(view.items, i32p), shape=(view.n_items,)).copy()
    else:
        items = np.zeros(0, dtype=np.int32)
    return nodes, items
