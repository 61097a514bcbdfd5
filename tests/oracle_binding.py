"""ctypes binding of oracle/liboracle.so (the CPU oracle).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never by the product package ``mlsgpu_amd``.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
# MLSGPU_ORACLE_LIB selects another build of the same sources (bench.py's cpu_baseline: liboracle_native.so, built with
# -O3 -march=native on the machine that times it)
LIB_PATH = os.environ.get("MLSGPU_ORACLE_LIB") or os.path.join(ORACLE_DIR, "liboracle.so")

SPLAT_DTYPE = np.dtype([("position", np.float32, 3), ("radius", np.float32),
                        ("normal", np.float32, 3), ("quality", np.float32)])
assert SPLAT_DTYPE.itemsize == 32

GENERATOR_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_float), C.c_size_t, C.c_void_p)
OUTPUT_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint64),
                        C.POINTER(C.c_uint32), C.c_uint64, C.c_uint64, C.c_uint64)


class Swathe(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("zStride", C.c_uint32),
                ("zBias", C.c_int32), ("zFirst", C.c_uint32), ("zLast", C.c_uint32)]


def build():
    src = os.path.join(ORACLE_DIR, "mlsgpu_oracle.cpp")
    if os.environ.get("MLSGPU_ORACLE_LIB"):
        return
    if (not os.path.exists(LIB_PATH)
            or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(LIB_PATH))):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        u32, i32, u64, f32, vp = C.c_uint32, C.c_int32, C.c_uint64, C.c_float, C.c_void_p
        L.orc_make_code.restype = u32
        L.orc_make_code.argtypes = [C.c_int] * 3
        L.orc_decode.argtypes = [u32, vp]
        L.orc_level_shift.restype = C.c_int
        L.orc_level_shift.argtypes = [vp, vp]
        L.orc_point_box_dist2.restype = f32
        L.orc_point_box_dist2.argtypes = [vp, vp, vp]
        L.orc_solve_quadratic.restype = f32
        L.orc_solve_quadratic.argtypes = [f32, f32, f32]
        L.orc_boundary_factor.restype = f32
        L.orc_boundary_factor.argtypes = [f32]
        L.orc_project_dist_origin_sphere.restype = f32
        L.orc_project_dist_origin_sphere.argtypes = [f32] * 5
        L.orc_fit_sphere.argtypes = [vp, u32, vp]
        L.orc_compute_key.restype = u64
        L.orc_compute_key.argtypes = [vp, vp]
        L.orc_compact_vertices.argtypes = [vp] * 7 + [u64, u64, u64]
        L.orc_make_tables.argtypes = [vp] * 6
        L.orc_compute_max_swathe.restype = u32
        L.orc_compute_max_swathe.argtypes = [u32] * 4
        L.orc_scale_bias.argtypes = [vp, u64, f32, f32, f32, f32]
        L.orc_tree_build.restype = vp
        L.orc_tree_build.argtypes = [vp, u64, u64, vp, vp, u32, u32]
        L.orc_tree_free.argtypes = [vp]
        for name in ("commands", "start"):
            getattr(L, "orc_tree_" + name).restype = vp
            getattr(L, "orc_tree_" + name).argtypes = [vp]
        for name in ("num_commands", "num_start", "commands_size", "start_size"):
            getattr(L, "orc_tree_" + name).restype = u64
            getattr(L, "orc_tree_" + name).argtypes = [vp]
        L.orc_tree_num_levels.restype = u32
        L.orc_tree_num_levels.argtypes = [vp]
        L.orc_process_corners.argtypes = [vp, u64, vp, vp, vp, u32, vp, u32, u32, u32, i32, u32, u32,
                                          f32, C.c_int, vp]
        L.orc_marching_create.restype = vp
        L.orc_marching_create.argtypes = [u32, u32, u32, u32, u64, vp]
        L.orc_marching_free.argtypes = [vp]
        L.orc_marching_generate.restype = C.c_int
        L.orc_marching_generate.argtypes = [vp, GENERATOR_FN, vp, OUTPUT_FN, vp, vp, vp]
        L.orc_marching_stats.argtypes = [vp, vp]
        L.orc_marching_copy_slice.argtypes = [vp, vp, u64, u32, u32, u32, u32, u32]
        L.orc_bucket.restype = C.c_int
        L.orc_bucket.argtypes = [vp, u64, u64, vp, vp, u32, u32, f32, C.c_int, u32, u32, u64,
                                 OUTPUT_FN, vp, vp]
        L.orc_num_threads.restype = C.c_int
        L.orc_bucket_partition.restype = C.c_int
        L.orc_bucket_partition.argtypes = [vp, u64, vp, f32, vp, u64, u32, u32, u32, u64, BUCKET_LEAF_FN, vp, vp]
        L.orc_splat_to_buckets.argtypes = [vp, vp, f32, vp, u32, vp, vp]
        L.orc_for_each_node.restype = C.c_int
        L.orc_for_each_node.argtypes = [vp, u32, vp, vp, C.c_int]
        L.orc_node_child.argtypes = [vp, u32, vp]
        L.orc_choose_micro_size.restype = u32
        L.orc_choose_micro_size.argtypes = [vp, u64, u64, u64, u32]
        _lib = L
    return _lib


BUCKET_LEAF_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_uint64), C.c_uint32, C.c_uint64,
                             C.POINTER(C.c_uint64))


class DensityError(RuntimeError):
    """Bucket::DensityError, src/bucket.h:52-65"""
    def __init__(self, cell_splats):
        super().__init__("Too many splats covering one cell")
        self.cell_splats = cell_splats


def bucket_partition(splats, reference, spacing, extents, max_splats, max_cells, chunk_cells, micro_cells, max_split):
    """Bucket::bucket: list of leaves dict(extents=(x0,x1,y0,y1,z0,z1), chunk, depth, ids) in callback order."""
    assert splats.dtype == SPLAT_DTYPE and splats.flags.c_contiguous
    leaves = []

    def leaf(_user, ext, chunk, depth, n, ids):
        leaves.append(dict(extents=tuple(int(ext[i]) for i in range(6)), chunk=tuple(int(chunk[i]) for i in range(3)),
                           depth=int(depth), ids=np.ctypeslib.as_array(ids, shape=(n,)).copy() if n else np.zeros(0, np.uint64)))
        return 0
    cb = BUCKET_LEAF_FN(leaf)
    ref = np.asarray(reference, np.float32)
    ext = np.asarray(extents, np.int32).reshape(6)
    cell = np.zeros(1, np.uint64)
    rc = lib().orc_bucket_partition(_p(splats), len(splats), _p(ref), float(spacing), _p(ext), max_splats, max_cells,
                                    chunk_cells, micro_cells, max_split, cb, None, _p(cell))
    if rc == 1:
        raise DensityError(int(cell[0]))
    if rc != 0:
        raise ValueError("orc_bucket_partition failed: %d" % rc)
    return leaves


def splat_to_buckets(splat, reference, spacing, extents, bucket_size):
    s = np.zeros(1, SPLAT_DTYPE)
    s[0] = splat
    lower, upper = np.zeros(3, np.int64), np.zeros(3, np.int64)
    lib().orc_splat_to_buckets(_p(s), _p(np.asarray(reference, np.float32)), float(spacing),
                               _p(np.asarray(extents, np.int32).reshape(6)), bucket_size, _p(lower), _p(upper))
    return lower, upper


def for_each_node(dims, levels, inside):
    out = np.zeros((4096, 4), np.uint32)
    n = lib().orc_for_each_node(_p(np.asarray(dims, np.uint32)), levels, _p(np.asarray(inside, np.uint32)), _p(out), 4096)
    return [tuple(int(v) for v in r) for r in out[:n]]


def node_child(node, idx):
    out = np.zeros(4, np.uint32)
    lib().orc_node_child(_p(np.asarray(node, np.uint32)), idx, _p(out))
    return tuple(int(v) for v in out)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _u3(v):
    return np.asarray(v, dtype=np.uint32)


def _i3(v):
    return np.asarray(v, dtype=np.int32)


def make_tables():
    count = np.zeros((256, 2), np.uint8)
    start = np.zeros((257, 2), np.uint16)
    data = np.zeros(8192, np.uint8)
    key = np.zeros((2432, 3), np.uint32)
    sizes = np.zeros(2, np.uint32)
    lib().orc_make_tables(_p(count), _p(start), _p(data), _p(key), _p(sizes[0:1]), _p(sizes[1:2]))
    return count, start, data, key, int(sizes[0]), int(sizes[1])


class Tree:
    """Result of the oracle's SplatTreeCL::enqueueBuild restatement."""

    def __init__(self, splats, first, num, size, offset, subsampling, levels):
        assert splats.dtype == SPLAT_DTYPE and splats.flags.c_contiguous
        self.h = lib().orc_tree_build(_p(splats), first, num, _p(_u3(size)), _p(_i3(offset)), subsampling, levels)
        if not self.h:
            raise ValueError("orc_tree_build failed (size exceeds octree)")
        L = lib()
        n = L.orc_tree_commands_size(self.h)
        self.commands = np.ctypeslib.as_array(C.cast(L.orc_tree_commands(self.h), C.POINTER(C.c_int32)), (n,)).copy()
        n = L.orc_tree_start_size(self.h)
        self.start = np.ctypeslib.as_array(C.cast(L.orc_tree_start(self.h), C.POINTER(C.c_int32)), (n,)).copy()
        self.num_commands = L.orc_tree_num_commands(self.h)
        self.num_start = L.orc_tree_num_start(self.h)
        self.num_levels = L.orc_tree_num_levels(self.h)
        L.orc_tree_free(self.h)
        self.h = None


def process_corners(field, splats, commands, start, subsampling, offset, width, height, zStride, zBias,
                    zFirst, zLast, boundary_factor, shape=0):
    assert field.dtype == np.float32 and field.ndim == 2 and field.flags.c_contiguous
    stats = np.zeros(2, np.uint64)
    commands = np.ascontiguousarray(commands, np.int32)
    start = np.ascontiguousarray(start, np.int32)
    lib().orc_process_corners(_p(field), field.shape[1], _p(splats), _p(commands), _p(start), subsampling,
                              _p(_i3(offset)), width, height, zStride, zBias, zFirst, zLast,
                              boundary_factor, shape, _p(stats))
    return stats


class MeshCollector:
    """Collects the batches handed to Marching's output functor."""

    def __init__(self):
        self.batches = []

        def cb(user, v, k, t, nv, nt, ni):
            nv, nt, ni = int(nv), int(nt), int(ni)
            verts = np.ctypeslib.as_array(v, (nv * 3,)).reshape(nv, 3).copy() if nv else np.zeros((0, 3), np.float32)
            keys = np.ctypeslib.as_array(k, (nv,)).copy() if nv else np.zeros(0, np.uint64)
            tris = np.ctypeslib.as_array(t, (nt * 3,)).reshape(nt, 3).copy() if nt else np.zeros((0, 3), np.uint32)
            self.batches.append(dict(vertices=verts, keys=keys, triangles=tris, num_internal=ni))
        self.cb = OUTPUT_FN(cb)


class MarchingOracle:
    def __init__(self, max_w, max_h, max_d, max_swathe, mesh_memory, alignment):
        self.h = lib().orc_marching_create(max_w, max_h, max_d, max_swathe, mesh_memory, _p(_u3(alignment)))
        if not self.h:
            raise ValueError("invalid Marching parameters")

    def generate(self, gen_fn, size, key_offset=(0, 0, 0)):
        """gen_fn(field2d: np.ndarray, swathe: Swathe) fills the slices zFirst..zLast."""
        out = MeshCollector()

        def gen(user, field, pitch, swp):
            sw = C.cast(swp, C.POINTER(Swathe)).contents
            rows = sw.zStride * (sw.zLast + 1) + sw.zBias + sw.zStride  # generous view
            arr = np.ctypeslib.as_array(field, (rows, pitch))
            gen_fn(arr, sw)
        genc = GENERATOR_FN(gen)
        rc = lib().orc_marching_generate(self.h, genc, None, out.cb, None, _p(_u3(size)), _p(_u3(key_offset)))
        if rc != 0:
            raise ValueError("orc_marching_generate failed: %d" % rc)
        return out.batches

    def stats(self):
        s = np.zeros(7, np.uint64)
        lib().orc_marching_stats(self.h, _p(s))
        return dict(zip(["occupied", "unwelded", "indices", "welded", "external", "shipouts", "overflows"],
                        [int(x) for x in s]))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_marching_free(self.h)
            self.h = None


def bucket(splats, first, num, size, offset, levels=6, subsampling=3, boundary_limit=1.0, shape=0,
           max_cells=255, max_swathe=None, mesh_memory=None):
    """Whole-bucket oracle (DeviceWorkerGroupBase::Worker::operator()). Mutates splats."""
    if max_swathe is None:
        max_swathe = lib().orc_compute_max_swathe(8192, max_cells + 1, 8, 8)
    if mesh_memory is None:
        mesh_memory = max_cells * max_cells * 2 * 872
    out = MeshCollector()
    stats = np.zeros(16, np.uint64)
    rc = lib().orc_bucket(_p(splats), first, num, _p(_u3(size)), _p(_i3(offset)), levels, subsampling,
                          boundary_limit, shape, max_cells, max_swathe, mesh_memory, out.cb, None, _p(stats))
    if rc != 0:
        raise ValueError("orc_bucket failed: %d" % rc)
    names = ["listed", "hits", "occupied", "unwelded", "indices", "welded", "external", "shipouts",
             "overflows", "commands", "tree_us", "mls_us", "marching_us"]
    return out.batches, dict(zip(names, [int(x) for x in stats[:13]]))
