"""ctypes binding of libmlsgpu_hip.so (include/mlsgpu_hip.h).

Class and method names follow the reference's C++ surface (SplatTreeCL, MlsFunctor, Marching,
DeviceWorkerGroupBase::Worker) so tests read like the reference's own.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

SPLAT_DTYPE = np.dtype([("position", np.float32, 3), ("radius", np.float32),
                        ("normal", np.float32, 3), ("quality", np.float32)])

SHAPE_SPHERE, SHAPE_PLANE = 0, 1


MLS_STATS_WORDS = 43       # MLSGPU_MLS_STATS_WORDS


class MlsError(Exception):
    """Base class of errors raised through the C-ABI."""


class InvalidArgument(MlsError, ValueError):
    """std::invalid_argument in the reference (MLSGPU_ASSERT)."""


class LengthError(MlsError, ValueError):
    """std::length_error in the reference."""


class FormatError(MlsError, ValueError):
    """FastPly::FormatError, src/fast_ply.h:60-75."""


class DensityError(MlsError, RuntimeError):
    """Bucket::DensityError (src/bucket.h:52-65): more than maxSplats splats cover a single cell."""


class HipError(MlsError, RuntimeError):
    """cl::Error in the reference."""


class Swathe(C.Structure):
    """Marching::Swathe, src/marching.h:173-198."""
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("zStride", C.c_uint32),
                ("zBias", C.c_int32), ("zFirst", C.c_uint32), ("zLast", C.c_uint32)]


class Mesh(C.Structure):
    """DeviceKeyMesh, src/mesh.h:101-123."""
    _fields_ = [("dVertices", C.c_void_p), ("dTriangles", C.c_void_p), ("dVertexKeys", C.c_void_p),
                ("numVertices", C.c_uint64), ("numTriangles", C.c_uint64), ("numInternalVertices", C.c_uint64)]


ENQUEUE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(Swathe))
OUTPUT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(Mesh))


class HostMesh(C.Structure):
    """HostKeyMesh, src/mesh.h:125-179 (mlsgpu_host_mesh)."""
    _fields_ = [("vertexKeys", C.c_void_p), ("vertices", C.c_void_p), ("triangles", C.c_void_p),
                ("numVertices", C.c_uint64), ("numTriangles", C.c_uint64), ("numInternalVertices", C.c_uint64)]


class Generator(C.Structure):
    """Marching::Generator, src/marching.h:204-253."""
    _fields_ = [("alignment", C.c_uint32 * 3), ("enqueue", ENQUEUE_FN), ("user", C.c_void_p)]


class WorkerConfig(C.Structure):
    _fields_ = [("maxBucketSplats", C.c_uint64), ("maxCells", C.c_uint32), ("meshMemory", C.c_uint64),
                ("levels", C.c_uint32), ("subsampling", C.c_uint32), ("boundaryLimit", C.c_float),
                ("shape", C.c_int), ("maxSwathe", C.c_uint32), ("gridSpacing", C.c_float),
                ("gridOrigin", C.c_float * 3)]


class SubItem(C.Structure):
    """mlsgpu_subitem: DeviceWorkerGroup::SubItem, src/workers.h:161-168."""
    _fields_ = [("firstSplat", C.c_uint64), ("numSplats", C.c_uint64), ("lowExtent", C.c_int32 * 3),
                ("numVertices", C.c_uint32 * 3), ("dSplats", C.c_void_p)]


class TreeBuild(C.Structure):
    """mlsgpu_tree_build: the arguments of SplatTreeCL::enqueueBuild for one bucket of a batch."""
    _fields_ = [("dSplats", C.c_void_p), ("firstSplat", C.c_uint64), ("numSplats", C.c_uint64), ("size", C.c_uint32 * 3),
                ("offset", C.c_int32 * 3)]


MAX_BATCH = 8

BATCH_OUTPUT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(Mesh))


class GridStruct(C.Structure):
    _fields_ = [("reference", C.c_float * 3), ("spacing", C.c_float), ("extents", C.c_int32 * 6)]


class BucketParams(C.Structure):
    _fields_ = [("maxSplats", C.c_uint64), ("maxCells", C.c_uint32), ("chunkCells", C.c_uint32), ("microCells", C.c_uint32),
                ("maxSplit", C.c_uint64)]


class BucketStruct(C.Structure):
    _fields_ = [("extents", C.c_int32 * 6), ("chunk", C.c_uint64 * 3), ("depth", C.c_uint32), ("numSplats", C.c_uint64),
                ("dIds", C.c_void_p), ("dSplats", C.c_void_p), ("consumed", C.POINTER(C.c_void_p))]


BUCKET_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(BucketStruct))


class FarmConfig(C.Structure):
    _fields_ = [("numDevices", C.c_uint32), ("devices", C.POINTER(C.c_int32)), ("workersPerDevice", C.c_uint32),
                ("spare", C.c_uint32), ("worker", WorkerConfig), ("copyThreads", C.c_uint32), ("stagingBuffers", C.c_uint32)]


FARM_OUTPUT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_uint64, C.c_void_p, C.POINTER(Mesh))
FARM_HOST_OUTPUT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_uint64, C.POINTER(HostMesh))


def library_path():
    return os.path.join(_HERE, "libmlsgpu_hip.so")


_lib = None


def lib():
    """Loads the HIP library; there is deliberately no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(make -C mlsgpu_amd/csrc). mlsgpu_amd has no CPU fallback." % path)
    L = C.CDLL(path)
    u32, i32, u64, f32, vp, sz = C.c_uint32, C.c_int32, C.c_uint64, C.c_float, C.c_void_p, C.c_size_t
    P = C.POINTER

    def sig(name, res, *args):
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = list(args)

    sig("mlsgpu_hip_last_error", C.c_char_p)
    sig("mlsgpu_hip_ctx_create", C.c_int, C.c_int, vp, P(vp))
    sig("mlsgpu_hip_ctx_destroy", None, vp)
    sig("mlsgpu_hip_ctx_stream", vp, vp)
    sig("mlsgpu_hip_ctx_synchronize", C.c_int, vp)
    sig("mlsgpu_hip_ctx_release_scratch", C.c_int, vp)
    sig("mlsgpu_hip_device_count", C.c_int, P(C.c_int))
    sig("mlsgpu_hip_malloc", C.c_int, vp, sz, P(vp))
    sig("mlsgpu_hip_free", C.c_int, vp, vp)
    sig("mlsgpu_hip_host_alloc", C.c_int, sz, P(vp))
    sig("mlsgpu_hip_host_free", C.c_int, vp)
    sig("mlsgpu_hip_memcpy_h2d", C.c_int, vp, vp, vp, sz, C.c_int)
    sig("mlsgpu_hip_memcpy_d2h", C.c_int, vp, vp, vp, sz, C.c_int)
    sig("mlsgpu_hip_memset", C.c_int, vp, vp, C.c_int, sz)
    sig("mlsgpu_hip_memcpy_d2d", C.c_int, vp, vp, vp, sz)
    sig("mlsgpu_hip_ctx_set_timing", C.c_int, vp, C.c_int)
    sig("mlsgpu_hip_ctx_get_stat", C.c_int, vp, C.c_char_p, P(C.c_double), P(u64))
    sig("mlsgpu_hip_ctx_reset_stats", C.c_int, vp)
    sig("mlsgpu_hip_ctx_dump_stats", sz, vp, C.c_char_p, sz)
    sig("mlsgpu_hip_tree_create", C.c_int, vp, u64, u64, P(vp))
    sig("mlsgpu_hip_tree_destroy", None, vp)
    sig("mlsgpu_hip_tree_resource_usage", u64, u64, u64)
    sig("mlsgpu_hip_tree_build", C.c_int, vp, vp, u64, u64, vp, vp, u32)
    sig("mlsgpu_hip_tree_set_mutate", C.c_int, vp, C.c_int)
    sig("mlsgpu_hip_tree_mutates", C.c_int, vp)
    sig("mlsgpu_hip_mls_set_raw_radius", C.c_int, vp, C.c_int)
    sig("mlsgpu_hip_worker_set_keep_splats", C.c_int, vp, C.c_int)
    sig("mlsgpu_hip_tree_clear_splats", None, vp)
    sig("mlsgpu_hip_tree_num_entries", C.c_int, vp, P(u64))
    sig("mlsgpu_hip_tree_splats", vp, vp)
    sig("mlsgpu_hip_tree_commands", vp, vp)
    sig("mlsgpu_hip_tree_start", vp, vp)
    sig("mlsgpu_hip_tree_commands_size", u64, vp)
    sig("mlsgpu_hip_tree_start_size", u64, vp)
    sig("mlsgpu_hip_tree_num_levels", u32, vp)
    sig("mlsgpu_hip_mls_create", C.c_int, vp, C.c_int, P(vp))
    sig("mlsgpu_hip_mls_destroy", None, vp)
    sig("mlsgpu_hip_mls_set", C.c_int, vp, vp, vp, u32)
    sig("mlsgpu_hip_mls_set_buffers", C.c_int, vp, vp, vp, vp, vp, u32)
    sig("mlsgpu_hip_mls_set_boundary_limit", C.c_int, vp, f32)
    sig("mlsgpu_hip_mls_enqueue", C.c_int, vp, vp, u64, u64, P(Swathe))
    sig("mlsgpu_hip_mls_generator", C.c_int, vp, P(Generator))
    sig("mlsgpu_hip_mls_set_variant", C.c_int, vp, C.c_int)
    sig("mlsgpu_hip_mls_set_stats", C.c_int, vp, vp)
    sig("mlsgpu_hip_marching_create", C.c_int, vp, u32, u32, u32, u32, u64, vp, P(vp))
    sig("mlsgpu_hip_marching_destroy", None, vp)
    sig("mlsgpu_hip_marching_resource_usage", u64, u32, u32, u32, u32, u64, vp)
    sig("mlsgpu_hip_marching_generate", C.c_int, vp, P(Generator), OUTPUT_FN, vp, vp, vp)
    sig("mlsgpu_hip_marching_set_vertex_transform", C.c_int, vp, C.c_int, f32, f32, f32, f32)
    sig("mlsgpu_hip_marching_counters", C.c_int, vp, vp)
    sig("mlsgpu_hip_marching_tables", C.c_int, vp, vp, vp, vp, vp)
    sig("mlsgpu_hip_marching_copy_slice", C.c_int, vp, vp, u64, u32, u32, u32, u32, u32)
    sig("mlsgpu_hip_compact_vertices", C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, u64, u64, u64)
    sig("mlsgpu_hip_mesh_host_bytes", u64, P(Mesh))
    sig("mlsgpu_hip_mesh_read", C.c_int, vp, P(Mesh), vp, C.c_int)
    sig("mlsgpu_hip_scale_bias", C.c_int, vp, P(Mesh), f32, f32, f32, f32)
    sig("mlsgpu_hip_mesh_checksum", C.c_int, vp, P(Mesh), vp)
    sig("mlsgpu_hip_worker_create", C.c_int, vp, P(WorkerConfig), P(vp))
    sig("mlsgpu_hip_worker_destroy", None, vp)
    sig("mlsgpu_hip_worker_resource_usage", u64, P(WorkerConfig))
    sig("mlsgpu_hip_worker_resource_usage_lanes", u64, P(WorkerConfig), u32)
    sig("mlsgpu_hip_worker_process", C.c_int, vp, vp, u64, u64, vp, vp, OUTPUT_FN, vp)
    sig("mlsgpu_hip_worker_set_batch", C.c_int, vp, u32)
    sig("mlsgpu_hip_worker_batch", u32, vp)
    sig("mlsgpu_hip_worker_batch_completed", u32, vp)
    sig("mlsgpu_hip_worker_set_marching_group", C.c_int, vp, u32)
    sig("mlsgpu_hip_worker_marching_group", u32, vp)
    sig("mlsgpu_hip_worker_process_batch", C.c_int, vp, vp, P(SubItem), u32, BATCH_OUTPUT_FN, vp)
    sig("mlsgpu_hip_worker_lane_tree", vp, vp, u32)
    sig("mlsgpu_hip_worker_lane_marching", vp, vp, u32)
    sig("mlsgpu_hip_tree_build_batch", C.c_int, P(vp), P(TreeBuild), u32, u32)
    sig("mlsgpu_hip_mls_of_generator", vp, P(Generator))
    sig("mlsgpu_hip_mls_copy_settings", C.c_int, vp, vp)
    sig("mlsgpu_hip_mls_enqueue_batch", C.c_int, P(vp), P(vp), P(u64), P(u64), P(Swathe), u32)
    sig("mlsgpu_hip_marching_generate_batch", C.c_int, P(vp), P(Generator), u32, BATCH_OUTPUT_FN, vp, vp, vp)
    sig("mlsgpu_hip_worker_tree", vp, vp)
    sig("mlsgpu_hip_worker_mls", vp, vp)
    sig("mlsgpu_hip_worker_marching", vp, vp)
    sig("mlsgpu_hip_compute_max_swathe", u32, u32, u32, u32, u32)
    sig("mlsgpu_hip_farm_create", C.c_int, P(FarmConfig), FARM_OUTPUT_FN, vp, P(vp))
    sig("mlsgpu_hip_farm_destroy", None, vp)
    sig("mlsgpu_hip_farm_submit", C.c_int, vp, vp, u64, vp, vp, u64)
    sig("mlsgpu_hip_farm_acquire", C.c_int, vp, u64, P(vp))
    sig("mlsgpu_hip_farm_push", C.c_int, vp, u64, vp, vp, u64)
    sig("mlsgpu_hip_farm_submit_device", C.c_int, vp, C.c_int, vp, vp, u64, P(GridStruct), vp, vp, u64)
    sig("mlsgpu_hip_farm_set_batch", C.c_int, vp, u32)
    sig("mlsgpu_hip_farm_finish", C.c_int, vp)
    sig("mlsgpu_hip_farm_stats", C.c_int, vp, vp)
    sig("mlsgpu_hip_farm_in_flight_max", C.c_int, vp, vp)
    sig("mlsgpu_hip_mesher_write_ply", C.c_int, vp, C.c_uint32, C.c_char_p, vp, C.c_uint32, u64)
    sig("mlsgpu_hip_farm_set_host_output", C.c_int, vp, u64, vp, vp)
    sig("mlsgpu_hip_farm_host_stats", C.c_int, vp, vp)
    sig("mlsgpu_hip_device_node", C.c_int, C.c_int)
    sig("mlsgpu_hip_topology", C.c_int, vp, vp)
    sig("mlsgpu_hip_plan_copy_sides", C.c_int, vp, u32, u32, vp, vp, vp)
    sig("mlsgpu_hip_bind_thread_to_node", C.c_int, C.c_int)
    sig("mlsgpu_hip_test_copy_pool", C.c_int, u32, u32, u64, C.c_int)
    sig("mlsgpu_hip_farm_placement", C.c_int, vp, vp)
    sig("mlsgpu_hip_farm_copy_clock", C.c_int, vp, vp)
    sig("mlsgpu_hip_farm_submit_device_async", C.c_int, vp, C.c_int, vp, vp, u64, vp, vp, vp, u64, vp)
    sig("mlsgpu_hip_farm_worker_clock", C.c_int, vp, vp)
    sig("mlsgpu_hip_farm_group_clock", C.c_int, vp, u32, vp)
    sig("mlsgpu_hip_host_mesher_trim_cache", u64, u64)
    sig("mlsgpu_hip_host_mesher_set_node", C.c_int, vp, C.c_int)
    sig("mlsgpu_hip_host_mesher_set_tmp_dir", C.c_int, vp, C.c_char_p, C.c_uint64)
    sig("mlsgpu_hip_host_mesher_tmp_usage", C.c_int, vp, C.POINTER(C.c_uint64))
    sig("mlsgpu_hip_host_mesher_node", C.c_int, vp)
    sig("mlsgpu_hip_host_mesher_create", C.c_int, P(vp))
    sig("mlsgpu_hip_host_mesher_destroy", None, vp)
    sig("mlsgpu_hip_host_mesher_set_prune_threshold", C.c_int, vp, C.c_double)
    sig("mlsgpu_hip_host_mesher_set_threads", C.c_int, vp, u32)
    sig("mlsgpu_hip_host_mesher_threads", u32, vp)
    sig("mlsgpu_hip_host_mesher_add", C.c_int, vp, u64, P(HostMesh))
    sig("mlsgpu_hip_host_mesher_landing", C.c_int, vp, u64, P(C.c_void_p))
    sig("mlsgpu_hip_host_mesher_landing_pinned", C.c_int, vp)
    sig("mlsgpu_hip_host_mesher_add_landed", C.c_int, vp, u64, P(HostMesh))
    sig("mlsgpu_hip_farm_set_host_landing", C.c_int, vp, vp, vp, vp)
    sig("mlsgpu_hip_host_mesher_farm_landing", C.c_int, vp, u64, P(C.c_void_p))
    sig("mlsgpu_hip_host_mesher_farm_output_landed", C.c_int, vp, C.c_int, u64, P(HostMesh))
    sig("mlsgpu_hip_host_mesher_farm_output", C.c_int, vp, C.c_int, u64, P(HostMesh))
    sig("mlsgpu_hip_host_mesher_finalize", C.c_int, vp, P(u32))
    sig("mlsgpu_hip_host_mesher_boundary", C.c_int, vp, P(u64), P(u64))
    sig("mlsgpu_hip_host_mesher_boundary_read", C.c_int, vp, vp, vp, vp, vp)
    sig("mlsgpu_hip_host_mesher_finalize_with", C.c_int, vp, vp, u64, P(u32))
    sig("mlsgpu_hip_host_mesher_chunk", C.c_int, vp, u32, P(u64), P(u64), P(u64), P(vp), P(vp))
    sig("mlsgpu_hip_host_mesher_stats", C.c_int, vp, vp)
    sig("mlsgpu_hip_transform_splats", None, vp, u64, vp, f32, vp)
    sig("mlsgpu_hip_ply_open", C.c_int, C.c_char_p, f32, f32, P(vp))
    sig("mlsgpu_hip_ply_close", None, vp)
    sig("mlsgpu_hip_ply_size", u64, vp)
    sig("mlsgpu_hip_ply_layout", C.c_int, vp, vp)
    sig("mlsgpu_hip_ply_read", C.c_int, vp, u64, u64, vp)
    sig("mlsgpu_hip_ply_load", C.c_int, vp, vp, u64, u64, vp, u32)
    sig("mlsgpu_hip_fileset_create", C.c_int, f32, f32, P(vp))
    sig("mlsgpu_hip_fileset_destroy", None, vp)
    sig("mlsgpu_hip_fileset_add_file", C.c_int, vp, C.c_char_p)
    sig("mlsgpu_hip_fileset_num_files", u64, vp)
    sig("mlsgpu_hip_fileset_num_splats", u64, vp)
    sig("mlsgpu_hip_fileset_set_buffer_size", C.c_int, vp, u64)
    sig("mlsgpu_hip_fileset_read", C.c_int, vp, u64, u64, vp)
    sig("mlsgpu_hip_fileset_load", C.c_int, vp, vp, u64, u64, vp, u32)
    sig("mlsgpu_hip_mesher_create", C.c_int, vp, P(vp))
    sig("mlsgpu_hip_mesher_destroy", None, vp)
    sig("mlsgpu_hip_mesher_set_prune_threshold", C.c_int, vp, C.c_double)
    sig("mlsgpu_hip_mesher_set_background", C.c_int, vp, C.c_int)
    sig("mlsgpu_hip_mesher_add", C.c_int, vp, vp, u64, vp)
    sig("mlsgpu_hip_mesher_reserve", C.c_int, vp, u64, u64, u64)
    sig("mlsgpu_hip_mesher_farm_output", C.c_int, vp, C.c_int, u64, vp, P(Mesh))
    sig("mlsgpu_hip_mesher_reset", C.c_int, vp)
    sig("mlsgpu_hip_mesher_boundary", C.c_int, vp, P(u64), P(u64))
    sig("mlsgpu_hip_mesher_boundary_read", C.c_int, vp, vp, vp, vp, vp)
    sig("mlsgpu_hip_mesher_finalize_with", C.c_int, vp, vp, u64, P(u32))
    sig("mlsgpu_hip_mesher_finalize", C.c_int, vp, P(u32))
    sig("mlsgpu_hip_mesher_chunk", C.c_int, vp, u32, P(u64), P(u64), P(u64), P(vp), P(vp))
    sig("mlsgpu_hip_mesher_stats", C.c_int, vp, vp)
    sig("mlsgpu_hip_write_ply", C.c_int, C.c_char_p, vp, u64, vp, u64, vp, u32)
    sig("mlsgpu_hip_bucket", C.c_int, vp, vp, u64, P(GridStruct), P(BucketParams), BUCKET_FN, vp, P(u64))
    sig("mlsgpu_hip_bucket_load", C.c_int, vp, vp, vp, u64, P(GridStruct), vp)
    sig("mlsgpu_hip_fileset_bounding_grid", C.c_int, vp, vp, C.c_float, C.c_uint32, u64, C.c_uint32, P(GridStruct))
    sig("mlsgpu_hip_bucket_stream", C.c_int, vp, vp, P(GridStruct), P(BucketParams), u64, u64, C.c_uint32, BUCKET_FN, vp, vp, vp)
    sig("mlsgpu_hip_bounding_grid", C.c_int, vp, vp, u64, f32, u32, P(GridStruct))
    sig("mlsgpu_hip_test_make_code", C.c_int, vp, C.c_int, C.c_int, C.c_int, P(u32))
    sig("mlsgpu_hip_test_level_shift", C.c_int, vp, vp, vp, P(i32))
    sig("mlsgpu_hip_test_point_box_dist2", C.c_int, vp, vp, vp, vp, P(f32))
    sig("mlsgpu_hip_test_solve_quadratic", C.c_int, vp, f32, f32, f32, P(f32))
    sig("mlsgpu_hip_test_fit_sphere", C.c_int, vp, vp, u32, vp)
    sig("mlsgpu_hip_test_compute_key", C.c_int, vp, vp, vp, P(u64))
    sig("mlsgpu_hip_test_scan_u32", C.c_int, vp, vp, u64, u32)
    sig("mlsgpu_hip_test_scan_u32_batch", C.c_int, vp, vp, vp, vp, vp, u32, u32)
    sig("mlsgpu_hip_test_sort_u32", C.c_int, vp, vp, vp, u64, u32)
    sig("mlsgpu_hip_test_sort_u64", C.c_int, vp, vp, vp, u64, u32)
    _lib = L
    return L


_ERRORS = {1: InvalidArgument, 2: LengthError, 3: HipError, 4: HipError, 5: MlsError, 6: DensityError, 7: FormatError}


def check(rc):
    if rc != 0:
        msg = lib().mlsgpu_hip_last_error().decode("utf-8", "replace")
        raise _ERRORS.get(rc, MlsError)("[%d] %s" % (rc, msg))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _u3(v):
    return np.ascontiguousarray(v, dtype=np.uint32)


def _i3(v):
    return np.ascontiguousarray(v, dtype=np.int32)


class Context:
    """One device + one in-order stream (the cl::Context + cl::CommandQueue of one worker)."""

    def __init__(self, device=0, stream=None):
        h = C.c_void_p()
        check(lib().mlsgpu_hip_ctx_create(device, stream, C.byref(h)))
        self.h = h
        self.device = device

    @property
    def stream(self):
        return lib().mlsgpu_hip_ctx_stream(self.h)

    def synchronize(self):
        check(lib().mlsgpu_hip_ctx_synchronize(self.h))

    def release_scratch(self):
        """Hands back the device scratch calls on this context keep from call to call (the bucketer's lists and counters)."""
        check(lib().mlsgpu_hip_ctx_release_scratch(self.h))

    def set_timing(self, enabled):
        check(lib().mlsgpu_hip_ctx_set_timing(self.h, int(enabled)))

    def reset_stats(self):
        check(lib().mlsgpu_hip_ctx_reset_stats(self.h))

    def stat(self, name):
        ms, n = C.c_double(), C.c_uint64()
        check(lib().mlsgpu_hip_ctx_get_stat(self.h, name.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def stats(self):
        n = lib().mlsgpu_hip_ctx_dump_stats(self.h, None, 0)
        buf = C.create_string_buffer(n + 16)
        lib().mlsgpu_hip_ctx_dump_stats(self.h, buf, n + 16)
        out = {}
        for line in buf.value.decode().splitlines():
            name, ms, cnt = line.rsplit(" ", 2)
            out[name] = (float(ms), int(cnt))
        return out

    def close(self):
        if self.h:
            lib().mlsgpu_hip_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceBuffer:
    """A device allocation (cl::Buffer)."""

    def __init__(self, ctx, nbytes=0, array=None, fill=None, borrow=None):
        self.ctx = ctx
        self.owned = borrow is None
        if borrow is not None:
            # memory that someone else owns (e.g. a torch tensor's data_ptr() on the same device); never freed here
            self.ptr, self.nbytes = int(borrow), int(nbytes)
            return
        if array is not None:
            array = np.ascontiguousarray(array)
            nbytes = array.nbytes
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        check(lib().mlsgpu_hip_malloc(ctx.h, self.nbytes, C.byref(p)))
        self.ptr = p.value
        if fill is not None:
            # pattern fill like createBuffer's 0xDEADBEEF (test/test_clh.cpp:88-99)
            pat = np.full((self.nbytes + 3) // 4, fill, np.uint32)
            check(lib().mlsgpu_hip_memcpy_h2d(ctx.h, self.ptr, _p(pat), self.nbytes, 0))
        if array is not None:
            self.upload(array)

    def upload(self, array, offset=0):
        array = np.ascontiguousarray(array)
        assert offset + array.nbytes <= self.nbytes
        check(lib().mlsgpu_hip_memcpy_h2d(self.ctx.h, self.ptr + offset, _p(array), array.nbytes, 0))

    def download(self, dtype, count=None, offset=0):
        dtype = np.dtype(dtype)
        if count is None:
            count = (self.nbytes - offset) // dtype.itemsize
        out = np.empty(count, dtype)
        check(lib().mlsgpu_hip_memcpy_d2h(self.ctx.h, _p(out), self.ptr + offset, out.nbytes, 0))
        return out

    def copy_from(self, other, nbytes=None):
        """Device-to-device copy, asynchronous on the context's stream."""
        check(lib().mlsgpu_hip_memcpy_d2d(self.ctx.h, self.ptr, other.ptr, self.nbytes if nbytes is None else nbytes))

    def free(self):
        if self.ptr and self.owned:
            lib().mlsgpu_hip_free(self.ctx.h, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class PinnedBuffer:
    """Page-locked host memory (CLH::PinnedMemory, src/clh.h:334-477) that grows on demand."""

    def __init__(self, nbytes):
        self.ptr, self.nbytes = None, 0
        self.ensure(nbytes)

    def ensure(self, nbytes):
        if nbytes <= self.nbytes:
            return
        self.free()
        p = C.c_void_p()
        check(lib().mlsgpu_hip_host_alloc(int(nbytes), C.byref(p)))
        self.ptr, self.nbytes = p.value, int(nbytes)

    def free(self):
        if self.ptr:
            lib().mlsgpu_hip_host_free(self.ptr)
        self.ptr, self.nbytes = None, 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def download_into_pinned(ctx, chunk, pinned):
    """Reads a device chunk of Mesher.chunk(i, download=False) into `pinned` (vertices then triangles), asynchronously
    on ctx's stream; returns the bytes.  The caller synchronises."""
    nv, nt = 12 * chunk["num_vertices"], 12 * chunk["num_triangles"]
    if nv + nt > pinned.nbytes:
        ctx.synchronize()
        pinned.ensure(int((nv + nt) * 1.1) + 4096)
    # in pieces: one copy command of hundreds of MB holds its DMA engine until it is done, and the host-to-device copies of
    # the NEXT job's splats wait behind it; pieces let them interleave (what the ring route's 23 MB read-backs do by nature)
    piece = int(os.environ.get("MLSGPU_HIP_READBACK_PIECE_MB", "32")) << 20
    for base, src, n in ((0, chunk["d_vertices"], nv), (nv, chunk["d_triangles"], nt)):
        o = 0
        while o < n:
            k = min(piece, n - o) if piece > 0 else n
            check(lib().mlsgpu_hip_memcpy_d2h(ctx.h, pinned.ptr + base + o, src + o, k, 1))
            o += k
    return nv + nt


def download_ptr(ctx, ptr, dtype, count):
    out = np.empty(count, np.dtype(dtype))
    if count:
        check(lib().mlsgpu_hip_memcpy_d2h(ctx.h, _p(out), ptr, out.nbytes, 0))
    return out


class SplatTree:
    """SplatTreeCL, src/splat_tree_cl.h:216-296."""

    def __init__(self, ctx, max_levels, max_splats):
        self.ctx = ctx
        h = C.c_void_p()
        check(lib().mlsgpu_hip_tree_create(ctx.h, max_levels, max_splats, C.byref(h)))
        self.h = h

    def enqueue_build(self, splats, first_splat, num_splats, size, offset, subsampling_shift):
        check(lib().mlsgpu_hip_tree_build(self.h, splats.ptr, first_splat, num_splats, _p(_u3(size)),
                                          _p(_i3(offset)), subsampling_shift))

    def set_mutate(self, mutate):
        """False: builds leave the splats untouched (the radius stays the radius); MlsFunctor.set() follows."""
        check(lib().mlsgpu_hip_tree_set_mutate(self.h, 1 if mutate else 0))

    def clear_splats(self):
        lib().mlsgpu_hip_tree_clear_splats(self.h)

    @property
    def num_levels(self):
        return lib().mlsgpu_hip_tree_num_levels(self.h)

    def commands(self):
        n = lib().mlsgpu_hip_tree_commands_size(self.h)
        return download_ptr(self.ctx, lib().mlsgpu_hip_tree_commands(self.h), np.int32, n)

    def start(self):
        n = lib().mlsgpu_hip_tree_start_size(self.h)
        return download_ptr(self.ctx, lib().mlsgpu_hip_tree_start(self.h), np.int32, n)

    def __del__(self):
        if getattr(self, "h", None):
            lib().mlsgpu_hip_tree_destroy(self.h)
            self.h = None


class MlsFunctor:
    """MlsFunctor, src/mls.h:79-170."""
    wgs = (8, 8, 8)
    subsampling_min = 3

    def __init__(self, ctx, shape=SHAPE_SPHERE):
        self.ctx = ctx
        h = C.c_void_p()
        check(lib().mlsgpu_hip_mls_create(ctx.h, shape, C.byref(h)))
        self.h = h

    def set(self, offset, tree, subsampling_shift):
        check(lib().mlsgpu_hip_mls_set(self.h, _p(_i3(offset)), tree.h, subsampling_shift))

    def set_buffers(self, offset, splats, commands, start, subsampling_shift):
        check(lib().mlsgpu_hip_mls_set_buffers(self.h, _p(_i3(offset)), splats.ptr, commands.ptr, start.ptr,
                                               subsampling_shift))

    def set_boundary_limit(self, limit):
        check(lib().mlsgpu_hip_mls_set_boundary_limit(self.h, limit))

    def set_raw_radius(self, raw):
        """With set_buffers: True if the splats' radius slot still holds the radius (not 1/r^2)."""
        check(lib().mlsgpu_hip_mls_set_raw_radius(self.h, 1 if raw else 0))

    def set_variant(self, variant):
        check(lib().mlsgpu_hip_mls_set_variant(self.h, variant))

    def set_stats(self, counters):
        """counters: DeviceBuffer of MLS_STATS_WORDS uint64 (or None): see mlsgpu_hip_mls_set_stats."""
        check(lib().mlsgpu_hip_mls_set_stats(self.h, counters.ptr if counters else None))

    def alignment(self):
        return self.wgs

    def enqueue(self, field, pitch, rows, swathe):
        check(lib().mlsgpu_hip_mls_enqueue(self.h, field.ptr, pitch, rows, C.byref(swathe)))

    def generator(self):
        g = Generator()
        check(lib().mlsgpu_hip_mls_generator(self.h, C.byref(g)))
        return g

    def __del__(self):
        if getattr(self, "h", None):
            lib().mlsgpu_hip_mls_destroy(self.h)
            self.h = None


def read_mesh(ctx, mesh):
    """enqueueReadMesh into a HostKeyMesh blob, returned as numpy views (src/mesh.cpp:51-102)."""
    nbytes = lib().mlsgpu_hip_mesh_host_bytes(C.byref(mesh))
    nv, nt, ni = int(mesh.numVertices), int(mesh.numTriangles), int(mesh.numInternalVertices)
    blob = np.zeros(max(nbytes // 8 + 1, 1), np.uint64)
    if nv or nt:
        check(lib().mlsgpu_hip_mesh_read(ctx.h, C.byref(mesh), _p(blob), 0))
    raw = blob.view(np.uint8)
    ne = nv - ni
    keys = raw[:8 * ne].view(np.uint64).copy()
    verts = raw[8 * ne:8 * ne + 12 * nv].view(np.float32).reshape(nv, 3).copy()
    tris = raw[8 * ne + 12 * nv:8 * ne + 12 * nv + 12 * nt].view(np.uint32).reshape(nt, 3).copy()
    full_keys = np.zeros(nv, np.uint64)
    full_keys[ni:] = keys
    return dict(vertices=verts, keys=full_keys, triangles=tris, num_internal=ni)


class MeshCollector:
    """An output functor that reads every batch back to the host."""

    def __init__(self, ctx, scale_bias=None):
        self.ctx = ctx
        self.batches = []
        self.error = None

        def cb(user, stream, meshp):
            try:
                mesh = meshp.contents
                if scale_bias is not None:
                    check(lib().mlsgpu_hip_scale_bias(ctx.h, meshp, *scale_bias))
                self.batches.append(read_mesh(ctx, mesh))
                return 0
            except Exception as e:   # never let an exception cross the C boundary
                self.error = e
                return 1
        self.cb = OUTPUT_FN(cb)


class SizeCollector:
    """An output functor that only records mesh sizes (bench: outputs stay in HBM)."""

    def __init__(self):
        self.vertices = self.triangles = self.external = self.batches = 0

        def cb(user, stream, meshp):
            m = meshp.contents
            self.vertices += m.numVertices
            self.triangles += m.numTriangles
            self.external += m.numVertices - m.numInternalVertices
            self.batches += 1
            return 0
        self.cb = OUTPUT_FN(cb)


def words_checksum(words):
    """The host twin of mlsgpu_hip_mesh_checksum for one array: sum of word[i] * (2 i + 1) modulo 2^64."""
    w = np.ascontiguousarray(words).view(np.uint32).ravel().astype(np.uint64)
    with np.errstate(over="ignore"):
        return int((w * (np.arange(len(w), dtype=np.uint64) * np.uint64(2) + np.uint64(1))).sum(dtype=np.uint64))


def batch_checksum(batch):
    """(vertices, triangles, external keys) checksums of a host batch, as mlsgpu_hip_mesh_checksum computes them."""
    ni = batch["num_internal"]
    return (words_checksum(batch["vertices"]), words_checksum(batch["triangles"]), words_checksum(batch["keys"][ni:]))


class ChecksumCollector:
    """An output functor that folds every ship-out into a digest on the device (nothing is copied to the host):
    sizes plus the three checksums of every batch, combined in ship-out order."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.vertices = self.triangles = self.external = self.batches = 0
        self.sums = []
        self.error = None

        def cb(user, stream, meshp):
            try:
                mm = meshp.contents
                out = np.zeros(3, np.uint64)
                check(lib().mlsgpu_hip_mesh_checksum(ctx.h, meshp, _p(out)))
                self.sums.append((int(mm.numVertices), int(mm.numTriangles), int(mm.numInternalVertices)) + tuple(int(x) for x in out))
                self.vertices += mm.numVertices
                self.triangles += mm.numTriangles
                self.external += mm.numVertices - mm.numInternalVertices
                self.batches += 1
                return 0
            except Exception as e:   # never let an exception cross the C boundary
                self.error = e
                return 1
        self.cb = OUTPUT_FN(cb)

    def digest(self):
        return digest_of_sums(self.sums)


class ExternalCollector(ChecksumCollector):
    """ChecksumCollector that also copies every batch's EXTERNAL vertices and their keys to the host (a few percent of
    a mesh): what a cross-bucket weld check needs at sizes where whole meshes are too big to read back."""

    def __init__(self, ctx):
        super().__init__(ctx)
        self.ext_keys, self.ext_vertices = [], []
        inner = self.cb

        def cb(user, stream, meshp):
            rc = inner(user, stream, meshp)
            if rc != 0:
                return rc
            try:
                mm = meshp.contents
                ni, ne = int(mm.numInternalVertices), int(mm.numVertices - mm.numInternalVertices)
                if ne:
                    self.ext_keys.append(download_ptr(ctx, mm.dVertexKeys + 8 * ni, np.uint64, ne))
                    self.ext_vertices.append(download_ptr(ctx, mm.dVertices + 12 * ni, np.float32, 3 * ne).reshape(-1, 3))
                return 0
            except Exception as e:
                self.error = e
                return 1
        self.cb = OUTPUT_FN(cb)


def digest_of_sums(sums):
    import hashlib
    h = hashlib.sha256()
    for rec in sums:
        h.update(np.array(rec, np.uint64).tobytes())
    return h.hexdigest()[:16]


class Mesher:
    """Device-resident mesh sink: OOCMesher's weld / components / prune / chunks (src/mesher.h:203-330)."""

    def __init__(self, ctx, prune_threshold=0.0):
        self.ctx = ctx
        h = C.c_void_p()
        check(lib().mlsgpu_hip_mesher_create(ctx.h, C.byref(h)))
        self.h = h
        if prune_threshold:
            self.set_prune_threshold(prune_threshold)

    def set_prune_threshold(self, threshold):
        check(lib().mlsgpu_hip_mesher_set_prune_threshold(self.h, float(threshold)))

    def set_background(self, on=True):
        """finalize runs beside other GPU work (the next job's buckets) and holds back: mlsgpu_hip_mesher_set_background."""
        check(lib().mlsgpu_hip_mesher_set_background(self.h, 1 if on else 0))

    def reserve(self, num_vertices, num_triangles, num_external):
        check(lib().mlsgpu_hip_mesher_reserve(self.h, num_vertices, num_triangles, num_external))

    def add_device(self, from_ctx, chunk_id, mesh_ptr):
        check(lib().mlsgpu_hip_mesher_add(self.h, from_ctx.h, chunk_id, mesh_ptr))

    def add(self, chunk_id, vertices, num_internal, keys, triangles, ctx=None):
        """Adds a host mesh (tests): uploads it as a DeviceKeyMesh first; keys are the external vertices' keys."""
        ctx = ctx or self.ctx
        vertices = np.ascontiguousarray(vertices, np.float32).reshape(-1, 3)
        triangles = np.ascontiguousarray(triangles, np.uint32).reshape(-1, 3)
        all_keys = np.zeros(len(vertices), np.uint64)
        all_keys[num_internal:] = np.asarray(keys, np.uint64)
        bufs = [DeviceBuffer(ctx, array=a) if a.size else None for a in (vertices, triangles, all_keys)]
        mesh = Mesh(bufs[0].ptr if bufs[0] else None, bufs[1].ptr if bufs[1] else None, bufs[2].ptr if bufs[2] else None,
                    len(vertices), len(triangles), num_internal)
        try:
            check(lib().mlsgpu_hip_mesher_add(self.h, ctx.h, chunk_id, C.byref(mesh)))
        finally:
            for b in bufs:
                if b is not None:
                    b.free()

    def collector(self, from_ctx, chunk_id):
        """An output functor for Worker.process / Marching.generate that feeds this mesher."""
        return MesherCollector(self, from_ctx, chunk_id)

    def reset(self):
        check(lib().mlsgpu_hip_mesher_reset(self.h))

    def boundary(self):
        """(keys, key_root, root_vertices, root_triangles) of what has been added: what a cross-rank merge needs
        (dist_sink.global_prune); the meshes stay in HBM."""
        nk, nr = C.c_uint64(), C.c_uint64()
        check(lib().mlsgpu_hip_mesher_boundary(self.h, C.byref(nk), C.byref(nr)))
        keys, kr = np.zeros(nk.value, np.uint64), np.zeros(nk.value, np.uint32)
        rv, rt = np.zeros(nr.value, np.uint64), np.zeros(nr.value, np.uint64)
        check(lib().mlsgpu_hip_mesher_boundary_read(self.h, _p(keys), _p(kr), _p(rv), _p(rt)))
        return keys, kr, rv, rt

    def finalize_with(self, keep_root):
        keep = np.ascontiguousarray(keep_root, np.uint8)
        n = C.c_uint32(0)
        check(lib().mlsgpu_hip_mesher_finalize_with(self.h, _p(keep), len(keep), C.byref(n)))
        return n.value

    def finalize(self):
        n = C.c_uint32(0)
        check(lib().mlsgpu_hip_mesher_finalize(self.h, C.byref(n)))
        return n.value

    def chunk(self, i, download=True):
        cid, nv, nt = C.c_uint64(), C.c_uint64(), C.c_uint64()
        pv, pt = C.c_void_p(), C.c_void_p()
        check(lib().mlsgpu_hip_mesher_chunk(self.h, i, C.byref(cid), C.byref(nv), C.byref(nt), C.byref(pv), C.byref(pt)))
        out = dict(chunk=cid.value, num_vertices=nv.value, num_triangles=nt.value, d_vertices=pv.value, d_triangles=pt.value)
        if download:
            v = np.empty((nv.value, 3), np.float32)
            t = np.empty((nt.value, 3), np.uint32)
            if v.size:
                check(lib().mlsgpu_hip_memcpy_d2h(self.ctx.h, _p(v), pv.value, v.nbytes, 0))
            if t.size:
                check(lib().mlsgpu_hip_memcpy_d2h(self.ctx.h, _p(t), pt.value, t.nbytes, 0))
            out["vertices"], out["triangles"] = v, t
        return out

    def write_ply(self, i, path, comments=(), buffer_bytes=0):
        """Output chunk i straight from HBM into FastPly::Writer's file through a bounded pinned buffer."""
        arr = (C.c_char_p * max(len(comments), 1))(*[c.encode("ascii") for c in comments])
        check(lib().mlsgpu_hip_mesher_write_ply(self.h, i, str(path).encode(), C.cast(arr, C.c_void_p) if comments else None,
                                                len(comments), buffer_bytes))

    def stats(self):
        out = np.zeros(8, np.uint64)
        check(lib().mlsgpu_hip_mesher_stats(self.h, _p(out)))
        names = ["total_vertices", "threshold", "components", "kept_components", "kept_vertices", "kept_triangles",
                 "vertices_added", "triangles_added"]
        return dict(zip(names, [int(x) for x in out]))

    def close(self):
        if getattr(self, "h", None):
            lib().mlsgpu_hip_mesher_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MesherCollector:
    def __init__(self, mesher, from_ctx, chunk_id):
        self.error = None
        self.batches = 0

        def cb(user, stream, meshp):
            try:
                mesher.add_device(from_ctx, chunk_id, meshp)
                self.batches += 1
                return 0
            except Exception as e:      # noqa: BLE001
                self.error = e
                return 1
        self.cb = OUTPUT_FN(cb)


def host_mesh_arrays(hm):
    """numpy COPIES of a mlsgpu_host_mesh (the memory behind it is only valid during the callback)."""
    nv, nt, ni = int(hm.numVertices), int(hm.numTriangles), int(hm.numInternalVertices)

    def view(ptr, dtype, count):
        if count == 0:
            return np.zeros(0, dtype)
        buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype, count=count).copy()
    keys = np.zeros(nv, np.uint64)
    keys[ni:] = view(hm.vertexKeys, np.uint64, nv - ni)
    return dict(vertices=view(hm.vertices, np.float32, 3 * nv).reshape(nv, 3), keys=keys,
                triangles=view(hm.triangles, np.uint32, 3 * nt).reshape(nt, 3), num_internal=ni)


class HostMesher:
    """OOCMesher's weld on the host (src/mesher.cpp:220-469), in memory: the cross-GPU welder behind
    BucketFarm.set_host_output (include/mlsgpu_hip.h, "host mesh sink")."""

    def __init__(self, prune_threshold=0.0, threads=0):
        h = C.c_void_p()
        check(lib().mlsgpu_hip_host_mesher_create(C.byref(h)))
        self.h = h
        check(lib().mlsgpu_hip_host_mesher_set_prune_threshold(self.h, prune_threshold))
        if threads:
            check(lib().mlsgpu_hip_host_mesher_set_threads(self.h, threads))

    def threads(self):
        return lib().mlsgpu_hip_host_mesher_threads(self.h)

    def set_node(self, node):
        """Bind the welder's threads to a NUMA node's CPUs (before the first add); -1 = unbound."""
        check(lib().mlsgpu_hip_host_mesher_set_node(self.h, int(node)))

    def set_tmp_dir(self, path, resident_bytes=0):
        """Bounded-memory mode (OOCMesher's temporary files): the welder's memory is file-backed, blocks beyond
        `resident_bytes` are written out and dropped once welded.  Before the first add."""
        check(lib().mlsgpu_hip_host_mesher_set_tmp_dir(self.h, None if path is None else str(path).encode(), int(resident_bytes)))

    def tmp_usage(self):
        out = (C.c_uint64 * 3)()
        check(lib().mlsgpu_hip_host_mesher_tmp_usage(self.h, out))
        return {"mapped": int(out[0]), "resident": int(out[1]), "paged_out": int(out[2])}

    def add(self, chunk_id, vertices, num_internal, keys, triangles):
        """keys: the external vertices' keys (len(vertices) - num_internal of them)."""
        v = np.ascontiguousarray(vertices, np.float32).reshape(-1, 3)
        t = np.ascontiguousarray(triangles, np.uint32).reshape(-1, 3)
        k = np.ascontiguousarray(keys, np.uint64)
        assert len(k) == len(v) - num_internal
        hm = HostMesh(k.ctypes.data, v.ctypes.data, t.ctypes.data, len(v), len(t), num_internal)
        check(lib().mlsgpu_hip_host_mesher_add(self.h, chunk_id, C.byref(hm)))

    def add_landed(self, chunk_id, vertices, num_internal, keys, triangles):
        """The same mesh through the in-place route: room from mlsgpu_hip_host_mesher_landing (what a farm's read-back lands
        in), filled here, adopted without a copy by mlsgpu_hip_host_mesher_add_landed."""
        v = np.ascontiguousarray(vertices, np.float32).reshape(-1, 3)
        t = np.ascontiguousarray(triangles, np.uint32).reshape(-1, 3)
        k = np.ascontiguousarray(keys, np.uint64)
        assert len(k) == len(v) - num_internal
        need = (k.nbytes + v.nbytes + t.nbytes + 63) & ~63
        ptr = C.c_void_p()
        check(lib().mlsgpu_hip_host_mesher_landing(self.h, need, C.byref(ptr)))
        base = ptr.value                                  # HostKeyMesh blob order: keys, vertices, triangles (src/mesh.cpp:51-102)
        for off, a in ((0, k), (k.nbytes, v), (k.nbytes + v.nbytes, t)):
            if a.nbytes:
                C.memmove(base + off, a.ctypes.data, a.nbytes)
        hm = HostMesh(base, base + k.nbytes, base + k.nbytes + v.nbytes, len(v), len(t), num_internal)
        check(lib().mlsgpu_hip_host_mesher_add_landed(self.h, chunk_id, C.byref(hm)))

    def landing_pinned(self):
        return bool(lib().mlsgpu_hip_host_mesher_landing_pinned(self.h))

    def finalize(self):
        n = C.c_uint32()
        check(lib().mlsgpu_hip_host_mesher_finalize(self.h, C.byref(n)))
        return n.value

    def boundary(self):
        """(keys, key_clump, clump_vertices, clump_triangles): what a cross-rank merge needs (dist_sink.global_prune)."""
        nk, nc = C.c_uint64(), C.c_uint64()
        check(lib().mlsgpu_hip_host_mesher_boundary(self.h, C.byref(nk), C.byref(nc)))
        keys, kc = np.zeros(nk.value, np.uint64), np.zeros(nk.value, np.uint32)
        cv, ct = np.zeros(nc.value, np.uint64), np.zeros(nc.value, np.uint64)
        check(lib().mlsgpu_hip_host_mesher_boundary_read(self.h, _p(keys), _p(kc), _p(cv), _p(ct)))
        return keys, kc, cv, ct

    def finalize_with(self, keep_clump):
        keep = np.ascontiguousarray(keep_clump, np.uint8)
        n = C.c_uint32()
        check(lib().mlsgpu_hip_host_mesher_finalize_with(self.h, _p(keep), len(keep), C.byref(n)))
        return n.value

    def chunk(self, i):
        cid, nv, nt = C.c_uint64(), C.c_uint64(), C.c_uint64()
        pv, pt = C.c_void_p(), C.c_void_p()
        check(lib().mlsgpu_hip_host_mesher_chunk(self.h, i, C.byref(cid), C.byref(nv), C.byref(nt), C.byref(pv), C.byref(pt)))

        def view(ptr, dtype, count):
            if count == 0:
                return np.zeros(0, dtype)
            buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(ptr.value)
            return np.frombuffer(buf, dtype=dtype, count=count).copy()
        return (int(cid.value), view(pv, np.float32, 3 * nv.value).reshape(-1, 3),
                view(pt, np.uint32, 3 * nt.value).reshape(-1, 3))

    def stats(self):
        out = np.zeros(8, np.uint64)
        check(lib().mlsgpu_hip_host_mesher_stats(self.h, _p(out)))
        names = ["total_vertices", "threshold", "components", "kept_components", "kept_vertices", "kept_triangles",
                 "vertices_added", "triangles_added"]
        return dict(zip(names, [int(x) for x in out]))

    def close(self):
        if getattr(self, "h", None):
            lib().mlsgpu_hip_host_mesher_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PlyReader:
    """FastPly::Reader, src/fast_ply.h:77-262."""

    def __init__(self, path, smooth=1.0, max_radius=float("inf")):
        h = C.c_void_p()
        check(lib().mlsgpu_hip_ply_open(str(path).encode(), smooth, max_radius, C.byref(h)))
        self.h = h

    def __len__(self):
        return int(lib().mlsgpu_hip_ply_size(self.h))

    def layout(self):
        out = np.zeros(10, np.uint64)
        check(lib().mlsgpu_hip_ply_layout(self.h, _p(out)))
        names = ["vertex_size", "vertex_count", "header_size", "x", "y", "z", "nx", "ny", "nz", "radius"]
        return dict(zip(names, [int(v) for v in out]))

    def read(self, first=0, count=None, out=None):
        if count is None:
            count = len(self) - first
        if out is None:
            out = np.zeros(count, SPLAT_DTYPE)
        assert out.dtype == SPLAT_DTYPE and out.flags.c_contiguous and len(out) >= count
        check(lib().mlsgpu_hip_ply_read(self.h, first, count, _p(out)))
        return out

    def load(self, ctx, d_out, first=0, count=None, host_threads=0):
        """File -> device buffer (decode overlapped with the H2D copies)."""
        if count is None:
            count = len(self) - first
        assert d_out.nbytes >= count * SPLAT_DTYPE.itemsize
        check(lib().mlsgpu_hip_ply_load(self.h, ctx.h, first, count, d_out.ptr, host_threads))

    def close(self):
        if getattr(self, "h", None):
            lib().mlsgpu_hip_ply_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FileSet:
    """SplatSet::FileSet (src/splat_set.h:383-700): several PLY files read as one splat sequence."""

    def __init__(self, paths=(), smooth=1.0, max_radius=float("inf"), buffer_size=None):
        h = C.c_void_p()
        check(lib().mlsgpu_hip_fileset_create(smooth, max_radius, C.byref(h)))
        self.h = h
        for p in paths:
            self.add_file(p)
        if buffer_size is not None:
            check(lib().mlsgpu_hip_fileset_set_buffer_size(self.h, buffer_size))

    def add_file(self, path):
        check(lib().mlsgpu_hip_fileset_add_file(self.h, str(path).encode()))

    def __len__(self):
        return int(lib().mlsgpu_hip_fileset_num_splats(self.h))

    def read(self, first=0, count=None, out=None):
        count = len(self) - first if count is None else count
        if out is None:
            out = np.empty(count, SPLAT_DTYPE)
        check(lib().mlsgpu_hip_fileset_read(self.h, first, count, _p(out)))
        return out

    def load(self, ctx, d_out, first=0, count=None, reader_threads=0):
        count = len(self) - first if count is None else count
        check(lib().mlsgpu_hip_fileset_load(self.h, ctx.h, first, count, d_out.ptr, reader_threads))
        return count

    def close(self):
        if getattr(self, "h", None):
            lib().mlsgpu_hip_fileset_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def write_ply(path, vertices, triangles, comments=()):
    """FastPly::Writer's file (src/fast_ply.cpp:443-521)."""
    vertices = np.ascontiguousarray(vertices, np.float32).reshape(-1, 3)
    triangles = np.ascontiguousarray(triangles, np.uint32).reshape(-1, 3)
    arr = (C.c_char_p * max(len(comments), 1))(*[c.encode("ascii") for c in comments])
    check(lib().mlsgpu_hip_write_ply(str(path).encode(), _p(vertices), len(vertices), _p(triangles), len(triangles),
                                     arr, len(comments)))


class Marching:
    """Marching, src/marching.h:494-608."""
    MAX_CELL_BYTES = 872

    def __init__(self, ctx, max_width, max_height, max_depth, max_swathe, mesh_memory, alignment):
        self.ctx = ctx
        h = C.c_void_p()
        check(lib().mlsgpu_hip_marching_create(ctx.h, max_width, max_height, max_depth, max_swathe, mesh_memory,
                                               _p(_u3(alignment)), C.byref(h)))
        self.h = h

    def generate(self, generator, size, key_offset=(0, 0, 0), collector=None):
        """generator: a Generator struct (MlsFunctor.generator()) or host_generator(...)."""
        col = collector or MeshCollector(self.ctx)
        gen = generator if isinstance(generator, Generator) else generator.struct
        rc = lib().mlsgpu_hip_marching_generate(self.h, C.byref(gen), col.cb, None, _p(_u3(size)),
                                                _p(_u3(key_offset)))
        if getattr(col, "error", None) is not None:
            raise col.error
        if getattr(generator, "error", None) is not None:
            raise generator.error
        check(rc)
        return col.batches if isinstance(col, MeshCollector) else col

    @staticmethod
    def generate_batch(marchings, generators, sizes, key_offsets=None):
        """Marching::generate for several buckets in lock-step (mlsgpu_hip_marching_generate_batch): marchings[k] takes
        generators[k] and sizes[k]; returns the ship-outs per bucket (read back), in order."""
        n = len(marchings)
        ctx = marchings[0].ctx
        hs = (C.c_void_p * n)(*[m.h for m in marchings])
        gens = (Generator * n)()
        for k, g in enumerate(generators):
            src = g if isinstance(g, Generator) else g.struct
            C.memmove(C.byref(gens[k]), C.byref(src), C.sizeof(Generator))
        sz = np.ascontiguousarray(sizes, np.uint32).reshape(n, 3)
        ko = np.zeros((n, 3), np.uint32) if key_offsets is None else np.ascontiguousarray(key_offsets, np.uint32).reshape(n, 3)
        cols = [MeshCollector(ctx) for _ in range(n)]

        def cb(user, index, stream, meshp):
            return cols[index].cb(user, stream, meshp)
        fn = BATCH_OUTPUT_FN(cb)
        rc = lib().mlsgpu_hip_marching_generate_batch(hs, gens, n, fn, None, _p(sz), _p(ko))
        for c in cols:
            if c.error is not None:
                raise c.error
        for g in generators:
            if getattr(g, "error", None) is not None:
                raise g.error
        check(rc)
        return [c.batches for c in cols]

    def counters(self):
        out = np.zeros(8, np.uint64)
        check(lib().mlsgpu_hip_marching_counters(self.h, _p(out)))
        names = ["overflows", "shipouts", "nonempty", "occupied", "unwelded", "indices", "welded", "external"]
        return dict(zip(names, [int(x) for x in out]))

    def tables(self):
        count = np.zeros((256, 2), np.uint8)
        start = np.zeros((257, 2), np.uint16)
        data = np.zeros(8192, np.uint8)
        key = np.zeros((2432, 3), np.uint32)
        check(lib().mlsgpu_hip_marching_tables(self.h, _p(count), _p(start), _p(data), _p(key)))
        return count, start, data, key

    def copy_slice(self, field, pitch, src, trg, width, height, z_stride):
        check(lib().mlsgpu_hip_marching_copy_slice(self.h, field.ptr, pitch, src, trg, width, height, z_stride))
        self.ctx.synchronize()

    def __del__(self):
        if getattr(self, "h", None):
            lib().mlsgpu_hip_marching_destroy(self.h)
            self.h = None


class HostGenerator:
    """A Marching::Generator whose values come from a host function, like the HostGenerator of
    test/test_marching.cpp:62-130: fn(xs, ys, z) -> float32 [height, width] per slice."""

    def __init__(self, ctx, fn, alignment):
        self.ctx = ctx
        self.error = None

        def enqueue(user, stream, field, pitch, swp):
            try:
                sw = swp.contents
                ys, xs = np.meshgrid(np.arange(sw.height, dtype=np.uint32), np.arange(sw.width, dtype=np.uint32),
                                     indexing="ij")
                for z in range(sw.zFirst, sw.zLast + 1):
                    vals = np.ascontiguousarray(fn(xs, ys, z), np.float32)
                    # one contiguous copy per slice: rows are `pitch` apart, the padding columns it
                    # overwrites are "undefined" by the Generator contract (src/marching.h:246-249)
                    padded = np.zeros((sw.height, pitch), np.float32)
                    padded[:, :sw.width] = vals
                    flat = padded.reshape(-1)[:(sw.height - 1) * pitch + sw.width]
                    row0 = z * sw.zStride + sw.zBias
                    check(lib().mlsgpu_hip_memcpy_h2d(ctx.h, field + row0 * pitch * 4, _p(flat), flat.nbytes, 0))
                return 0
            except Exception as e:
                self.error = e
                return 5
        self._enqueue = ENQUEUE_FN(enqueue)
        self.struct = Generator()
        for i in range(3):
            self.struct.alignment[i] = alignment[i]
        self.struct.enqueue = self._enqueue
        self.struct.user = None


class Worker:
    """DeviceWorkerGroupBase::Worker, src/workers.cpp:207-286."""

    def __init__(self, ctx, max_bucket_splats, max_cells=255, mesh_memory=0, levels=6, subsampling=3,
                 boundary_limit=1.0, shape=SHAPE_SPHERE, max_swathe=0, grid_spacing=1.0, grid_origin=(0, 0, 0)):
        self.ctx = ctx
        cfg = WorkerConfig()
        cfg.maxBucketSplats = max_bucket_splats
        cfg.maxCells = max_cells
        cfg.meshMemory = mesh_memory
        cfg.levels = levels
        cfg.subsampling = subsampling
        cfg.boundaryLimit = boundary_limit
        cfg.shape = shape
        cfg.maxSwathe = max_swathe
        cfg.gridSpacing = grid_spacing
        for i in range(3):
            cfg.gridOrigin[i] = grid_origin[i]
        self.cfg = cfg
        h = C.c_void_p()
        check(lib().mlsgpu_hip_worker_create(ctx.h, C.byref(cfg), C.byref(h)))
        self.h = h

    def resource_usage(self):
        """Device bytes of the worker's buffers: one set per lane (set_batch)."""
        return lib().mlsgpu_hip_worker_resource_usage_lanes(C.byref(self.cfg), max(1, lib().mlsgpu_hip_worker_batch(self.h)))

    def process(self, splats, first_splat, num_splats, low_extent, num_vertices, collector=None):
        col = collector or MeshCollector(self.ctx)
        rc = lib().mlsgpu_hip_worker_process(self.h, splats.ptr, first_splat, num_splats, _p(_i3(low_extent)),
                                             _p(_u3(num_vertices)), col.cb, None)
        if getattr(col, "error", None) is not None:
            raise col.error
        check(rc)
        return col.batches if isinstance(col, MeshCollector) else col

    def set_batch(self, lanes):
        """Room for `lanes` buckets in lock-step (process_batch); every lane owns a tree, a field and a mesh arena."""
        check(lib().mlsgpu_hip_worker_set_batch(self.h, lanes))

    def set_marching_group(self, buckets):
        """Buckets per set of processCorners / marching launches within the lanes (0: all lanes; default 2)."""
        check(lib().mlsgpu_hip_worker_set_marching_group(self.h, buckets))

    def process_batch(self, splats, items, collector=None):
        """The SubItems of a WorkItem (src/workers.cpp:232-286) through mlsgpu_hip_worker_process_batch: `items` is a list
        of (first_splat, num_splats, low_extent, num_vertices) -- or objects with .first / .count / .low / .num_vertices -- in
        the device buffer `splats`.  Without a collector every bucket's ship-outs are read back: a list (per bucket) of
        lists of batches; with one, every mesh goes to it in bucket order and it is returned."""
        arr = (SubItem * max(len(items), 1))()
        for i, it in enumerate(items):
            first, count, low, nv = (it if isinstance(it, (tuple, list)) else (it.first, it.count, it.low, it.num_vertices))
            arr[i].firstSplat, arr[i].numSplats = first, count
            for a in range(3):
                arr[i].lowExtent[a] = int(low[a])
                arr[i].numVertices[a] = int(nv[a])
        cols = [collector] * len(items) if collector is not None else [MeshCollector(self.ctx) for _ in items]

        def cb(user, index, stream, meshp):
            return cols[index].cb(user, stream, meshp)
        fn = BATCH_OUTPUT_FN(cb)
        rc = lib().mlsgpu_hip_worker_process_batch(self.h, splats.ptr, arr, len(items), fn, None)
        for c in cols:
            if getattr(c, "error", None) is not None:
                raise c.error
        check(rc)
        return collector if collector is not None else [c.batches for c in cols]

    def set_mls_variant(self, variant):
        check(lib().mlsgpu_hip_mls_set_variant(lib().mlsgpu_hip_worker_mls(self.h), variant))

    def set_keep_splats(self, keep):
        """True: process() leaves the splats as they came (resident splats can be processed again)."""
        check(lib().mlsgpu_hip_worker_set_keep_splats(self.h, 1 if keep else 0))

    def set_mls_stats(self, counters):
        """counters: DeviceBuffer of MLS_STATS_WORDS uint64 (or None): see mlsgpu_hip_mls_set_stats."""
        check(lib().mlsgpu_hip_mls_set_stats(lib().mlsgpu_hip_worker_mls(self.h), counters.ptr if counters else None))

    def marching_counters(self, lane=None):
        """Marching's counters: of one lane, or (default) summed over the worker's lanes."""
        n = lib().mlsgpu_hip_worker_batch(self.h)
        out = np.zeros(8, np.uint64)
        for k in (range(n) if lane is None else [lane]):
            one = np.zeros(8, np.uint64)
            check(lib().mlsgpu_hip_marching_counters(lib().mlsgpu_hip_worker_lane_marching(self.h, k), _p(one)))
            out += one
        names = ["overflows", "shipouts", "nonempty", "occupied", "unwelded", "indices", "welded", "external"]
        return dict(zip(names, [int(x) for x in out]))

    def tree_num_entries(self, lane=0):
        n = C.c_uint64()
        check(lib().mlsgpu_hip_tree_num_entries(lib().mlsgpu_hip_worker_lane_tree(self.h, lane), C.byref(n)))
        return n.value

    def tree_arrays(self, lane=0):
        t = lib().mlsgpu_hip_worker_lane_tree(self.h, lane)
        commands = download_ptr(self.ctx, lib().mlsgpu_hip_tree_commands(t), np.int32,
                                lib().mlsgpu_hip_tree_commands_size(t))
        start = download_ptr(self.ctx, lib().mlsgpu_hip_tree_start(t), np.int32, lib().mlsgpu_hip_tree_start_size(t))
        return commands, start

    def __del__(self):
        if getattr(self, "h", None):
            lib().mlsgpu_hip_worker_destroy(self.h)
            self.h = None


class _BorrowedContext:
    """A worker's context handed to an output functor: valid only during the call."""

    def __init__(self, h):
        self.h = h

    def synchronize(self):
        check(lib().mlsgpu_hip_ctx_synchronize(self.h))


class BucketFarm:
    """CopyGroup + one DeviceWorkerGroup per GPU (include/mlsgpu_hip.h, "bucket farm")."""

    def __init__(self, devices, max_bucket_splats, workers_per_device=1, spare=1, collect=False, sink=None, **worker_kw):
        cfg = FarmConfig()
        cfg.numDevices = len(devices)
        self._devs = (C.c_int32 * len(devices))(*devices)
        cfg.devices = C.cast(self._devs, C.POINTER(C.c_int32))
        cfg.workersPerDevice = workers_per_device
        cfg.spare = spare
        cfg.copyThreads = worker_kw.get("copy_threads", 0)      # host threads of one bucket's copy into pinned staging; 0 = 4
        cfg.stagingBuffers = worker_kw.get("staging_buffers", 0)  # pinned staging buffers per copy side; 0 = its GPUs + 2
        w = cfg.worker
        w.maxBucketSplats = max_bucket_splats
        w.maxCells = worker_kw.get("max_cells", 255)
        w.meshMemory = worker_kw.get("mesh_memory", 0)
        w.levels = worker_kw.get("levels", 6)
        w.subsampling = worker_kw.get("subsampling", 3)
        w.boundaryLimit = worker_kw.get("boundary_limit", 1.0)
        w.shape = worker_kw.get("shape", SHAPE_SPHERE)
        w.maxSwathe = worker_kw.get("max_swathe", 0)
        w.gridSpacing = worker_kw.get("grid_spacing", 1.0)
        for i, v in enumerate(worker_kw.get("grid_origin", (0, 0, 0))):
            w.gridOrigin[i] = v
        self.meshes = {}          # chunkId -> list of batches (collect=True)
        self.error = None
        import threading
        self._lock = threading.Lock()

        def cb(user, device, chunk, ctxh, meshp):
            try:
                batch = read_mesh(_BorrowedContext(ctxh), meshp.contents)
                batch["device"] = device
                with self._lock:
                    self.meshes.setdefault(int(chunk), []).append(batch)
                return 0
            except Exception as e:   # never let an exception cross the C boundary
                self.error = e
                return 1
        self.checksums = True     # collect="checksum": set False to stop recording (the callback then only returns)
        self.sums = {}            # chunkId -> [(vertices, triangles, internal, 3 checksums)] (collect="checksum")
        self.shipouts = {}        # chunkId -> ship-outs seen

        def cb_sum(user, device, chunk, ctxh, meshp):
            # sizes + device-side checksums of every ship-out, per chunk (nothing is copied to the host): with one chunk id
            # per bucket the digest below does not depend on which worker ran which bucket
            try:
                if not self.checksums:      # timed passes: the farm's own counters suffice
                    return 0
                mm = meshp.contents
                out = np.zeros(3, np.uint64)
                check(lib().mlsgpu_hip_mesh_checksum(ctxh, meshp, _p(out)))
                rec = (int(mm.numVertices), int(mm.numTriangles), int(mm.numInternalVertices)) + tuple(int(x) for x in out)
                with self._lock:
                    self.sums.setdefault(int(chunk), []).append(rec)
                return 0
            except Exception as e:   # never let an exception cross the C boundary
                self.error = e
                return 1
        if collect == "checksum":
            self._cb = FARM_OUTPUT_FN(cb_sum)
        else:
            self._cb = FARM_OUTPUT_FN(cb) if collect else C.cast(None, FARM_OUTPUT_FN)
        user = None
        if sink is not None:
            # a device Mesher: the workers append their ship-outs to it, in C (mlsgpu_hip_mesher_farm_output)
            assert not collect
            self._sink = sink
            self._cb = C.cast(lib().mlsgpu_hip_mesher_farm_output, FARM_OUTPUT_FN)
            user = sink.h
        h = C.c_void_p()
        check(lib().mlsgpu_hip_farm_create(C.byref(cfg), self._cb, user, C.byref(h)))
        self.h = h

    def set_batch(self, lanes):
        """The workers take the buckets of a device item `lanes` at a time through one set of launches."""
        check(lib().mlsgpu_hip_farm_set_batch(self.h, lanes))

    def submit(self, splats, low_extent, num_vertices, chunk_id):
        splats = np.ascontiguousarray(splats)
        check(lib().mlsgpu_hip_farm_submit(self.h, _p(splats), len(splats), _p(_i3(low_extent)), _p(_u3(num_vertices)),
                                           chunk_id))

    def acquire(self, num_splats):
        """CopyGroup::get: a numpy view of pinned staging memory for one bucket; fill it, then push()."""
        ptr = C.c_void_p()
        check(lib().mlsgpu_hip_farm_acquire(self.h, num_splats, C.byref(ptr)))
        if num_splats == 0:
            return np.zeros(0, SPLAT_DTYPE)
        buf = (C.c_char * (num_splats * SPLAT_DTYPE.itemsize)).from_address(ptr.value)
        return np.frombuffer(buf, dtype=SPLAT_DTYPE, count=num_splats)

    def push(self, num_splats, low_extent, num_vertices, chunk_id):
        check(lib().mlsgpu_hip_farm_push(self.h, num_splats, _p(_i3(low_extent)), _p(_u3(num_vertices)), chunk_id))

    def submit_device(self, device, d_splats, d_ids_ptr, num_splats, reference, spacing, extents, low_extent, num_vertices,
                      chunk_id, wait=True):
        """A bucket whose splats are on the device already (the bucketer's callback): loaded by a device kernel.
        wait=False: returns as soon as the load is enqueued, with the event (an integer handle) that fires when the load has
        read d_splats and the id list -- what bucket_cloud's on_bucket returns to the bucketer."""
        g = _grid_struct(reference, spacing, extents)
        if wait:
            check(lib().mlsgpu_hip_farm_submit_device(self.h, device, d_splats.ptr, d_ids_ptr, num_splats, C.byref(g),
                                                      _p(_i3(low_extent)), _p(_u3(num_vertices)), chunk_id))
            return None
        event = C.c_void_p()
        check(lib().mlsgpu_hip_farm_submit_device_async(self.h, device, d_splats.ptr, d_ids_ptr, num_splats, C.byref(g),
                                                        _p(_i3(low_extent)), _p(_u3(num_vertices)), chunk_id, C.byref(event)))
        return event.value

    def set_host_landing(self, sink):
        """The host route without the ring: every ship-out is read back straight into memory of the HostMesher `sink`
        (page-locked slabs of its own) and adopted there without a copy (mlsgpu_hip_farm_set_host_landing)."""
        assert isinstance(sink, HostMesher)
        self._host_sink = sink
        check(lib().mlsgpu_hip_farm_set_host_landing(self.h, C.cast(lib().mlsgpu_hip_host_mesher_farm_landing, C.c_void_p),
                                                     C.cast(lib().mlsgpu_hip_host_mesher_farm_output_landed, C.c_void_p), sink.h))
        if lib().mlsgpu_hip_host_mesher_node(sink.h) < 0:
            node = self.placement()["ring_node"]
            if node >= 0:
                lib().mlsgpu_hip_host_mesher_set_node(sink.h, node)

    def set_host_output(self, ring_bytes, sink=None):
        """Every ship-out is read back through a pinned circular buffer of `ring_bytes` and handed to ONE mesher thread
        (OutputGeneratorBuilder::Functor + MesherGroup, src/workers.h:488-509, src/workers.cpp:47-85).  `sink`: None
        (read back and dropped), a HostMesher (welded on the host, in C++), or a callable(device, chunk_id, batch) that
        receives numpy copies."""
        self._host_sink = sink
        fn, user = None, None
        if isinstance(sink, HostMesher):
            fn = C.cast(lib().mlsgpu_hip_host_mesher_farm_output, C.c_void_p)
            user = sink.h
        elif sink is not None:
            def cb(_user, device, chunk, hm):
                try:
                    sink(device, int(chunk), host_mesh_arrays(hm.contents))
                    return 0
                except Exception as e:   # never let an exception cross the C boundary
                    self.error = e
                    return 1
            self._host_cb = FARM_HOST_OUTPUT_FN(cb)
            fn = C.cast(self._host_cb, C.c_void_p)
        check(lib().mlsgpu_hip_farm_set_host_output(self.h, ring_bytes, fn, user))
        if isinstance(sink, HostMesher) and lib().mlsgpu_hip_host_mesher_node(sink.h) < 0:
            # a welder that has not been placed joins the ring's NUMA node (refused, and ignored, once its threads exist)
            node = self.placement()["ring_node"]
            if node >= 0:
                lib().mlsgpu_hip_host_mesher_set_node(sink.h, node)

    def placement(self):
        """Where the farm put things: {"nodes", "ring_node", "devices": [{device, node, side}], "sides": [{node,
        staging_node, staging_buffers, copy_threads}]} (mlsgpu_hip_farm_placement)."""
        out = np.zeros(100, np.int32)
        check(lib().mlsgpu_hip_farm_placement(self.h, _p(out)))
        return {"nodes": int(out[2]), "ring_node": int(out[1]),
                "devices": [dict(device=int(out[4 + 3 * d]), node=int(out[5 + 3 * d]), side=int(out[6 + 3 * d]))
                            for d in range(min(int(out[3]), 16))],
                "sides": [dict(node=int(out[52 + 4 * k]), staging_node=int(out[53 + 4 * k]), staging_buffers=int(out[54 + 4 * k]),
                               copy_threads=int(out[55 + 4 * k])) for k in range(min(int(out[0]), 12))]}

    def copy_clock(self):
        """Seconds the copy side spent (mlsgpu_hip_farm_copy_clock): fill, wait_staging, wait_item, h2d, span; counts."""
        out = np.zeros(8, np.float64)
        check(lib().mlsgpu_hip_farm_copy_clock(self.h, _p(out)))
        return dict(fill_s=float(out[0]), wait_staging_s=float(out[1]), wait_item_s=float(out[2]), h2d_s=float(out[3]),
                    span_s=float(out[4]), copies=int(out[5]), cross_side=int(out[6]), enqueue_s=float(out[7]))

    def worker_clock(self):
        """The device workers' clock (mlsgpu_hip_farm_worker_clock): launch sets, buckets, idle and busy seconds."""
        out = np.zeros(4, np.float64)
        check(lib().mlsgpu_hip_farm_worker_clock(self.h, _p(out)))
        return dict(launch_sets=int(out[0]), buckets=int(out[1]), idle_s=float(out[2]), busy_s=float(out[3]))

    def group_clock(self, group):
        """One device group's share of worker_clock() (mlsgpu_hip_farm_group_clock)."""
        out = np.zeros(4, np.float64)
        check(lib().mlsgpu_hip_farm_group_clock(self.h, group, _p(out)))
        return dict(launch_sets=int(out[0]), buckets=int(out[1]), idle_s=float(out[2]), busy_s=float(out[3]))

    def host_stats(self):
        out = np.zeros(4, np.uint64)
        check(lib().mlsgpu_hip_farm_host_stats(self.h, _p(out)))
        return dict(zip(["meshes", "bytes", "ring_waits", "largest_mesh_bytes"], [int(x) for x in out]))

    def finish(self):
        rc = lib().mlsgpu_hip_farm_finish(self.h)
        if self.error is not None:
            raise self.error
        check(rc)

    def digest(self):
        """sha256/16 over (chunk id, ship-out records) in chunk order (collect="checksum")."""
        import hashlib
        h = hashlib.sha256()
        for chunk in sorted(self.sums):
            h.update(np.array([chunk, len(self.sums[chunk])], np.uint64).tobytes())
            for rec in self.sums[chunk]:
                h.update(np.array(rec, np.uint64).tobytes())
        return h.hexdigest()[:16]

    def stats(self):
        out = np.zeros(24, np.uint64)
        check(lib().mlsgpu_hip_farm_stats(self.h, _p(out)))
        names = ["buckets", "splats", "h2d_bytes", "items", "shipouts", "vertices", "triangles", "external"]
        d = dict(zip(names, [int(x) for x in out[:8]]))
        d["per_device"] = [int(x) for x in out[8:24]]
        hi = C.c_uint64(0)
        check(lib().mlsgpu_hip_farm_in_flight_max(self.h, C.byref(hi)))
        d["in_flight_max"] = int(hi.value)
        return d

    def close(self):
        if getattr(self, "h", None):
            lib().mlsgpu_hip_farm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _grid_struct(reference, spacing, extents):
    g = GridStruct()
    for i in range(3):
        g.reference[i] = float(reference[i])
    g.spacing = float(spacing)
    ext = np.asarray(extents, np.int64).reshape(6)
    for i in range(6):
        g.extents[i] = int(ext[i])
    return g


def bucket_cloud(ctx, d_splats, num_splats, reference, spacing, extents, max_splats, max_cells, chunk_cells=0,
                 micro_cells=0, max_split=1 << 30, on_bucket=None):
    """Bucket::bucket (src/bucket.h:170-180) over a cloud resident on the device.

    on_bucket(leaf, d_ids_ptr) is called for every bucket while its id list is valid on the device; without it the
    ids are copied to the host.  An on_bucket that only ENQUEUES its read of the list returns the event behind that read
    (BucketFarm.submit_device(..., wait=False)): the bucketer orders the list's reuse behind it on the GPU
    (mlsgpu_bucket::consumed).  Returns the list of leaves dict(extents, chunk, depth, num_splats[, ids])."""
    leaves = []
    failure = []

    def cb(_user, _ctx, b):
        try:
            b = b.contents
            leaf = dict(extents=tuple(int(b.extents[i]) for i in range(6)), chunk=tuple(int(b.chunk[i]) for i in range(3)),
                        depth=int(b.depth), num_splats=int(b.numSplats))
            if on_bucket is not None:
                event = on_bucket(leaf, b.dIds)
                if event:
                    b.consumed[0] = event
            else:
                ids = np.empty(leaf["num_splats"], np.uint32)
                if len(ids):
                    check(lib().mlsgpu_hip_memcpy_d2h(ctx.h, _p(ids), b.dIds, ids.nbytes, 0))
                leaf["ids"] = ids
            leaves.append(leaf)
            return 0
        except Exception as e:      # noqa: BLE001 - reported after the C call returns
            failure.append(e)
            return 1
    fn = BUCKET_FN(cb)
    g = _grid_struct(reference, spacing, extents)
    p = BucketParams(max_splats, max_cells, chunk_cells, micro_cells, max_split)
    cell = C.c_uint64(0)
    rc = lib().mlsgpu_hip_bucket(ctx.h, d_splats.ptr if d_splats is not None else None, num_splats, C.byref(g), C.byref(p),
                                 fn, None, C.byref(cell))
    if failure:
        raise failure[0]
    if rc == 6:
        e = DensityError("[6] " + lib().mlsgpu_hip_last_error().decode("utf-8", "replace"))
        e.cell_splats = int(cell.value)
        raise e
    check(rc)
    return leaves


class _DevicePtr:
    """what BucketFarm.submit_device and bucket_load take for the cloud: anything with .ptr"""

    def __init__(self, ptr):
        self.ptr = ptr


def bucket_cloud_stream(ctx, fileset, reference, spacing, extents, max_splats, max_cells, budget_splats, chunk_splats,
                        chunk_cells=0, micro_cells=0, max_split=1 << 30, reader_threads=0, on_bucket=None):
    """Bucket::bucket over a FileSet that need not fit the device (mlsgpu_hip_bucket_stream).  on_bucket(leaf, d_splats,
    d_ids_ptr): d_splats (.ptr) is the resident batch the ids index, valid during the call.  Returns (leaves, stats)."""
    leaves = []
    failure = []

    def cb(_user, _ctx, b):
        try:
            b = b.contents
            leaf = dict(extents=tuple(int(b.extents[i]) for i in range(6)), chunk=tuple(int(b.chunk[i]) for i in range(3)),
                        depth=int(b.depth), num_splats=int(b.numSplats))
            if on_bucket is not None:
                on_bucket(leaf, _DevicePtr(b.dSplats), b.dIds)
            leaves.append(leaf)
            return 0
        except Exception as e:      # noqa: BLE001 - reported after the C call returns
            failure.append(e)
            return 1
    fn = BUCKET_FN(cb)
    g = _grid_struct(reference, spacing, extents)
    p = BucketParams(max_splats, max_cells, chunk_cells, micro_cells, max_split)
    cell = C.c_uint64(0)
    stats = np.zeros(4, np.uint64)
    rc = lib().mlsgpu_hip_bucket_stream(ctx.h, fileset.h, C.byref(g), C.byref(p), budget_splats, chunk_splats, reader_threads,
                                        fn, None, C.byref(cell), _p(stats))
    if failure:
        raise failure[0]
    if rc == 6:
        e = DensityError("[6] " + lib().mlsgpu_hip_last_error().decode("utf-8", "replace"))
        e.cell_splats = int(cell.value)
        raise e
    check(rc)
    return leaves, dict(zip(["file_passes", "batches", "batch_splats", "chunks_skipped"], [int(x) for x in stats]))


def bounding_grid_files(ctx, fileset, spacing, bucket_size, chunk_splats, reader_threads=0):
    """FastBlobSet::makeBoundingGrid for a FileSet that need not fit the device: (reference, spacing, extents)."""
    g = GridStruct()
    check(lib().mlsgpu_hip_fileset_bounding_grid(fileset.h, ctx.h, spacing, bucket_size, chunk_splats, reader_threads, C.byref(g)))
    return tuple(g.reference), float(g.spacing), tuple(int(x) for x in g.extents)


def bounding_grid(ctx, d_splats, num_splats, spacing, bucket_size):
    """FastBlobSet::makeBoundingGrid (src/splat_set_impl.h:770-811): (reference, spacing, extents)."""
    g = GridStruct()
    check(lib().mlsgpu_hip_bounding_grid(ctx.h, d_splats.ptr, num_splats, spacing, bucket_size, C.byref(g)))
    return tuple(g.reference), float(g.spacing), tuple(int(v) for v in g.extents)


def bucket_load(ctx, d_splats, d_ids_ptr, num_splats, reference, spacing, extents, d_out):
    """BucketLoader's gather + world -> full-grid vertex transform on the device (src/bucket_loader.cpp:77-85)."""
    g = _grid_struct(reference, spacing, extents)
    check(lib().mlsgpu_hip_bucket_load(ctx.h, d_splats.ptr, d_ids_ptr, num_splats, C.byref(g), d_out.ptr))


def transform_splats(splats, reference, spacing, low_extent):
    """BucketLoader's world -> grid transform, in place (src/bucket_loader.cpp:77-85)."""
    assert splats.dtype == SPLAT_DTYPE and splats.flags.c_contiguous
    lib().mlsgpu_hip_transform_splats(_p(splats), len(splats), _p(np.ascontiguousarray(reference, np.float32)),
                                      spacing, _p(_i3(low_extent)))
