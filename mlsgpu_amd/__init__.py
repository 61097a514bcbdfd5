"""mlsgpu_amd -- MI355X (gfx950) implementation of mlsgpu's per-bucket device pipeline.

The product is the HIP library ``libmlsgpu_hip.so`` (C-ABI: include/mlsgpu_hip.h) and the C++
classes in ``mlsgpu_amd/host`` that mirror the reference's call surface.  This Python package is
a thin ctypes binding of the same C-ABI, used by the tests and by bench.py.  There is no CPU
fallback: importing :mod:`mlsgpu_amd.binding` without the built library raises.
"""
from .binding import (BucketFarm, Context, DensityError, DeviceBuffer, FormatError, HipError, HostMesher, InvalidArgument, LengthError, Marching, Mesher, MlsError,
                      MlsFunctor, SPLAT_DTYPE, SplatTree, Swathe, Worker, WorkerConfig, lib, library_path)

__all__ = ["BucketFarm", "Context", "DensityError", "DeviceBuffer", "FormatError", "HipError", "HostMesher", "InvalidArgument", "LengthError", "Marching", "Mesher", "MlsError",
           "MlsFunctor", "SPLAT_DTYPE", "SplatTree", "Swathe", "Worker", "WorkerConfig", "lib", "library_path"]
