"""MI355X-native Whitted renderer for P3D .p3f scenes -- Python harness over the C-ABI.

The product is csrc/ (HIP kernels, C-ABI, C++ host layer).  This package is the thin ctypes
binding that tests/ and bench.py drive it through; it holds no rendering logic and has no CPU
fallback: importing `api` without a built libp3d_hip.so raises.
"""
from .api import (PathTracer, pt_debug_hash, ACCEL_BVH, ACCEL_GRID, ACCEL_NONE, Counters, DeviceScene, HostScene, P3DError,
                  build_native, debug_intersect, debug_powf, debug_check_rcp, device_count, host_bvh, lib, local_rows, Comm, comm_unique_id,
                  gather_all, tune_schedule)

__all__ = ["PathTracer", "pt_debug_hash", "ACCEL_BVH", "ACCEL_GRID", "ACCEL_NONE", "Counters", "DeviceScene", "HostScene", "P3DError",
           "build_native", "debug_intersect", "debug_powf", "debug_check_rcp", "device_count", "host_bvh", "lib", "local_rows", "Comm", "comm_unique_id",
           "gather_all", "tune_schedule"]
