"""`simple_knn._C` of the reference, bound to the C ABI of include/ex4d_knn.h (libex4d_hip.so) with ctypes.

distCUDA2(points[P,3] float32 on a ROCm device) -> [P] float32: mean squared distance to the 3 nearest other points
(submodules/simple-knn/spatial.cu:15-27, simple_knn.cu:129-221).  No CPU fallback.
"""
import ctypes as C

import torch

from .. import _C as _lib_loader

EXPORTS = ("ex4d_knn_last_error", "ex4d_dist2_scratch_bytes", "ex4d_dist2")


def _lib():
    lib = _lib_loader.load()
    if not getattr(lib, "_knn_ready", False):
        lib.ex4d_knn_last_error.restype = C.c_char_p
        lib.ex4d_dist2_scratch_bytes.restype = C.c_size_t
        lib.ex4d_dist2_scratch_bytes.argtypes = [C.c_int32]
        lib.ex4d_dist2.restype = C.c_int
        lib.ex4d_dist2.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib._knn_ready = True
    return lib


def distCUDA2(points):
    lib = _lib()
    if not points.is_cuda:
        raise RuntimeError(f"points are on {points.device}: distCUDA2 only runs on a ROCm GPU (no CPU fallback)")
    if points.dim() != 2 or points.shape[1] != 3:
        raise RuntimeError("points must be [P,3]")
    pts = points.contiguous().float()
    P = pts.shape[0]
    means = torch.full((P,), 0.0, dtype=torch.float32, device=pts.device)          # spatial.cu:21
    if P == 0:
        return means
    scratch = torch.empty(lib.ex4d_dist2_scratch_bytes(P), dtype=torch.uint8, device=pts.device)
    with torch.cuda.device(pts.device):
        rc = lib.ex4d_dist2(P, pts.data_ptr(), means.data_ptr(), scratch.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc:
        raise RuntimeError(lib.ex4d_knn_last_error().decode())
    return means
