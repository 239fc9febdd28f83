"""ctypes binding of the C ABI in include/xslam_amd.h (libxslam_hip.so, built in-tree by
x-slam_amd/csrc/Makefile).  This is the only compute path: if the library is missing the
import fails loudly — there is no CPU fallback.

Device memory, streams and collectives come from PyTorch-ROCm (plumbing); every argument
crossing this boundary is a raw device pointer, a pitch in bytes or a small host array.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libxslam_hip.so")

_vp = C.c_void_p
_sz = C.c_size_t
_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int)
_f64p = C.POINTER(C.c_double)


class XsError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
            "or make -C x-slam_amd/csrc).  There is no CPU fallback.")
    return C.CDLL(LIB_PATH)


_lib = _load()

_SIGS = {
    "xs_last_error": (C.c_char_p, []),
    "xs_abi_version": (C.c_int, []),
    "xs_init_volume": (C.c_int, [_vp, _vp, _vp, _sz, _i32p, C.c_int, C.c_int, _vp]),
    "xs_scale_depth": (C.c_int, [_vp, _sz, C.c_int, C.c_int, _vp, _sz, _vp]),
    "xs_integrate_tsdf_volume": (C.c_int, [_vp, _sz, C.c_int, C.c_int, _f32p, C.c_int, _i32p, C.c_float, _f32p, _f32p, C.c_float,
                                           _vp, _vp, _vp, _sz, _vp, _sz, C.c_float, C.c_int, C.c_int, _vp, _vp]),
    "xs_integrate_scaled": (C.c_int, [_vp, _sz, C.c_int, C.c_int, _f32p, C.c_int, _i32p, C.c_float, _f32p, _f32p, C.c_float,
                                      _vp, _vp, _vp, _sz, C.c_float, C.c_int, C.c_int, _vp, _vp]),
}


def _bind():
    for name, (res, args) in _SIGS.items():
        f = getattr(_lib, name)
        f.restype = res
        f.argtypes = args


_bind()


def check(rc):
    if rc != 0:
        raise XsError(f"hip error {rc}: {_lib.xs_last_error().decode()}")


def _fa(x, n):
    a = np.ascontiguousarray(x, dtype=np.float32).reshape(-1)
    assert a.size == n, (a.size, n)
    return a


def _ia(x, n):
    a = np.ascontiguousarray(x, dtype=np.int32).reshape(-1)
    assert a.size == n
    return a


def _ptr(t):
    """Device address of a torch tensor (or None / int passthrough)."""
    if t is None:
        return None
    if isinstance(t, int):
        return t
    return t.data_ptr()


def _stream(stream):
    if stream is None:
        return None
    return stream.cuda_stream if hasattr(stream, "cuda_stream") else int(stream)


def abi_version():
    return _lib.xs_abi_version()


def init_volume(value, weight, grad, step_bytes, res, z0=0, z1=None, stream=None):
    r = _ia(res, 3)
    z1 = int(r[2]) if z1 is None else z1
    check(_lib.xs_init_volume(_ptr(value), _ptr(weight), _ptr(grad), step_bytes, r.ctypes.data_as(_i32p), z0, z1, _stream(stream)))


def scale_depth(depth, depth_step, rows, cols, scaled, scaled_step, stream=None):
    check(_lib.xs_scale_depth(_ptr(depth), depth_step, rows, cols, _ptr(scaled), scaled_step, _stream(stream)))


def integrate_tsdf_volume(depth, depth_step, rows, cols, intr, max_weight, res, voxel_size, Rv2c, tv2c, tranc_dist, value, weight,
                          grad, vol_step, depth_scaled, scaled_step, threshold=0.0, z0=0, z1=None, updated=None, stream=None):
    r = _ia(res, 3)
    k, R, t = _fa(intr, 4), _fa(Rv2c, 18), _fa(tv2c, 6)
    z1 = int(r[2]) if z1 is None else z1
    check(_lib.xs_integrate_tsdf_volume(_ptr(depth), depth_step, rows, cols, k.ctypes.data_as(_f32p), max_weight,
                                        r.ctypes.data_as(_i32p), voxel_size, R.ctypes.data_as(_f32p), t.ctypes.data_as(_f32p),
                                        tranc_dist, _ptr(value), _ptr(weight), _ptr(grad), vol_step, _ptr(depth_scaled), scaled_step,
                                        threshold, z0, z1, _ptr(updated), _stream(stream)))


def integrate_scaled(depth_scaled, scaled_step, rows, cols, intr, max_weight, res, voxel_size, Rv2c, tv2c, tranc_dist, value, weight,
                     grad, vol_step, threshold=0.0, z0=0, z1=None, updated=None, stream=None):
    r = _ia(res, 3)
    k, R, t = _fa(intr, 4), _fa(Rv2c, 18), _fa(tv2c, 6)
    z1 = int(r[2]) if z1 is None else z1
    check(_lib.xs_integrate_scaled(_ptr(depth_scaled), scaled_step, rows, cols, k.ctypes.data_as(_f32p), max_weight,
                                   r.ctypes.data_as(_i32p), voxel_size, R.ctypes.data_as(_f32p), t.ctypes.data_as(_f32p), tranc_dist,
                                   _ptr(value), _ptr(weight), _ptr(grad), vol_step, threshold, z0, z1, _ptr(updated), _stream(stream)))
