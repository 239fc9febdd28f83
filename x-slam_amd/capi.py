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

class IntegrateOpts(C.Structure):
    """xs_integrate_opts (include/xslam_amd.h): what an integrate / classify call takes besides its arguments proper."""
    _fields_ = [("struct_bytes", C.c_uint), ("flags", C.c_uint), ("depth_tiles", C.c_void_p), ("signmap", C.c_void_p),
                ("start_event", C.c_void_p), ("stop_event", C.c_void_p), ("pose_mailbox", C.c_void_p), ("mailbox_seq", C.c_uint),
                ("mailbox_slack", C.c_float), ("pose_dev", C.c_void_p)]


class RaycastOpts(C.Structure):
    """xs_raycast_opts (include/xslam_amd.h)."""
    _fields_ = [("struct_bytes", C.c_uint), ("signmap_shift", C.c_int), ("signmap", C.c_void_p), ("signmap_tranc_dist", C.c_float),
                ("pyr_vmap1", C.c_void_p), ("pyr_nmap1", C.c_void_p), ("pyr_step1", C.c_size_t), ("pyr_vmap2", C.c_void_p), ("pyr_nmap2", C.c_void_p),
                ("pyr_step2", C.c_size_t), ("completion_event", C.c_void_p), ("steps_dev", C.c_void_p), ("pyramid_built", C.c_int)]


_SIGS = {
    "xs_integrate_scaled_ex2": (C.c_int, [_vp, _sz, C.c_int, C.c_int, _f32p, C.c_int, _i32p, C.c_float, _f32p, _f32p, C.c_float, _vp, _vp, _vp,
                                         _sz, C.c_float, C.c_int, C.c_int, _vp, _vp, _vp, C.POINTER(IntegrateOpts), _vp]),
    "xs_integrate_classify_ex": (C.c_int, [C.c_int, C.c_int, _f32p, _i32p, C.c_float, _f32p, _f32p, C.c_float, C.c_int, C.c_int, _vp, _vp,
                                           C.c_float, C.POINTER(IntegrateOpts), _vp]),
    "xs_raycast_ex": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _f32p, C.c_float, _i32p, C.c_float, _vp, _vp, _sz, _vp, _vp, _sz,
                                C.c_int, C.c_int, _vp, _vp, C.POINTER(RaycastOpts), _vp]),
    "xs_raycast_slab_ex": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _f32p, C.c_float, _i32p, C.c_float, _vp, _vp, _sz, C.c_int, C.c_int,
                                     C.c_int, C.c_int, _vp, _vp, _sz, C.c_int, C.c_int, _vp, C.POINTER(RaycastOpts), _vp]),
    "xs_resize_pyramid_ex": (C.c_int, [_vp, _vp, _sz, C.c_int, C.c_int, _vp, _vp, _sz, _vp, _vp, _sz, _vp, _vp]),
    "xs_last_error": (C.c_char_p, []),
    "xs_abi_version": (C.c_int, []),
    "xs_init_volume": (C.c_int, [_vp, _vp, _vp, _sz, _i32p, C.c_int, C.c_int, _vp]),
    "xs_scale_depth": (C.c_int, [_vp, _sz, C.c_int, C.c_int, _vp, _sz, _vp]),
    "xs_integrate_tsdf_volume": (C.c_int, [_vp, _sz, C.c_int, C.c_int, _f32p, C.c_int, _i32p, C.c_float, _f32p, _f32p, C.c_float,
                                           _vp, _vp, _vp, _sz, _vp, _sz, C.c_float, C.c_int, C.c_int, _vp, _vp]),
    "xs_integrate_scaled": (C.c_int, [_vp, _sz, C.c_int, C.c_int, _f32p, C.c_int, _i32p, C.c_float, _f32p, _f32p, C.c_float,
                                      _vp, _vp, _vp, _sz, C.c_float, C.c_int, C.c_int, _vp, _vp, _vp, _vp]),
    "xs_integrate_scaled_ex": (C.c_int, [_vp, _sz, C.c_int, C.c_int, _f32p, C.c_int, _i32p, C.c_float, _f32p, _f32p, C.c_float, _vp, _vp, _vp,
                                        _sz, C.c_float, C.c_int, C.c_int, _vp, _vp, _vp, C.c_uint, _vp]),
    "xs_integrate_workspace_clear": (C.c_int, [_vp, _vp]),
    "xs_integrate_fold_counts": (C.c_int, [_vp, _vp, _vp]),
    "xs_integrate_classify": (C.c_int, [C.c_int, C.c_int, _f32p, _i32p, C.c_float, _f32p, _f32p, C.c_float, C.c_int, C.c_int, _vp, _vp,
                                        C.c_float, C.c_uint, _vp]),
    "xs_integrate_list_covers": (C.c_int, [C.c_int, C.c_int, _f32p, _i32p, C.c_float, _f32p, _f32p, C.c_float, _f32p, _f32p]),
    "xs_integrate_workspace_bytes": (_sz, [_i32p, C.c_int]),
    "xs_scale_depth_max": (C.c_int, [_vp, _sz, C.c_int, C.c_int, _vp, _sz, _vp, _vp]),
    "xs_scale_depth_tiles": (C.c_int, [_vp, _sz, C.c_int, C.c_int, _vp, _sz, _vp, _vp, _vp]),
    "xs_depth_tiles": (C.c_int, [_vp, _sz, C.c_int, C.c_int, _vp, _vp]),
    "xs_depth_tiles_bytes": (_sz, [C.c_int, C.c_int]),
    "xs_tsdf_reduce_workspace_bytes": (_sz, []),
    "xs_tsdf_reduce_workspace_init": (C.c_int, [_vp, _vp]),
    "xs_compute_local_tsdf_hessian": (C.c_int, [_vp, _sz, C.c_int, C.c_int, _f32p, _i32p, C.c_float, _f32p, _f32p, C.c_float, _vp,
                                                _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp]),
    "xs_compute_local_tsdf_loss": (C.c_int, [_vp, _sz, C.c_int, C.c_int, _f32p, _i32p, C.c_float, _f32p, _f32p, C.c_float, _vp,
                                             _vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp]),
    "xs_bilateral_filter": (C.c_int, [_vp, _sz, C.c_int, C.c_int, _vp, _sz, _vp]),
    "xs_pyr_down": (C.c_int, [_vp, _sz, C.c_int, C.c_int, _vp, _sz, _vp]),
    "xs_create_vmap": (C.c_int, [_f32p, _vp, _sz, C.c_int, C.c_int, _vp, _sz, _vp]),
    "xs_create_nmap": (C.c_int, [_vp, _vp, _sz, C.c_int, C.c_int, _vp]),
    "xs_create_vnmaps": (C.c_int, [C.c_int, _f32p, C.POINTER(_vp), C.POINTER(_sz), C.c_int, C.c_int, C.POINTER(_vp), C.POINTER(_vp),
                                   C.POINTER(_sz), _vp]),
    "xs_create_vnmaps_real": (C.c_int, [C.c_int, _f32p, C.POINTER(_vp), C.POINTER(_sz), C.c_int, C.c_int, C.POINTER(_vp), C.POINTER(_vp),
                                        C.POINTER(_sz), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_sz), _vp]),
    "xs_resize_vmap": (C.c_int, [_vp, _sz, C.c_int, C.c_int, _vp, _sz, _vp]),
    "xs_resize_nmap": (C.c_int, [_vp, _sz, C.c_int, C.c_int, _vp, _sz, _vp]),
    "xs_raycast": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _f32p, C.c_float, _i32p, C.c_float, _vp, _vp, _sz, _vp, _vp, _sz,
                             C.c_int, C.c_int, _vp, _vp, _vp]),
    "xs_raycast_slab": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _f32p, C.c_float, _i32p, C.c_float, _vp, _vp, _sz, C.c_int, C.c_int,
                                  C.c_int, C.c_int, _vp, _vp, _sz, C.c_int, C.c_int, _vp, _vp]),
    "xs_raycast_compose_mask": (C.c_int, [_vp, _vp, _vp, _vp, _sz, C.c_int, C.c_int, _vp]),
    "xs_raycast_compose_finish": (C.c_int, [_vp, _vp, _vp, _sz, C.c_int, C.c_int, _vp, _vp]),
    "xs_raycast_compose_entry_bytes": (_sz, []),
    "xs_raycast_compose_pack": (C.c_int, [_vp, _vp, _vp, _vp, _sz, C.c_int, C.c_int, _vp, _vp, _vp]),
    "xs_raycast_compose_scatter": (C.c_int, [_vp, C.c_long, _vp, _vp, _sz, C.c_int, C.c_int, _vp]),
    "xs_resize_pyramid": (C.c_int, [_vp, _vp, _sz, C.c_int, C.c_int, _vp, _vp, _sz, _vp, _vp, _sz, _vp]),
    "xs_tsdf_gauss_newton_terms": (C.c_int, [_vp, _sz, C.c_int, C.c_int, _f32p, _i32p, C.c_float, _f32p, _f32p, C.c_float, _vp, C.c_int,
                                             C.c_int, _vp, _vp, _vp]),
    "xs_tsdf_gauss_newton_terms_ex": (C.c_int, [_vp, _sz, C.c_int, C.c_int, _f32p, _i32p, C.c_float, _f32p, _f32p, C.c_float, _vp, C.c_int,
                                                C.c_int, _vp, _vp, _vp, _vp]),
    "xs_gn_publish_bytes": (_sz, []),
    "xs_gn_mailbox_bytes": (_sz, []),
    "xs_gn_post_poses": (None, [_vp, _f32p, _f32p, C.c_uint, C.c_int]),
    "xs_gn_publish_sums": (C.c_int, [_vp, C.c_int, _vp, C.c_ulonglong, _vp]),
    "xs_extract_workspace_bytes": (_sz, [_i32p]),
    "xs_extract_points": (C.c_int, [_vp, _sz, _i32p, C.c_float, C.c_int, C.c_int, C.c_int, _vp, _sz, _vp, C.POINTER(_sz), C.POINTER(_sz), _vp]),
    "xs_extract_normals": (C.c_int, [_vp, _sz, _i32p, C.c_float, C.c_int, C.c_int, _vp, _sz, _vp, _vp]),
    "xs_icp_wait_pairs": (C.c_int, [_vp, C.c_ulonglong, _vp, C.c_longlong]),
    "xs_icp_workspace_bytes": (_sz, []),
    "xs_icp_workspace_init": (C.c_int, [_vp, _vp]),
    "xs_icp_accumulate": (C.c_int, [_f32p, _f32p, _vp, _vp, _f32p, _f32p, _f32p, _vp, _vp, _sz, C.c_int, C.c_int, C.c_float,
                                    C.c_float, C.c_int, C.c_int, _vp, _vp, _vp, C.c_ulonglong, _vp]),
    "xs_icp_pose_state_bytes": (_sz, []),
    "xs_icp_mailbox_bytes": (_sz, []),
    "xs_icp_mailbox_alloc": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_int)]),
    "xs_icp_mailbox_free": (C.c_int, [_vp, C.c_int]),
    "xs_icp_accumulate_posted": (C.c_int, [_vp, C.c_uint, _vp, _vp, _f32p, _f32p, _f32p, _vp, _vp, _sz, C.c_int, C.c_int, C.c_float,
                                           C.c_float, C.c_int, C.c_int, _vp, _vp, _vp, C.c_ulonglong, _vp]),
    "xs_icp_post_pose": (None, [_vp, _f32p, _f32p, C.c_uint, C.c_int]),
    "xs_icp_accumulate_real": (C.c_int, [_f32p, _f32p, _vp, _vp, _vp, _vp, _sz, _f32p, _f32p, _f32p, _vp, _vp, _sz, C.c_int, C.c_int, C.c_float,
                                         C.c_float, C.c_int, C.c_int, _vp, _vp, _vp, C.c_ulonglong, _vp]),
    "xs_icp_accumulate_posted_real": (C.c_int, [_vp, C.c_uint, _vp, _vp, _vp, _vp, _sz, _f32p, _f32p, _f32p, _vp, _vp, _sz, C.c_int, C.c_int,
                                                C.c_float, C.c_float, C.c_int, C.c_int, _vp, _vp, _vp, C.c_ulonglong, _vp]),
    "xs_icp_records_bytes": (_sz, []),
    "xs_icp_records_count": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "xs_icp_accumulate_records": (C.c_int, [_f32p, _f32p, _vp, C.c_uint, _vp, _vp, _f32p, _f32p, _f32p, _vp, _vp, _sz, C.c_int, C.c_int, C.c_float,
                                            C.c_float, C.c_int, C.c_int, _vp, C.c_ulonglong, _vp]),
    "xs_icp_sum_records": (C.c_int, [_vp, C.c_int, C.c_ulonglong, _f64p, C.c_longlong]),
    "xs_icp_gate_selftest": (C.c_int, [_vp, C.c_int, C.c_float, C.c_int, _vp, _vp]),
    "xs_icp_iterate": (C.c_int, [_f32p, _f32p, _vp, _vp, _f32p, _f32p, _f32p, _vp, _vp, _sz, C.c_int, C.c_int, C.c_float,
                                 C.c_float, _vp, _vp, _vp, _vp, _vp, _vp, C.c_ulonglong, _vp]),
    "xs_estimate_combined": (C.c_int, [_f32p, _f32p, _vp, _vp, _f32p, _f32p, _f32p, _vp, _vp, _sz, C.c_int, C.c_int, C.c_float,
                                       C.c_float, _vp, _vp, _f64p, _f64p, C.POINTER(C.c_longlong), _vp]),
    "xs_icp_unpack": (None, [_f64p, _f64p, _f64p]),
    "xs_csfd_array_op": (C.c_int, [C.c_int, C.c_int, _vp, _vp, _vp, C.c_long, _vp]),
    "xs_dcsfd_f1": (C.c_int, [_vp, _vp, _vp, C.c_long, _vp]),
    "xs_complex_table": (C.c_int, [C.c_int, C.c_int, _vp, _vp, _vp, C.c_long, _vp]),
    "xs_integrate_pose_covered": (C.c_int, [C.c_int, C.c_int, _f32p, _i32p, C.c_float, _f32p, _f32p, C.c_float, _f32p, _f32p]),
    "xs_const_div_prepare": (C.c_uint, [C.c_float]),
    "xs_raycast_signmap_shift": (C.c_int, [_f32p, C.c_float, C.c_float]),
    "xs_signmap_bytes": (C.c_size_t, [_i32p, C.c_int]),
    "xs_signmap_reset": (C.c_int, [_vp, _i32p, C.c_int, C.c_float, _vp]),
    "xs_signmap_rebuild": (C.c_int, [_vp, _i32p, C.c_int, C.c_float, _vp, C.c_size_t, _vp]),
    "xs_signmap_rebuild_slab": (C.c_int, [_vp, _i32p, C.c_int, C.c_float, _vp, C.c_size_t, C.c_int, C.c_int, _vp]),
    "xs_const_div_state": (C.c_uint, [C.c_float]),
    "xs_const_div_enable": (C.c_int, [C.c_int]),
}


def _bind():
    for name, (res, args) in _SIGS.items():
        f = getattr(_lib, name)
        f.restype = res
        f.argtypes = args


_bind()


def const_div_prepare(c):
    """xs_const_div_prepare: the exhaustive device check that lets the kernels divide by the constant c with the short form."""
    return int(_lib.xs_const_div_prepare(float(c)))


def const_div_state(c):
    return int(_lib.xs_const_div_state(float(c)))


def const_div_enable(on):
    return int(_lib.xs_const_div_enable(int(bool(on))))


_prepared = set()


def _prep(*consts):
    """The wrappers below prepare the constants their kernels divide by, as the C++ orchestrator does in AllocateBuffers."""
    for c in consts:
        c = float(np.float32(c))
        if c not in _prepared:
            _prepared.add(c)
            _lib.xs_const_div_prepare(c)


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


def scale_depth_max(depth, depth_step, rows, cols, scaled, scaled_step, max_dev, stream=None):
    check(_lib.xs_scale_depth_max(_ptr(depth), depth_step, rows, cols, _ptr(scaled), scaled_step, _ptr(max_dev), _stream(stream)))


def depth_tiles_bytes(rows, cols):
    return _lib.xs_depth_tiles_bytes(rows, cols)


def scale_depth_tiles(depth, depth_step, rows, cols, scaled, scaled_step, max_dev, tiles, stream=None):
    """scale_depth_max + the per-tile depth range table {lo, hi} per 8 x 8 pixels (depth_tiles_bytes(rows, cols) bytes)."""
    check(_lib.xs_scale_depth_tiles(_ptr(depth), depth_step, rows, cols, _ptr(scaled), scaled_step, _ptr(max_dev), _ptr(tiles), _stream(stream)))


def depth_tiles(scaled, scaled_step, rows, cols, tiles, stream=None):
    check(_lib.xs_depth_tiles(_ptr(scaled), scaled_step, rows, cols, _ptr(tiles), _stream(stream)))


def integrate_scaled(depth_scaled, scaled_step, rows, cols, intr, max_weight, res, voxel_size, Rv2c, tv2c, tranc_dist, value, weight,
                     grad, vol_step, threshold=0.0, z0=0, z1=None, updated=None, depth_max=None, workspace=None, stream=None):
    r = _ia(res, 3)
    k, R, t = _fa(intr, 4), _fa(Rv2c, 18), _fa(tv2c, 6)
    z1 = int(r[2]) if z1 is None else z1
    check(_lib.xs_integrate_scaled(_ptr(depth_scaled), scaled_step, rows, cols, k.ctypes.data_as(_f32p), max_weight,
                                   r.ctypes.data_as(_i32p), voxel_size, R.ctypes.data_as(_f32p), t.ctypes.data_as(_f32p), tranc_dist,
                                   _ptr(value), _ptr(weight), _ptr(grad), vol_step, threshold, z0, z1, _ptr(updated), _ptr(depth_max),
                                   _ptr(workspace), _stream(stream)))


def integrate_scaled_ex(depth_scaled, scaled_step, rows, cols, intr, max_weight, res, voxel_size, Rv2c, tv2c, tranc_dist, value, weight,
                        grad, vol_step, flags, threshold=0.0, z0=0, z1=None, updated=None, depth_max=None, workspace=None, stream=None, **opts):
    """xs_integrate_scaled_ex2 with its options as keyword arguments (integrate_opts' fields: depth_tiles, signmap, start_event, stop_event,
    pose_mailbox, mailbox_seq, mailbox_slack, pose_dev) and the flags positional, as the tests write it."""
    return integrate_scaled_ex2(depth_scaled, scaled_step, rows, cols, intr, max_weight, res, voxel_size, Rv2c, tv2c, tranc_dist, value, weight, grad, vol_step,
                                integrate_opts(flags=flags, **opts), threshold=threshold, z0=z0, z1=z1, updated=updated, depth_max=depth_max,
                                workspace=workspace, stream=stream)


def integrate_opts(flags=0, depth_tiles=None, signmap=None, start_event=None, stop_event=None, pose_mailbox=None, mailbox_seq=0, mailbox_slack=2.0,
                   pose_dev=None):
    """An xs_integrate_opts for integrate_scaled_ex2 / integrate_classify_ex (tensors or raw addresses for the pointers)."""
    o = IntegrateOpts()
    o.struct_bytes = C.sizeof(IntegrateOpts)
    o.flags = flags
    o.depth_tiles, o.signmap, o.pose_mailbox, o.pose_dev = _ptr(depth_tiles), _ptr(signmap), _ptr(pose_mailbox), _ptr(pose_dev)
    o.start_event = start_event.value if hasattr(start_event, "value") else start_event
    o.stop_event = stop_event.value if hasattr(stop_event, "value") else stop_event
    o.mailbox_seq, o.mailbox_slack = mailbox_seq, mailbox_slack
    return o


def integrate_scaled_ex2(depth_scaled, scaled_step, rows, cols, intr, max_weight, res, voxel_size, Rv2c, tv2c, tranc_dist, value, weight,
                         grad, vol_step, opts, threshold=0.0, z0=0, z1=None, updated=None, depth_max=None, workspace=None, stream=None):
    """xs_integrate_scaled_ex2: everything the call needs in its arguments (opts: integrate_opts(...)); reads no per-thread state."""
    r = _ia(res, 3)
    k, R, t = _fa(intr, 4), _fa(Rv2c, 18), _fa(tv2c, 6)
    z1 = int(r[2]) if z1 is None else z1
    check(_lib.xs_integrate_scaled_ex2(_ptr(depth_scaled), scaled_step, rows, cols, k.ctypes.data_as(_f32p), max_weight,
                                       r.ctypes.data_as(_i32p), voxel_size, R.ctypes.data_as(_f32p), t.ctypes.data_as(_f32p), tranc_dist,
                                       _ptr(value), _ptr(weight), _ptr(grad), vol_step, threshold, z0, z1, _ptr(updated), _ptr(depth_max),
                                       _ptr(workspace), C.byref(opts), _stream(stream)))


def integrate_classify_ex(rows, cols, intr, res, voxel_size, Rv2c, tv2c, tranc_dist, workspace, opts, slack_scale=2.0, z0=0, z1=None,
                          depth_max=None, stream=None):
    r = _ia(res, 3)
    k, R, t = _fa(intr, 4), _fa(Rv2c, 18), _fa(tv2c, 6)
    check(_lib.xs_integrate_classify_ex(rows, cols, k.ctypes.data_as(_f32p), r.ctypes.data_as(_i32p), voxel_size, R.ctypes.data_as(_f32p),
                                        t.ctypes.data_as(_f32p), tranc_dist, z0, int(r[2]) if z1 is None else z1, _ptr(depth_max), _ptr(workspace),
                                        slack_scale, C.byref(opts), _stream(stream)))


def integrate_classify(rows, cols, intr, res, voxel_size, Rv2c, tv2c, tranc_dist, workspace, slack_scale=2.0, flags=0, z0=0, z1=None,
                       depth_max=None, stream=None, **opts):
    """The brick classification of an integrate call on its own, for a pose near the final one (xs_integrate_classify_ex; options — depth_tiles,
    stop_event — as keyword arguments)."""
    return integrate_classify_ex(rows, cols, intr, res, voxel_size, Rv2c, tv2c, tranc_dist, workspace, integrate_opts(flags=flags, **opts), slack_scale=slack_scale,
                                 z0=z0, z1=z1, depth_max=depth_max, stream=stream)


def integrate_list_covers(rows, cols, intr, res, voxel_size, Rv2c_list, tv2c_list, slack_scale, Rv2c, tv2c):
    r = _ia(res, 3)
    k = _fa(intr, 4)
    a, b, c, d = _fa(Rv2c_list, 18), _fa(tv2c_list, 6), _fa(Rv2c, 18), _fa(tv2c, 6)
    P = lambda x: x.ctypes.data_as(_f32p)
    return int(_lib.xs_integrate_list_covers(rows, cols, P(k), r.ctypes.data_as(_i32p), voxel_size, P(a), P(b), slack_scale, P(c), P(d)))


def integrate_pose_covered(rows, cols, intr, res, voxel_size, Rv2c_list, tv2c_list, slack_scale, Rv2c, tv2c):
    r = _ia(res, 3)
    k = _fa(intr, 4)
    a, b, c, d = _fa(Rv2c_list, 18), _fa(tv2c_list, 6), _fa(Rv2c, 18), _fa(tv2c, 6)
    P = lambda x: x.ctypes.data_as(_f32p)
    return bool(_lib.xs_integrate_pose_covered(rows, cols, P(k), r.ctypes.data_as(_i32p), voxel_size, P(a), P(b), slack_scale, P(c), P(d)))


def integrate_workspace_clear(workspace, stream=None):
    check(_lib.xs_integrate_workspace_clear(_ptr(workspace), _stream(stream)))


def integrate_fold_counts(workspace, updated, stream=None):
    check(_lib.xs_integrate_fold_counts(_ptr(workspace), _ptr(updated), _stream(stream)))


def integrate_workspace_bytes(res, nz=None):
    r = _ia(res, 3)
    return _lib.xs_integrate_workspace_bytes(r.ctypes.data_as(_i32p), int(r[2]) if nz is None else nz)


def integrate_listed(workspace):
    """(bricks with planes to walk, other bricks) of the list the last classification left in an integrate workspace (a torch uint8
    tensor): the two runs of the list (front / back of its region) — read back from the header, for tests and probes."""
    import torch
    pair = workspace[:8].view(torch.int32).cpu().numpy()
    return int(pair[0]), int(pair[1])


def integrate_list_layout(res, nz=None):
    """Byte offsets inside an integrate workspace: (list region, entries it holds, class words, second list region)."""
    r = _ia(res, 3)
    nz = int(r[2]) if nz is None else nz
    cap = -(-int(r[0]) // 32) * -(-int(r[1]) // 8) * -(-nz // 2)
    first = 256 + 8192 * 4     # the header and the update counts' room (a word per workgroup of the brick kernel) lie in front
    list_bytes = (first + cap * 4 + 255) // 256 * 256
    class_bytes = (cap * 16 + 255) // 256 * 256
    return first, cap, list_bytes, list_bytes + class_bytes + (1 << 20)


def tsdf_reduce_workspace_init(workspace, stream=None):
    """Zero the residual kernels' workspace ticket (once after allocation; torch.zeros does the same)."""
    check(_lib.xs_tsdf_reduce_workspace_init(_ptr(workspace), _stream(stream)))


def tsdf_reduce_workspace_bytes():
    return _lib.xs_tsdf_reduce_workspace_bytes()


def compute_local_tsdf_hessian(depth_scaled, scaled_step, rows, cols, intr, res, voxel_size, Rv2c, tv2c, tranc_dist, gt, workspace, out4,
                               volumes=None, z0=0, z1=None, stream=None):
    r = _ia(res, 3)
    k, R, t = _fa(intr, 4), _fa(Rv2c, 36), _fa(tv2c, 12)
    z1 = int(r[2]) if z1 is None else z1
    vols = [_ptr(v) for v in volumes] if volumes else [None] * 4
    check(_lib.xs_compute_local_tsdf_hessian(_ptr(depth_scaled), scaled_step, rows, cols, k.ctypes.data_as(_f32p), r.ctypes.data_as(_i32p),
                                             voxel_size, R.ctypes.data_as(_f32p), t.ctypes.data_as(_f32p), tranc_dist, _ptr(gt), vols[0],
                                             vols[1], vols[2], vols[3], z0, z1, _ptr(workspace), _ptr(out4), _stream(stream)))


def compute_local_tsdf_loss(depth_scaled, scaled_step, rows, cols, intr, res, voxel_size, Rv2c, tv2c, tranc_dist, gt, workspace, out2,
                            volumes=None, z0=0, z1=None, stream=None):
    r = _ia(res, 3)
    k, R, t = _fa(intr, 4), _fa(Rv2c, 9), _fa(tv2c, 3)
    z1 = int(r[2]) if z1 is None else z1
    vols = [_ptr(v) for v in volumes] if volumes else [None] * 2
    check(_lib.xs_compute_local_tsdf_loss(_ptr(depth_scaled), scaled_step, rows, cols, k.ctypes.data_as(_f32p), r.ctypes.data_as(_i32p),
                                          voxel_size, R.ctypes.data_as(_f32p), t.ctypes.data_as(_f32p), tranc_dist, _ptr(gt), vols[0],
                                          vols[1], z0, z1, _ptr(workspace), _ptr(out2), _stream(stream)))


def bilateral_filter(src, src_step, rows, cols, dst, dst_step, stream=None):
    check(_lib.xs_bilateral_filter(_ptr(src), src_step, rows, cols, _ptr(dst), dst_step, _stream(stream)))


def pyr_down(src, src_step, src_rows, src_cols, dst, dst_step, stream=None):
    check(_lib.xs_pyr_down(_ptr(src), src_step, src_rows, src_cols, _ptr(dst), dst_step, _stream(stream)))


def create_vmap(intr, depth, depth_step, rows, cols, vmap, vmap_step, stream=None):
    k = _fa(intr, 4)
    check(_lib.xs_create_vmap(k.ctypes.data_as(_f32p), _ptr(depth), depth_step, rows, cols, _ptr(vmap), vmap_step, _stream(stream)))


def create_vnmaps(intrs, depths, depth_steps, rows0, cols0, vmaps, nmaps, map_steps, stream=None, vreal=None, nreal=None, real_steps=None):
    """Vertex + normal maps of all pyramid levels in one launch.  intrs: per-level [fx, fy, cx, cy].  vreal / nreal / real_steps:
    optionally the real parts again as float planes (xs_create_vnmaps_real)."""
    n = len(depths)
    k = np.ascontiguousarray(intrs, dtype=np.float32).reshape(-1)
    assert k.size == 4 * n
    P = lambda ts: (_vp * n)(*[_ptr(t) for t in ts])
    S = lambda xs: (_sz * n)(*[int(x) for x in xs])
    if vreal is None:
        check(_lib.xs_create_vnmaps(n, k.ctypes.data_as(_f32p), P(depths), S(depth_steps), rows0, cols0, P(vmaps), P(nmaps), S(map_steps),
                                    _stream(stream)))
    else:
        check(_lib.xs_create_vnmaps_real(n, k.ctypes.data_as(_f32p), P(depths), S(depth_steps), rows0, cols0, P(vmaps), P(nmaps), S(map_steps),
                                         P(vreal), P(nreal), S(real_steps), _stream(stream)))


def tsdf_gauss_newton_terms(depth_scaled, scaled_step, rows, cols, intr, res, voxel_size, Rv2c6, tv2c6, tranc_dist, gt, workspace, out29,
                            z0=0, z1=None, stream=None):
    r = _ia(res, 3)
    k, R, t = _fa(intr, 4), _fa(Rv2c6, 108), _fa(tv2c6, 36)
    z1 = int(r[2]) if z1 is None else z1
    check(_lib.xs_tsdf_gauss_newton_terms(_ptr(depth_scaled), scaled_step, rows, cols, k.ctypes.data_as(_f32p), r.ctypes.data_as(_i32p),
                                          voxel_size, R.ctypes.data_as(_f32p), t.ctypes.data_as(_f32p), tranc_dist, _ptr(gt), z0, z1,
                                          _ptr(workspace), _ptr(out29), _stream(stream)))


class GnOpts(C.Structure):
    """xs_gn_opts (include/xslam_amd.h)"""
    _fields_ = [("struct_bytes", C.c_uint), ("mailbox_seq", C.c_uint), ("pose_mailbox", C.c_void_p), ("publish_host", C.c_void_p),
                ("publish_seq", C.c_ulonglong)]


def tsdf_gauss_newton_terms_ex(depth_scaled, scaled_step, rows, cols, intr, res, voxel_size, Rv2c6, tv2c6, tranc_dist, gt, workspace, out29,
                               z0=0, z1=None, stream=None, pose_mailbox=None, mailbox_seq=0, publish_host=None, publish_seq=0):
    """xs_tsdf_gauss_newton_terms_ex: the pass with the loop protocol's options.  Rv2c6 / tv2c6 None: the launch takes its six poses from
    `pose_mailbox` (gn_post_poses); publish_host: a pinned host tensor of gn_publish_bytes() bytes the last workgroup writes."""
    r = _ia(res, 3)
    k = _fa(intr, 4)
    z1 = int(r[2]) if z1 is None else z1
    o = GnOpts()
    o.struct_bytes = C.sizeof(GnOpts)
    o.mailbox_seq, o.pose_mailbox, o.publish_host, o.publish_seq = mailbox_seq, _ptr(pose_mailbox), _ptr(publish_host), publish_seq
    if Rv2c6 is None:
        pR = pt = None
    else:
        R, t = _fa(Rv2c6, 108), _fa(tv2c6, 36)
        pR, pt = R.ctypes.data_as(_f32p), t.ctypes.data_as(_f32p)
    check(_lib.xs_tsdf_gauss_newton_terms_ex(_ptr(depth_scaled), scaled_step, rows, cols, k.ctypes.data_as(_f32p), r.ctypes.data_as(_i32p),
                                             voxel_size, pR, pt, tranc_dist, _ptr(gt), z0, z1, _ptr(workspace), _ptr(out29), C.byref(o),
                                             _stream(stream)))


def gn_publish_bytes():
    return int(_lib.xs_gn_publish_bytes())


def gn_post_poses(mailbox, Rv2c6, tv2c6, mailbox_seq, cmd=0):
    """Host side of the six-pose mailbox (an icp_mailbox_alloc address): cmd 0 = run with these poses, 1 = leave."""
    if Rv2c6 is None:
        _lib.xs_gn_post_poses(_ptr(mailbox), None, None, mailbox_seq, cmd)
    else:
        R, t = _fa(Rv2c6, 108), _fa(tv2c6, 36)
        _lib.xs_gn_post_poses(_ptr(mailbox), R.ctypes.data_as(_f32p), t.ctypes.data_as(_f32p), mailbox_seq, cmd)


def extract_workspace_bytes(res):
    r = _ia(res, 3)
    return _lib.xs_extract_workspace_bytes(r.ctypes.data_as(_i32p))


def extract_points(value, vol_step, res, voxel_size, points, capacity, workspace, zs0=0, z0=0, z1=None, stream=None):
    """extractPoints: fills points[capacity, 3] (CUDA float32); returns (stored, found).  Synchronises."""
    r = _ia(res, 3)
    z1 = int(r[2]) - 1 if z1 is None else z1
    cnt, found = _sz(0), _sz(0)
    check(_lib.xs_extract_points(_ptr(value), vol_step, r.ctypes.data_as(_i32p), voxel_size, zs0, z0, z1, _ptr(points), capacity,
                                 _ptr(workspace), C.byref(cnt), C.byref(found), _stream(stream)))
    return int(cnt.value), int(found.value)


def extract_normals(value, vol_step, res, voxel_size, points, n, normals, zs0=0, zs1=None, stream=None):
    r = _ia(res, 3)
    zs1 = int(r[2]) if zs1 is None else zs1
    check(_lib.xs_extract_normals(_ptr(value), vol_step, r.ctypes.data_as(_i32p), voxel_size, zs0, zs1, _ptr(points), n, _ptr(normals),
                                  _stream(stream)))


def resize_pyramid(vmap0, nmap0, in_step, rows0, cols0, vmap1, nmap1, mid_step, vmap2, nmap2, out_step, stream=None):
    check(_lib.xs_resize_pyramid(_ptr(vmap0), _ptr(nmap0), in_step, rows0, cols0, _ptr(vmap1), _ptr(nmap1), mid_step, _ptr(vmap2), _ptr(nmap2),
                                 out_step, _stream(stream)))


def create_nmap(vmap, nmap, map_step, rows, cols, stream=None):
    check(_lib.xs_create_nmap(_ptr(vmap), _ptr(nmap), map_step, rows, cols, _stream(stream)))


def resize_vmap(src, src_step, src_rows, src_cols, dst, dst_step, stream=None):
    check(_lib.xs_resize_vmap(_ptr(src), src_step, src_rows, src_cols, _ptr(dst), dst_step, _stream(stream)))


def resize_nmap(src, src_step, src_rows, src_cols, dst, dst_step, stream=None):
    check(_lib.xs_resize_nmap(_ptr(src), src_step, src_rows, src_cols, _ptr(dst), dst_step, _stream(stream)))


def raycast(intr, Rc2v, tc2v, Rv2w, tv2w, tranc_dist, res, voxel_size, value, grad, vol_step, vmap, nmap, map_step, rows, cols,
            hits=None, workspace=None, stream=None, **opts):
    """xs_raycast; with options (signmap, signmap_shift, signmap_tranc_dist, pyramid, completion_event, steps) xs_raycast_ex — returns the options
    struct then (its pyramid_built field says whether the call built the pyramid)."""
    if opts:
        o = _ray_opts(opts)
        raycast_ex(intr, Rc2v, tc2v, Rv2w, tv2w, tranc_dist, res, voxel_size, value, grad, vol_step, vmap, nmap, map_step, rows, cols, o, hits=hits,
                   workspace=workspace, stream=stream)
        return o
    r = _ia(res, 3)
    _prep(voxel_size)
    k, a, b, c, d = _fa(intr, 4), _fa(Rc2v, 18), _fa(tc2v, 6), _fa(Rv2w, 18), _fa(tv2w, 6)
    P = lambda x: x.ctypes.data_as(_f32p)
    check(_lib.xs_raycast(P(k), P(a), P(b), P(c), P(d), tranc_dist, r.ctypes.data_as(_i32p), voxel_size, _ptr(value), _ptr(grad),
                          vol_step, _ptr(vmap), _ptr(nmap), map_step, rows, cols, _ptr(hits), _ptr(workspace), _stream(stream)))


def _ray_opts(kw):
    """raycast_opts from the keyword options of raycast / raycast_slab (whose own tranc_dist argument is the march's)."""
    kw = dict(kw)
    if "signmap_shift" in kw:
        kw["shift"] = kw.pop("signmap_shift")
    if "signmap_tranc_dist" in kw:
        kw["tranc_dist"] = kw.pop("signmap_tranc_dist")
    return raycast_opts(**kw)


def raycast_opts(signmap=None, shift=3, tranc_dist=0.0, pyramid=None, completion_event=None, steps=None):
    """An xs_raycast_opts; pyramid = (vmap1, nmap1, step1, vmap2, nmap2, step2) or None."""
    o = RaycastOpts()
    o.struct_bytes = C.sizeof(RaycastOpts)
    o.signmap, o.signmap_shift, o.signmap_tranc_dist = _ptr(signmap), shift, tranc_dist
    if pyramid is not None:
        o.pyr_vmap1, o.pyr_nmap1, o.pyr_step1, o.pyr_vmap2, o.pyr_nmap2, o.pyr_step2 = (_ptr(pyramid[0]), _ptr(pyramid[1]), pyramid[2],
                                                                                       _ptr(pyramid[3]), _ptr(pyramid[4]), pyramid[5])
    o.completion_event = completion_event.value if hasattr(completion_event, "value") else completion_event
    o.steps_dev = _ptr(steps)
    return o


def raycast_ex(intr, Rc2v, tc2v, Rv2w, tv2w, tranc_dist, res, voxel_size, value, grad, vol_step, vmap, nmap, map_step, rows, cols, opts,
               hits=None, workspace=None, stream=None):
    """xs_raycast_ex: xs_raycast with its options as an argument (opts: raycast_opts(...); opts.pyramid_built is set by the call)."""
    _prep(voxel_size)
    r = _ia(res, 3)
    k = _fa(intr, 4)
    a, b, c, d = _fa(Rc2v, 18), _fa(tc2v, 6), _fa(Rv2w, 18), _fa(tv2w, 6)
    P = lambda x: x.ctypes.data_as(_f32p)
    check(_lib.xs_raycast_ex(P(k), P(a), P(b), P(c), P(d), tranc_dist, r.ctypes.data_as(_i32p), voxel_size, _ptr(value), _ptr(grad), vol_step,
                             _ptr(vmap), _ptr(nmap), map_step, rows, cols, _ptr(hits), _ptr(workspace), C.byref(opts), _stream(stream)))


def raycast_signmap_shift(intr, voxel_size, tranc_dist):
    """xs_raycast_signmap_shift: the finest sign map the march can use for this configuration (log2 of the brick edge in voxels), 0 if none."""
    return int(_lib.xs_raycast_signmap_shift(_fa(intr, 4).ctypes.data_as(_f32p), voxel_size, tranc_dist))


def signmap_bytes(res, shift=3):
    """xs_signmap_bytes: device bytes of a sign map (one byte per brick of 2^shift voxels a side, twice, + the march's time table)."""
    return int(_lib.xs_signmap_bytes(_ia(res, 3).ctypes.data_as(_i32p), shift))


def signmap_reset(signmap, res, shift, tranc_dist, stream=None):
    check(_lib.xs_signmap_reset(_ptr(signmap), _ia(res, 3).ctypes.data_as(_i32p), shift, tranc_dist, _stream(stream)))


def signmap_rebuild(signmap, res, shift, tranc_dist, value, vol_step, stream=None):
    check(_lib.xs_signmap_rebuild(_ptr(signmap), _ia(res, 3).ctypes.data_as(_i32p), shift, tranc_dist, _ptr(value), vol_step, _stream(stream)))


def raycast_slab(intr, Rc2v, tc2v, Rv2w, tv2w, tranc_dist, res, voxel_size, value, grad, vol_step, zs0, zs1, z0, z1, vmap, nmap,
                 map_step, rows, cols, keys, stream=None, **opts):
    """xs_raycast_slab; with options (signmap, signmap_shift, signmap_tranc_dist) xs_raycast_slab_ex."""
    r = _ia(res, 3)
    _prep(voxel_size)
    k, a, b, c, d = _fa(intr, 4), _fa(Rc2v, 18), _fa(tc2v, 6), _fa(Rv2w, 18), _fa(tv2w, 6)
    P = lambda x: x.ctypes.data_as(_f32p)
    if opts:
        o = _ray_opts(opts)
        check(_lib.xs_raycast_slab_ex(P(k), P(a), P(b), P(c), P(d), tranc_dist, r.ctypes.data_as(_i32p), voxel_size, _ptr(value), _ptr(grad),
                                      vol_step, zs0, zs1, z0, z1, _ptr(vmap), _ptr(nmap), map_step, rows, cols, _ptr(keys), C.byref(o), _stream(stream)))
        return o
    check(_lib.xs_raycast_slab(P(k), P(a), P(b), P(c), P(d), tranc_dist, r.ctypes.data_as(_i32p), voxel_size, _ptr(value), _ptr(grad),
                               vol_step, zs0, zs1, z0, z1, _ptr(vmap), _ptr(nmap), map_step, rows, cols, _ptr(keys), _stream(stream)))


def raycast_compose_mask(own_keys, min_keys, vmap, nmap, map_step, rows, cols, stream=None):
    check(_lib.xs_raycast_compose_mask(_ptr(own_keys), _ptr(min_keys), _ptr(vmap), _ptr(nmap), map_step, rows, cols, _stream(stream)))


def raycast_compose_finish(min_keys, vmap, nmap, map_step, rows, cols, hits=None, stream=None):
    check(_lib.xs_raycast_compose_finish(_ptr(min_keys), _ptr(vmap), _ptr(nmap), map_step, rows, cols, _ptr(hits), _stream(stream)))


def raycast_compose_entry_bytes():
    return _lib.xs_raycast_compose_entry_bytes()


def raycast_compose_pack(own_keys, min_keys, vmap, nmap, map_step, rows, cols, entries, count, stream=None):
    """The pixels this rank owns with a vertex, packed (52-byte entries); *count (int32, zeroed by the caller) advances by their number."""
    check(_lib.xs_raycast_compose_pack(_ptr(own_keys), _ptr(min_keys), _ptr(vmap), _ptr(nmap), map_step, rows, cols, _ptr(entries), _ptr(count), _stream(stream)))


def raycast_compose_scatter(entries, n, vmap, nmap, map_step, rows, cols, stream=None):
    check(_lib.xs_raycast_compose_scatter(_ptr(entries), int(n), _ptr(vmap), _ptr(nmap), map_step, rows, cols, _stream(stream)))


def icp_workspace_bytes():
    return _lib.xs_icp_workspace_bytes()


def icp_workspace_init(workspace, stream=None):
    check(_lib.xs_icp_workspace_init(_ptr(workspace), _stream(stream)))


ICP_PUBLISH_PAIRS = 1      # XS_ICP_PUBLISH_PAIRS: pass as done_flag; `sums` is then ICP_PAIRS_BYTES of host-coherent pinned memory (icp_wait_pairs)
ICP_PAIRS_BYTES = 55 * 16


def icp_accumulate(Rcurr, tcurr, vmap_curr, nmap_curr, Rprev_inv, tprev, intr, vmap_g_prev, nmap_g_prev, map_step, rows, cols,
                   distThres, angleThres, workspace, sums, y0=0, y1=None, stream=None, done_flag=None, done_seq=0):
    a, b, c, d, k = _fa(Rcurr, 18), _fa(tcurr, 6), _fa(Rprev_inv, 18), _fa(tprev, 6), _fa(intr, 4)
    P = lambda x: x.ctypes.data_as(_f32p)
    y1 = rows if y1 is None else y1
    check(_lib.xs_icp_accumulate(P(a), P(b), _ptr(vmap_curr), _ptr(nmap_curr), P(c), P(d), P(k), _ptr(vmap_g_prev), _ptr(nmap_g_prev),
                                 map_step, rows, cols, distThres, angleThres, y0, y1, _ptr(workspace), _ptr(sums), _ptr(done_flag), done_seq,
                                 _stream(stream)))


def icp_wait_pairs(pairs, seq, max_spins=2_000_000_000):
    """(rc, sums[55]): spins until all 55 pairs of the pinned buffer carry seq; rc 1 = the launch gave up, 2 = nothing came."""
    out = np.zeros(55, np.float64)
    rc = _lib.xs_icp_wait_pairs(_ptr(pairs), seq, out.ctypes.data, max_spins)
    return rc, out


def icp_accumulate_real(Rcurr, tcurr, vmap_curr_real, nmap_curr_real, real_step, Rprev_inv, tprev, intr, vmap_g_prev, nmap_g_prev, map_step, rows,
                        cols, distThres, angleThres, workspace, sums, y0=0, y1=None, stream=None, vmap_curr=None, nmap_curr=None):
    """icp_accumulate reading the current-frame maps' real parts from float planes (create_vnmaps(..., vreal=, nreal=))."""
    a, b, c, d, k = _fa(Rcurr, 18), _fa(tcurr, 6), _fa(Rprev_inv, 18), _fa(tprev, 6), _fa(intr, 4)
    P = lambda x: x.ctypes.data_as(_f32p)
    y1 = rows if y1 is None else y1
    check(_lib.xs_icp_accumulate_real(P(a), P(b), _ptr(vmap_curr), _ptr(nmap_curr), _ptr(vmap_curr_real), _ptr(nmap_curr_real), real_step, P(c), P(d),
                                      P(k), _ptr(vmap_g_prev), _ptr(nmap_g_prev), map_step, rows, cols, distThres, angleThres, y0, y1,
                                      _ptr(workspace), _ptr(sums), None, 0, _stream(stream)))


def icp_accumulate_posted_real(mailbox, mailbox_seq, vmap_curr_real, nmap_curr_real, real_step, Rprev_inv, tprev, intr, vmap_g_prev, nmap_g_prev,
                               map_step, rows, cols, distThres, angleThres, workspace, sums, y0=0, y1=None, stream=None, done_flag=None, done_seq=0):
    c, d, k = _fa(Rprev_inv, 18), _fa(tprev, 6), _fa(intr, 4)
    P = lambda x: x.ctypes.data_as(_f32p)
    check(_lib.xs_icp_accumulate_posted_real(_ptr(mailbox), mailbox_seq, None, None, _ptr(vmap_curr_real), _ptr(nmap_curr_real), real_step, P(c), P(d),
                                             P(k), _ptr(vmap_g_prev), _ptr(nmap_g_prev), map_step, rows, cols, distThres, angleThres, y0,
                                             rows if y1 is None else y1, _ptr(workspace), _ptr(sums), _ptr(done_flag), done_seq, _stream(stream)))


def icp_iterate(Rcurr, tcurr, vmap_curr, nmap_curr, Rprev_inv, tprev, intr, vmap_g_prev, nmap_g_prev, map_step, rows, cols,
                distThres, angleThres, workspace, sums, pose_state, stream=None):
    """One ICP iteration with the pose update on the device (xs_icp_iterate).  Rcurr / tcurr None:
    continue from the pose the previous launch left in pose_state (a 128-byte device tensor)."""
    c, d, k = _fa(Rprev_inv, 18), _fa(tprev, 6), _fa(intr, 4)
    P = lambda x: x.ctypes.data_as(_f32p)
    if Rcurr is None:
        pa = pb = None
    else:
        a, b = _fa(Rcurr, 18), _fa(tcurr, 6)
        pa, pb = P(a), P(b)
    check(_lib.xs_icp_iterate(pa, pb, _ptr(vmap_curr), _ptr(nmap_curr), P(c), P(d), P(k), _ptr(vmap_g_prev), _ptr(nmap_g_prev),
                              map_step, rows, cols, distThres, angleThres, _ptr(workspace), _ptr(sums), None, _ptr(pose_state), None, None, 0,
                              _stream(stream)))


def icp_accumulate_posted(mailbox, mailbox_seq, vmap_curr, nmap_curr, Rprev_inv, tprev, intr, vmap_g_prev, nmap_g_prev, map_step, rows,
                          cols, distThres, angleThres, workspace, sums, y0=0, y1=None, stream=None, done_flag=None, done_seq=0):
    """Enqueue an ICP reduction whose pose arrives later through `mailbox` (a 128-byte tensor in pinned
    host memory, see icp_post_pose).  The launch polls until the sequence number shows up — post it, or
    the launch gives up after about a second."""
    c, d, k = _fa(Rprev_inv, 18), _fa(tprev, 6), _fa(intr, 4)
    P = lambda x: x.ctypes.data_as(_f32p)
    check(_lib.xs_icp_accumulate_posted(_ptr(mailbox), mailbox_seq, _ptr(vmap_curr), _ptr(nmap_curr), P(c), P(d), P(k), _ptr(vmap_g_prev),
                                        _ptr(nmap_g_prev), map_step, rows, cols, distThres, angleThres, y0, rows if y1 is None else y1,
                                        _ptr(workspace), _ptr(sums), _ptr(done_flag), done_seq, _stream(stream)))


def icp_post_pose(mailbox, Rcurr, tcurr, mailbox_seq, cmd=0):
    """Host side of the pose mailbox: cmd 0 = run with this pose, 1 = abandon the launch."""
    P = lambda x: x.ctypes.data_as(_f32p)
    if Rcurr is None:
        _lib.xs_icp_post_pose(_ptr(mailbox), None, None, mailbox_seq, cmd)
    else:
        a, b = _fa(Rcurr, 18), _fa(tcurr, 6)
        _lib.xs_icp_post_pose(_ptr(mailbox), P(a), P(b), mailbox_seq, cmd)


def icp_records_bytes():
    return int(_lib.xs_icp_records_bytes())


def icp_records_count(cols, y0, y1):
    return int(_lib.xs_icp_records_count(cols, y0, y1))


def icp_accumulate_records(Rcurr, tcurr, vmap_curr, nmap_curr, Rprev_inv, tprev, intr, vmap_g_prev, nmap_g_prev, map_step, rows, cols, distThres,
                           angleThres, records, seq, y0=0, y1=None, mailbox=None, mailbox_seq=0, stream=None):
    """estimateCombined with the final addition on the host: enqueue the launch; every workgroup writes its record into `records`
    (a pinned host tensor of icp_records_bytes() bytes).  Rcurr / tcurr None: the pose is posted through `mailbox`."""
    c, d, k = _fa(Rprev_inv, 18), _fa(tprev, 6), _fa(intr, 4)
    P = lambda x: x.ctypes.data_as(_f32p)
    a, b = (None, None) if Rcurr is None else (P(_fa(Rcurr, 18)), P(_fa(tcurr, 6)))
    check(_lib.xs_icp_accumulate_records(a, b, _ptr(mailbox), mailbox_seq, _ptr(vmap_curr), _ptr(nmap_curr), P(c), P(d), P(k), _ptr(vmap_g_prev),
                                         _ptr(nmap_g_prev), map_step, rows, cols, distThres, angleThres, y0, rows if y1 is None else y1,
                                         _ptr(records), seq, _stream(stream)))


def icp_sum_records(records, count, seq, max_spins=2000000000):
    """Host: wait for the `count` records of launch `seq` and add them in index order.  (status, sums[55])."""
    out = np.zeros(55, np.float64)
    rc = _lib.xs_icp_sum_records(_ptr(records), count, seq, out.ctypes.data_as(_f64p), max_spins)
    return rc, out


def icp_gate_selftest(z, thres, or_equal, stream=None):
    """Gate shortcut of the ICP search against the full complex square root on the complex64 tensor `z` (device).
    Returns (disagreements, values that needed the square root)."""
    import torch
    zz = torch.view_as_real(z.contiguous()).contiguous()
    counts = torch.zeros(2, dtype=torch.int32, device=z.device)
    check(_lib.xs_icp_gate_selftest(_ptr(zz), z.numel(), float(thres), int(bool(or_equal)), _ptr(counts), _stream(stream)))
    torch.cuda.synchronize()
    c = counts.cpu().numpy().astype(np.int64)
    return int(c[0]), int(c[1])


def icp_mailbox_bytes():
    return int(_lib.xs_icp_mailbox_bytes())


def icp_mailbox_alloc():
    """(address, in_device_memory) of a zeroed pose mailbox; release with icp_mailbox_free."""
    p, dev = C.c_void_p(), C.c_int(0)
    check(_lib.xs_icp_mailbox_alloc(C.byref(p), C.byref(dev)))
    return p.value, dev.value


def icp_mailbox_free(address, in_device_memory):
    check(_lib.xs_icp_mailbox_free(address, in_device_memory))


def estimate_combined(Rcurr, tcurr, vmap_curr, nmap_curr, Rprev_inv, tprev, intr, vmap_g_prev, nmap_g_prev, map_step, rows, cols,
                      distThres, angleThres, workspace, sums, stream=None):
    """Returns (A[72], b[12], inliers) on the host after synchronising, like estimateCombined."""
    a, b, c, d, k = _fa(Rcurr, 18), _fa(tcurr, 6), _fa(Rprev_inv, 18), _fa(tprev, 6), _fa(intr, 4)
    P = lambda x: x.ctypes.data_as(_f32p)
    A = np.zeros(72, np.float64)
    bb = np.zeros(12, np.float64)
    inl = C.c_longlong(0)
    check(_lib.xs_estimate_combined(P(a), P(b), _ptr(vmap_curr), _ptr(nmap_curr), P(c), P(d), P(k), _ptr(vmap_g_prev), _ptr(nmap_g_prev),
                                    map_step, rows, cols, distThres, angleThres, _ptr(workspace), _ptr(sums), A.ctypes.data_as(_f64p),
                                    bb.ctypes.data_as(_f64p), C.byref(inl), _stream(stream)))
    return A, bb, inl.value


def icp_unpack(sums54):
    s = np.ascontiguousarray(sums54, dtype=np.float64)
    A = np.zeros(72, np.float64)
    b = np.zeros(12, np.float64)
    _lib.xs_icp_unpack(s.ctypes.data_as(_f64p), A.ctypes.data_as(_f64p), b.ctypes.data_as(_f64p))
    return A, b


CSFD_OPS = {"mul": 0, "div": 1, "exp": 2, "sin": 3, "pow": 4}


def csfd_array_op(name, variant, a, b, out, n, stream=None):
    check(_lib.xs_csfd_array_op(CSFD_OPS[name], 1 if variant == "our" else 0, _ptr(a), _ptr(b), _ptr(out), n, _stream(stream)))


def dcsfd_f1(x, y, out, n, stream=None):
    check(_lib.xs_dcsfd_f1(_ptr(x), _ptr(y), _ptr(out), n, _stream(stream)))


def complex_table(dual, op, a, b, out, n, stream=None):
    check(_lib.xs_complex_table(1 if dual else 0, op, _ptr(a), _ptr(b), _ptr(out), n, _stream(stream)))
