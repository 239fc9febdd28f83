"""ctypes binding of the host orchestrator's C ABI (include/xslam_amd_pipeline.h,
libxslam_host.so built from x-slam_amd/host/).  The orchestrator itself is C++
(KinectFusionReconstruction, mirroring the reference class); this module only lets Python
callers — bench.py and the parity tests — drive it.  No CPU fallback: a missing library is an
ImportError."""
import ctypes as C
import os

import numpy as np

from . import capi  # loads libxslam_hip.so first (libxslam_host.so links against it)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libxslam_host.so")
if not os.path.exists(LIB_PATH):
    raise ImportError(f"{LIB_PATH} not found: run __graft_entry__.build() (make -C x-slam_amd/host). There is no CPU fallback.")
_lib = C.CDLL(LIB_PATH)

_vp, _sz = C.c_void_p, C.c_size_t
_f32p, _f64p, _i32p = C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int)
COLLECTIVE_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_void_p, C.c_long)  # (user, op, dev_ptr, count)
_SIGS = {
    "xs_kf_create_sharded": (_vp, [C.c_char_p, C.c_int, C.c_int, COLLECTIVE_CB, _vp]),
    "xs_kf_shard_planes": (None, [_vp, _i32p, _i32p]),
    "xs_host_double_complex_table": (C.c_int, [C.c_int, C.c_long, _f32p, _f32p, _f32p]),
    "xs_host_complex_table": (C.c_int, [C.c_int, C.c_long, _f32p, _f32p, _f32p]),
    "xs_flat_yaml_get": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]),
    "xs_kf_set_stream": (None, [_vp]),
    "xs_kf_create": (_vp, [C.c_char_p]),
    "xs_kf_destroy": (None, [_vp]),
    "xs_kf_set_gt_poses": (None, [_vp, C.c_int, _f32p]),
    "xs_kf_process_frame": (C.c_int, [_vp, _vp, _sz]),
    "xs_kf_process_frame_host": (C.c_int, [_vp, _vp]),
    "xs_kf_ingest_buffer": (_vp, [_vp]),
    "xs_kf_get_camera2volume": (None, [_vp, _f32p]),
    "xs_kf_gauss_newton_terms": (C.c_int, [_vp, _vp, _sz, _f32p, _f64p]),
    "xs_kf_relocalize": (C.c_int, [_vp, _vp, _sz, _f32p, C.c_int, C.c_float, _f64p]),
    "xs_kf_export_point_cloud": (C.c_longlong, [_vp, C.c_int, _f32p, _f32p]),
    "xs_kf_export_ply": (C.c_longlong, [_vp, C.c_int, C.c_char_p]),
    "xs_kf_synchronize": (None, [_vp]),
    "xs_kf_frame_id": (C.c_int, [_vp]),
    "xs_kf_num_poses": (C.c_int, [_vp]),
    "xs_kf_get_world2camera": (None, [_vp, C.c_int, _f32p]),
    "xs_kf_tranc_dist": (C.c_float, [_vp]),
    "xs_kf_last_updated_voxels": (C.c_longlong, [_vp]),
    "xs_kf_last_raycast_hits": (C.c_longlong, [_vp]),
    "xs_kf_icp_log": (C.c_int, [_vp, _f64p, C.c_int]),
    "xs_kf_download_volume": (C.c_int, [_vp, _f32p, _i32p, _f32p]),
    "xs_kf_download_map": (C.c_int, [_vp, C.c_int, C.c_int, _f32p]),
    "xs_kf_volume_ptr": (_vp, [_vp, C.c_int, C.POINTER(_sz)]),
    "xs_kf_set_profiling": (None, [_vp, C.c_int]),
    "xs_kf_stage_times": (None, [_vp, _f64p, C.POINTER(C.c_longlong)]),
    "xs_kf_reset_stage_times": (None, [_vp]),
    "xs_kf_icp_iteration_times": (None, [_vp, _f64p, C.POINTER(C.c_longlong)]),
    "xs_kf_tail_host_times": (None, [_vp, _f64p, C.POINTER(C.c_longlong)]),
    "xs_kf_set_gn_post_pose": (None, [_vp, C.c_int]),
    "xs_kf_gn_poll_times": (None, [_vp, _f64p, C.POINTER(C.c_longlong), C.c_int]),
    "xs_kf_gn_times": (None, [_vp, _f64p, C.POINTER(C.c_longlong), _f64p, C.POINTER(C.c_longlong), C.c_int]),
    "xs_kf_debug_set_icp_sequence": (None, [_vp, C.c_ulonglong]),
    "xs_kf_debug_fail_icp_iteration": (None, [_vp, C.c_int]),
    "xs_kf_debug_post_delay": (None, [_vp, C.c_int, C.c_int]),
    "xs_kf_composite_bytes": (C.c_longlong, [_vp]),
    "xs_kf_rebuild_sign_map": (None, [_vp]),
    "xs_kf_hint_next_frame": (None, [_vp, _vp, _sz]),
    "xs_kf_posted_integrate_counts": (None, [_vp, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "xs_kf_list_cover_counts": (None, [_vp, C.POINTER(C.c_longlong)]),
    "xs_kf_cumulative_counters": (None, [_vp, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "xs_kf_save_checkpoint": (C.c_int, [_vp, C.c_char_p]),
    "xs_kf_load_checkpoint": (C.c_int, [_vp, C.c_char_p]),
    "xs_kf_save_tsdf_volume": (C.c_int, [_vp, C.c_char_p]),
    "xs_refshape_create": (_vp, [C.c_char_p]),
    "xs_refshape_destroy": (None, [_vp]),
    "xs_refshape_process_frame_host": (C.c_int, [_vp, _vp]),
    "xs_refshape_process_frame": (C.c_int, [_vp, _vp, _sz]),
    "xs_refshape_frame_id": (C.c_int, [_vp]),
    "xs_refshape_get_world2camera": (None, [_vp, C.c_int, _f32p]),
    "xs_refshape_download_volume": (C.c_int, [_vp, _f32p, _i32p, _f32p]),
    "xs_refshape_download_map": (C.c_int, [_vp, C.c_int, C.c_int, _f32p]),
    "xs_refshape_icp_log": (C.c_int, [_vp, _f64p, C.c_int]),
    "xs_refshape_hessian": (C.c_int, [_vp, _sz, C.c_int, C.c_int, _f32p, _i32p, C.c_float, _f32p, _f32p, C.c_float, _vp, C.c_int, _f32p, _f32p]),
    "xs_refshape_loss": (C.c_int, [_vp, _sz, C.c_int, C.c_int, _f32p, _i32p, C.c_float, _f32p, _f32p, C.c_float, _vp, C.c_int, _f32p, _f32p]),
}
for _n, (_r, _a) in _SIGS.items():
    _f = getattr(_lib, _n)
    _f.restype, _f.argtypes = _r, _a

STAGES = ("surface", "icp", "scale", "integrate", "raycast", "resize")
MAPS = {"depths_curr": 0, "vmaps_curr": 1, "nmaps_curr": 2, "vmaps_g_prev": 3, "nmaps_g_prev": 4}


def yaml_text(params: dict) -> str:
    """Flat YAML in the reference's format (ICL_traj2.yaml) from a dict of its keys."""
    out = []
    for k, v in params.items():
        if isinstance(v, bool):
            v = "true" if v else "false"
        elif isinstance(v, float):
            v = repr(float(np.float32(v))) if abs(v) < 1e-3 and v != 0 else repr(v)
        out.append(f"{k}: {v}")
    return "\n".join(out) + "\n"


HDC_OPS = {"add": 0, "sub": 1, "mul": 2, "div": 3, "sqrt": 4, "abs": 5, "exp": 6, "log": 7, "sin": 8, "cos": 9, "pow": 10, "f1": 11,
           "conj": 12, "norm": 13, "cmp": 14}


def host_double_complex(op, a, b=None):
    """Elementwise host DoubleComplex op over [n, 4] float32 arrays (CPU; no GPU involved)."""
    a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1, 4)
    b = a if b is None else np.ascontiguousarray(b, dtype=np.float32).reshape(-1, 4)
    out = np.empty_like(a)
    rc = _lib.xs_host_double_complex_table(HDC_OPS[op], a.shape[0], a.ctypes.data_as(_f32p), b.ctypes.data_as(_f32p), out.ctypes.data_as(_f32p))
    if rc != 0:
        raise ValueError("bad op")
    return out


def host_complex(op_code, a, b=None):
    """Elementwise complex<float> op of csrc/xs_complex.h compiled for the host, over [n, 2] float32 arrays (CPU)."""
    a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1, 2)
    b = a if b is None else np.ascontiguousarray(b, dtype=np.float32).reshape(-1, 2)
    out = np.empty_like(a)
    rc = _lib.xs_host_complex_table(int(op_code), a.shape[0], a.ctypes.data_as(_f32p), b.ctypes.data_as(_f32p), out.ctypes.data_as(_f32p))
    if rc != 0:
        raise ValueError("bad op")
    return out


def flat_yaml_get(text, key):
    buf = C.create_string_buffer(1024)
    n = _lib.xs_flat_yaml_get(text.encode(), key.encode(), buf, 1024)
    return None if n < 0 else buf.value.decode()


def set_stream(stream):
    _lib.xs_kf_set_stream(None if stream is None else (stream.cuda_stream if hasattr(stream, "cuda_stream") else int(stream)))


class KinectFusion:
    """Handle to a C++ KinectFusionReconstruction."""

    def __init__(self, params, gt_poses=None):
        text = params if isinstance(params, str) else yaml_text(params)
        self.cfg = {}
        for line in text.splitlines():
            if ":" in line:
                k, v = line.split(":", 1)
                self.cfg[k.strip()] = v.split("#")[0].strip()
        self.h = _lib.xs_kf_create(text.encode())
        if not self.h:
            raise ValueError("xs_kf_create failed (missing config key?)")
        self.res = [int(self.cfg[f"tsdf_size_{a}"]) for a in "xyz"]
        self.width, self.height = int(self.cfg["depth_width"]), int(self.cfg["depth_height"])
        if gt_poses is not None:
            g = np.ascontiguousarray(gt_poses, dtype=np.float32).reshape(-1, 32)
            _lib.xs_kf_set_gt_poses(self.h, g.shape[0], g.ctypes.data_as(_f32p))

    def close(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.xs_kf_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def process_frame(self, depth_dev, step_bytes=None):
        """depth_dev: torch int16/uint16 CUDA tensor [H, W] (u16 millimetres) or a raw device address."""
        ptr = depth_dev if isinstance(depth_dev, int) else depth_dev.data_ptr()
        step = step_bytes if step_bytes is not None else self.width * 2
        return _lib.xs_kf_process_frame(self.h, ptr, step)

    def hint_next_frame(self, depth_dev, step_bytes=None):
        """The device depth image the NEXT process_frame call will be given (unchanged until then): its maps are built
        during this frame's ICP loop.  Call before process_frame of the current frame."""
        _lib.xs_kf_hint_next_frame(self.h, depth_dev.data_ptr(), step_bytes or self.width * 2)

    def process_frame_host(self, depth_u16):
        """Host frame [H, W] u16: staged through pinned memory and copied asynchronously on the second stream
        (an array returned by ingest_buffer() is used in place)."""
        d = np.ascontiguousarray(depth_u16, dtype=np.uint16)
        return _lib.xs_kf_process_frame_host(self.h, d.ctypes.data)

    def ingest_buffer(self):
        """The next host-pinned staging buffer as a [H, W] uint16 array: fill it, then process_frame_host(it)."""
        ptr = _lib.xs_kf_ingest_buffer(self.h)
        n = self.width * self.height
        return np.ctypeslib.as_array((C.c_uint16 * n).from_address(ptr)).reshape(self.height, self.width)

    def camera2volume(self):
        out = np.zeros(32, np.float32)
        _lib.xs_kf_get_camera2volume(self.h, out.ctypes.data_as(_f32p))
        return out.reshape(4, 4, 2)

    def gauss_newton_terms(self, depth_dev, c2v):
        """29 doubles: J^T J upper triangle (21), J^T r (6), sum r^2, count, for the residual of the depth frame
        against the map at camera2volume c2v [4, 4, 2]."""
        m = np.ascontiguousarray(c2v, dtype=np.float32).reshape(32)
        out = np.zeros(29, np.float64)
        ok = _lib.xs_kf_gauss_newton_terms(self.h, depth_dev.data_ptr(), self.width * 2, m.ctypes.data_as(_f32p), out.ctypes.data_as(_f64p))
        return out if ok == 1 else None

    def relocalize(self, depth_dev, c2v, iterations=5, damping=1e-3):
        """Gauss-Newton refinement of camera2volume against the map: (ok, refined c2v [4, 4, 2], loss history)."""
        m = np.ascontiguousarray(c2v, dtype=np.float32).reshape(32).copy()
        hist = np.zeros(iterations + 1, np.float64)
        ok = _lib.xs_kf_relocalize(self.h, depth_dev.data_ptr(), self.width * 2, m.ctypes.data_as(_f32p), iterations, damping,
                                   hist.ctypes.data_as(_f64p))
        return ok == 1, m.reshape(4, 4, 2), hist

    def export_point_cloud(self, max_buffer=1000000):
        """ExportPointCloud: (points [n, 3], normals [n, 3]) float32 on the host."""
        p = np.zeros((max_buffer, 3), np.float32)
        nr = np.zeros((max_buffer, 3), np.float32)
        n = _lib.xs_kf_export_point_cloud(self.h, max_buffer, p.ctypes.data_as(_f32p), nr.ctypes.data_as(_f32p))
        return p[:n], nr[:n]

    def export_ply(self, filename, max_buffer=1000000):
        return _lib.xs_kf_export_ply(self.h, max_buffer, str(filename).encode())

    def synchronize(self):
        _lib.xs_kf_synchronize(self.h)

    @property
    def frame_id(self):
        return _lib.xs_kf_frame_id(self.h)

    def num_poses(self):
        return _lib.xs_kf_num_poses(self.h)

    def world2camera(self, idx=-1):
        out = np.zeros(32, np.float32)
        _lib.xs_kf_get_world2camera(self.h, idx, out.ctypes.data_as(_f32p))
        return out.reshape(4, 4, 2)

    def tranc_dist(self):
        return _lib.xs_kf_tranc_dist(self.h)

    def last_U(self):
        return _lib.xs_kf_last_updated_voxels(self.h)

    def last_hits(self):
        return _lib.xs_kf_last_raycast_hits(self.h)

    def icp_log(self):
        n = _lib.xs_kf_icp_log(self.h, None, 0)
        out = np.zeros(n, np.float64)
        if n:
            _lib.xs_kf_icp_log(self.h, out.ctypes.data_as(_f64p), n)
        return out.reshape(-1, 55)

    def volume(self):
        n = self.res[0] * self.res[1] * self.res[2]
        v, w, g = np.zeros(n, np.float32), np.zeros(n, np.int32), np.zeros(n, np.float32)
        _lib.xs_kf_download_volume(self.h, v.ctypes.data_as(_f32p), w.ctypes.data_as(_i32p), g.ctypes.data_as(_f32p))
        return v, w, g

    def map(self, which, level):
        rows, cols = self.height >> level, self.width >> level
        planes = 1 if which == "depths_curr" else 3
        out = np.zeros((planes * rows, cols, 2), np.float32)
        rc = _lib.xs_kf_download_map(self.h, MAPS[which], level, out.ctypes.data_as(_f32p))
        assert rc == 0
        return out

    def volume_ptr(self, which):
        step = _sz(0)
        p = _lib.xs_kf_volume_ptr(self.h, {"value": 0, "weight": 1, "grad": 2}[which], C.byref(step))
        return p, step.value

    def set_profiling(self, level=2):
        """0 off, 1 integrate-kernel events + counters only, 2 (True) every stage."""
        _lib.xs_kf_set_profiling(self.h, 2 if level is True else int(level))

    def stage_times(self):
        ms = np.zeros(6, np.float64)
        calls = (C.c_longlong * 6)()
        _lib.xs_kf_stage_times(self.h, ms.ctypes.data_as(_f64p), calls)
        return {s: (float(ms[i]), int(calls[i])) for i, s in enumerate(STAGES)}

    def cumulative_counters(self):
        u, h = C.c_longlong(0), C.c_longlong(0)
        _lib.xs_kf_cumulative_counters(self.h, C.byref(u), C.byref(h))
        return u.value, h.value

    def reset_stage_times(self):
        _lib.xs_kf_reset_stage_times(self.h)

    def icp_iteration_times(self):
        """{slot: (microseconds summed, iterations)} of the ICP loop's host-side iteration period since the last reset; slots 0..2 = pyramid
        level, 3 = the first iteration of each frame (which also waits for the previous frame's tail)."""
        us = np.zeros(4, np.float64)
        calls = (C.c_longlong * 4)()
        _lib.xs_kf_icp_iteration_times(self.h, us.ctypes.data_as(_f64p), calls)
        return {lv: (float(us[lv]), int(calls[lv])) for lv in range(4)}

    def tail_host_times(self):
        """Mean host microseconds per frame of the tail's four host stretches since the last reset (xs_kf_tail_host_times)."""
        us = np.zeros(4, np.float64)
        n = C.c_longlong(0)
        _lib.xs_kf_tail_host_times(self.h, us.ctypes.data_as(_f64p), C.byref(n))
        k = max(int(n.value), 1)
        return {"frames": int(n.value), "sums_seen_to_integrate_entered": round(float(us[0]) / k, 2), "entered_to_launch_call": round(float(us[1]) / k, 2),
                "integrate_launch_call": round(float(us[2]) / k, 2), "launch_returned_to_raycast_launched": round(float(us[3]) / k, 2)}

    def gn_poll_times(self, reset=False):
        """Mean microseconds a Gauss-Newton kernel that was enqueued ahead waited for its poses (the device's own clock), and how many such passes."""
        pu, n = C.c_double(0), C.c_longlong(0)
        _lib.xs_kf_gn_poll_times(self.h, C.byref(pu), C.byref(n), int(reset))
        return {"passes": int(n.value), "wait_us": pu.value / max(n.value, 1)}

    def set_gn_post_pose(self, on):
        _lib.xs_kf_set_gn_post_pose(self.h, int(bool(on)))

    def gn_times(self, reset=False):
        """Gauss-Newton passes run by relocalize since the last reset: mean wall clock per pass and (profiling on) mean kernel duration, microseconds."""
        pu, km = C.c_double(0), C.c_double(0)
        n, kc = C.c_longlong(0), C.c_longlong(0)
        _lib.xs_kf_gn_times(self.h, C.byref(pu), C.byref(n), C.byref(km), C.byref(kc), int(reset))
        return {"passes": int(n.value), "pass_us": pu.value / max(n.value, 1), "kernel_calls": int(kc.value), "kernel_us": 1e3 * km.value / max(kc.value, 1)}

    def debug_set_icp_sequence(self, v):
        _lib.xs_kf_debug_set_icp_sequence(self.h, int(v))

    def debug_fail_icp_iteration(self, n):
        _lib.xs_kf_debug_fail_icp_iteration(self.h, int(n))

    def debug_post_delay(self, min_us, max_us):
        """Test aid: a random host sleep of [min_us, max_us] microseconds in front of every ICP pose post."""
        _lib.xs_kf_debug_post_delay(self.h, int(min_us), int(max_us))

    def composite_bytes(self):
        """Shard mode: bytes this rank received through the raycast composite's collectives so far (see xs_kf_composite_bytes)."""
        return int(_lib.xs_kf_composite_bytes(self.h))

    def list_cover_counts(self):
        """Frames whose list / box classes decided ahead held for the final pose: {"neither": n, "list_only": n, "both": n}."""
        c = (C.c_longlong * 4)()
        _lib.xs_kf_list_cover_counts(self.h, c)
        return {"neither": int(c[0]), "list_only": int(c[1]), "both": int(c[3])}

    def posted_integrate_counts(self):
        """(accepted, refused) posted integrate launches so far (integrate_post_pose)."""
        a, r = C.c_longlong(0), C.c_longlong(0)
        _lib.xs_kf_posted_integrate_counts(self.h, C.byref(a), C.byref(r))
        return int(a.value), int(r.value)

    def rebuild_sign_map(self):
        """After writing the value array through volume_ptr: the ray march's sign map is rebuilt from the volume."""
        _lib.xs_kf_rebuild_sign_map(self.h)

    def save_checkpoint(self, path):
        _lib.xs_kf_save_checkpoint(self.h, path.encode())

    def load_checkpoint(self, path):
        return _lib.xs_kf_load_checkpoint(self.h, path.encode()) == 0

    def save_tsdf_volume(self, path):
        _lib.xs_kf_save_tsdf_volume(self.h, path.encode())


class ReferenceCallShape:
    """Handle to a C++ ReferenceCallShape (x-slam_amd/host/reference_shape.hpp): the reference's own per-frame call
    sequence over the reference-signature launchers alone, a stream drain where the reference drains the device."""

    def __init__(self, params):
        text = params if isinstance(params, str) else yaml_text(params)
        cfg = {}
        for line in text.splitlines():
            if ":" in line:
                k, v = line.split(":", 1)
                cfg[k.strip()] = v.split("#")[0].strip()
        self.h = _lib.xs_refshape_create(text.encode())
        if not self.h:
            raise ValueError("xs_refshape_create failed (missing config key?)")
        self.res = [int(cfg[f"tsdf_size_{a}"]) for a in "xyz"]
        self.width, self.height = int(cfg["depth_width"]), int(cfg["depth_height"])

    def close(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.xs_refshape_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def process_frame(self, depth_dev, step_bytes=None):
        ptr = depth_dev if isinstance(depth_dev, int) else depth_dev.data_ptr()
        return _lib.xs_refshape_process_frame(self.h, ptr, step_bytes if step_bytes is not None else self.width * 2)

    def process_frame_host(self, depth_u16):
        d = np.ascontiguousarray(depth_u16, dtype=np.uint16)
        return _lib.xs_refshape_process_frame_host(self.h, d.ctypes.data)

    @property
    def frame_id(self):
        return _lib.xs_refshape_frame_id(self.h)

    def world2camera(self, idx=-1):
        out = np.zeros(32, np.float32)
        _lib.xs_refshape_get_world2camera(self.h, idx, out.ctypes.data_as(_f32p))
        return out.reshape(4, 4, 2)

    def volume(self):
        n = self.res[0] * self.res[1] * self.res[2]
        v, w, g = np.zeros(n, np.float32), np.zeros(n, np.int32), np.zeros(n, np.float32)
        _lib.xs_refshape_download_volume(self.h, v.ctypes.data_as(_f32p), w.ctypes.data_as(_i32p), g.ctypes.data_as(_f32p))
        return v, w, g

    def map(self, which, level):
        rows, cols = self.height >> level, self.width >> level
        planes = 1 if which == "depths_curr" else 3
        out = np.zeros((planes * rows, cols, 2), np.float32)
        assert _lib.xs_refshape_download_map(self.h, MAPS[which], level, out.ctypes.data_as(_f32p)) == 0
        return out

    def icp_log(self):
        n = _lib.xs_refshape_icp_log(self.h, None, 0)
        out = np.zeros(n, np.float64)
        if n:
            _lib.xs_refshape_icp_log(self.h, out.ctypes.data_as(_f64p), n)
        return out.reshape(-1, 54)


def reference_tsdf_hessian(depth_dev, rows, cols, intr4, res, voxel_size, R_dual, t_dual, tranc_dist, gt_dev, with_volumes=False):
    """ComputeLocalTsdf_hessian through its TsdfFusion.h:48-53 signature (xs_launchers.hpp): float4 {loss, gradient, second
    derivative, count} and, with_volumes, the per-voxel (real, grad, hessian, count) volumes."""
    k = np.ascontiguousarray(intr4, np.float32)
    r3 = np.ascontiguousarray(res, np.int32)
    R = np.ascontiguousarray(R_dual, np.float32).reshape(36)
    t = np.ascontiguousarray(t_dual, np.float32).reshape(12)
    n = int(r3[0]) * int(r3[1]) * int(r3[2])
    out = np.zeros(4, np.float32)
    vols = np.zeros(4 * n if with_volumes else 1, np.float32)
    rc = _lib.xs_refshape_hessian(depth_dev.data_ptr(), cols * 2, rows, cols, k.ctypes.data_as(_f32p), r3.ctypes.data_as(_i32p), voxel_size,
                                  R.ctypes.data_as(_f32p), t.ctypes.data_as(_f32p), tranc_dist, gt_dev.data_ptr(), int(with_volumes),
                                  out.ctypes.data_as(_f32p), vols.ctypes.data_as(_f32p) if with_volumes else None)
    assert rc == 0
    if not with_volumes:
        return out
    return out, (vols[:n], vols[n:2 * n], vols[2 * n:3 * n], vols[3 * n:].view(np.int32))


def reference_tsdf_loss(depth_dev, rows, cols, intr4, res, voxel_size, R9, t3, tranc_dist, gt_dev, with_volumes=False):
    """ComputeLocalTsdf_loss through its TsdfFusion.h:55-60 signature: float2 {loss, count} (+ the (real, count) volumes)."""
    k = np.ascontiguousarray(intr4, np.float32)
    r3 = np.ascontiguousarray(res, np.int32)
    R = np.ascontiguousarray(R9, np.float32).reshape(9)
    t = np.ascontiguousarray(t3, np.float32).reshape(3)
    n = int(r3[0]) * int(r3[1]) * int(r3[2])
    out = np.zeros(2, np.float32)
    vols = np.zeros(2 * n if with_volumes else 1, np.float32)
    rc = _lib.xs_refshape_loss(depth_dev.data_ptr(), cols * 2, rows, cols, k.ctypes.data_as(_f32p), r3.ctypes.data_as(_i32p), voxel_size,
                               R.ctypes.data_as(_f32p), t.ctypes.data_as(_f32p), tranc_dist, gt_dev.data_ptr(), int(with_volumes),
                               out.ctypes.data_as(_f32p), vols.ctypes.data_as(_f32p) if with_volumes else None)
    assert rc == 0
    if not with_volumes:
        return out
    return out, (vols[:n], vols[n:].view(np.int32))
