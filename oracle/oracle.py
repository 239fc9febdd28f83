"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes loader for the CPU restatement (oracle/liboracle.so) and, where it was
built, for oracle/_ref/liboracle_ref.so (the same kernel text instantiated over
the reference's own ``::complex<float>`` from cuda_complex.hpp).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package (x-slam_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int)
_u16p = C.POINTER(C.c_uint16)


class KfParams(C.Structure):
    """Mirror of oc::KfParams (oracle/oc_host.hpp): the 25 YAML keys + CSFD seed."""
    _fields_ = [
        ("tsdf_size", C.c_int * 3), ("tsdf_voxel_size", C.c_float), ("max_integration_weight", C.c_int),
        ("thres_range", C.c_float), ("init", C.c_float * 3), ("r_deg", C.c_float * 3),
        ("depth_width", C.c_int), ("depth_height", C.c_int),
        ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
        ("num_levels", C.c_int), ("distThres", C.c_float), ("angleThres_deg", C.c_float),
        ("biInterpolate_threshold", C.c_float), ("trunc_logistic_k", C.c_float),
        ("flag_use_gtPose", C.c_int), ("frame_step", C.c_int),
        ("seed_row", C.c_int), ("seed_col", C.c_int), ("seed_h", C.c_float),
    ]


def build(ref=False, quiet=True):
    """Compile the oracle (and oracle/_ref when /root/reference is present)."""
    targets = ["all"]
    if ref and os.path.isdir("/root/reference/DeviceArray/include"):
        targets.append("ref")
    subprocess.run(["make", "-C", _HERE] + targets, check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


def _p(a, ct):
    return a.ctypes.data_as(ct) if a is not None else None


class Oracle:
    """Thin numpy front end.  Complex arrays are float32 with a trailing (re, im) axis."""

    def __init__(self, ref=False):
        self.ref = ref
        path = os.path.join(_HERE, "_ref", "liboracle_ref.so") if ref else os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.lib = C.CDLL(path)
        self.pfx = "orcref_" if ref else "orc_"
        L = self.lib
        self._fn("num_threads", C.c_int, [])
        self._fn("set_num_threads", None, [C.c_int])
        self._fn("cop", C.c_int, [C.c_int, C.c_long, _f32p, _f32p, _f32p])
        self._fn("cop_f64", C.c_int, [C.c_int, C.c_long, _f64p, _f64p, _f64p])
        self._fn("dop", C.c_int, [C.c_int, C.c_long, _f32p, _f32p, _f32p])
        if not ref:
            self._fn("hdop", C.c_int, [C.c_int, C.c_long, _f32p, _f32p, _f32p])
            self._fn("csfd_op", C.c_int, [C.c_int, C.c_int, C.c_long, _f32p, _f32p, _f32p])
            self._fn("csfd_chain_rule", None, [C.c_float, C.c_float, _f32p])
        self._fn("init_volume", None, [_f32p, _i32p, _f32p, C.c_size_t, _i32p])
        self._fn("scale_depth", None, [_u16p, C.c_size_t, C.c_int, C.c_int, _f32p, C.c_size_t])
        self._fn("integrate", C.c_longlong, [_f32p, C.c_size_t, C.c_int, C.c_int, _f32p, _i32p, _f32p, C.c_size_t, _i32p,
                                              C.c_float, C.c_int, _f32p, _f32p, _f32p, C.c_float, C.c_float, C.c_int, C.c_int])
        self._fn("raycast", C.c_longlong, [_f32p, _f32p, _f32p, _f32p, _f32p, C.c_float, _i32p, C.c_float, _f32p, _f32p,
                                            C.c_size_t, _f32p, _f32p, C.c_size_t, C.c_int, C.c_int])
        self._fn("raycast_slab", C.c_longlong, [_f32p, _f32p, _f32p, _f32p, _f32p, C.c_float, _i32p, C.c_float, _f32p, _f32p, C.c_size_t,
                                                 C.c_int, C.c_int, C.c_int, C.c_int, _f32p, _f32p, C.c_size_t, C.c_int, C.c_int, _i32p])
        self._fn("bilateral", None, [_u16p, C.c_size_t, C.c_int, C.c_int, _f32p, C.c_size_t])
        self._fn("pyr_down", None, [_f32p, C.c_size_t, C.c_int, C.c_int, _f32p, C.c_size_t])
        self._fn("create_vmap", None, [_f32p, _f32p, C.c_size_t, C.c_int, C.c_int, _f32p, C.c_size_t])
        self._fn("create_nmap", None, [C.c_int, C.c_int, _f32p, _f32p, C.c_size_t])
        self._fn("resize_map", None, [C.c_int, C.c_int, C.c_int, _f32p, C.c_size_t, _f32p, C.c_size_t])
        self._fn("tsdf_gn_terms", None, [_f32p, C.c_size_t, C.c_int, C.c_int, _i32p, C.c_float, _f32p, _f32p, C.c_float, _f32p, _f32p,
                                         C.c_int, C.c_int, _f64p])
        self._fn("extract_points", C.c_longlong, [_f32p, C.c_size_t, _i32p, C.c_float, C.c_int, C.c_int, C.c_int, _f32p, C.c_longlong])
        self._fn("extract_normals", None, [_f32p, C.c_size_t, _i32p, C.c_float, _f32p, C.c_longlong, _f32p])
        self._fn("icp_combined", C.c_longlong, [_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_size_t,
                                                 C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, _f64p, _f64p, _f64p])
        self._fn("tsdf_hessian", None, [_f32p, C.c_size_t, C.c_int, C.c_int, _i32p, C.c_float, _f32p, _f32p, C.c_float, _f32p,
                                         _f32p, _f32p, _f32p, _f32p, _i32p, C.c_int, C.c_int, _f64p])
        self._fn("tsdf_loss", None, [_f32p, C.c_size_t, C.c_int, C.c_int, _i32p, C.c_float, _f32p, _f32p, C.c_float, _f32p,
                                      _f32p, _f32p, _i32p, C.c_int, C.c_int, _f64p])
        self._fn("m4_inverse", None, [_f32p, _f32p])
        self._fn("m4_mul", None, [_f32p, _f32p, _f32p])
        self._fn("m3_inverse", None, [_f32p, _f32p])
        self._fn("det6_real", C.c_double, [_f64p])
        self._fn("llt_solve6", None, [_f64p, _f64p, _f64p])
        self._fn("rinc", None, [_f32p, _f32p, _f32p, _f32p])
        self._fn("kf_create", C.c_void_p, [C.POINTER(KfParams)])
        self._fn("kf_destroy", None, [C.c_void_p])
        self._fn("kf_set_gt_poses", None, [C.c_void_p, C.c_int, _f32p])
        self._fn("kf_process_frame", C.c_int, [C.c_void_p, _u16p])
        self._fn("kf_frame_id", C.c_int, [C.c_void_p])
        self._fn("kf_num_poses", C.c_int, [C.c_void_p])
        self._fn("kf_get_world2camera", None, [C.c_void_p, C.c_int, _f32p])
        self._fn("kf_tranc_dist", C.c_float, [C.c_void_p])
        self._fn("kf_last_U", C.c_longlong, [C.c_void_p])
        self._fn("kf_last_hits", C.c_longlong, [C.c_void_p])
        self._fn("kf_icp_log_size", C.c_int, [C.c_void_p])
        self._fn("kf_icp_log", None, [C.c_void_p, _f64p])
        self._fn("kf_value", _f32p, [C.c_void_p])
        self._fn("kf_grad", _f32p, [C.c_void_p])
        self._fn("kf_weight", _i32p, [C.c_void_p])
        self._fn("kf_map", _f32p, [C.c_void_p, C.c_int, C.c_int])

    def _fn(self, name, restype, argtypes):
        f = getattr(self.lib, self.pfx + name)
        f.restype = restype
        f.argtypes = argtypes
        setattr(self, "_" + name, f)

    # ---- scalar tables -----------------------------------------------------
    COP = {"add": 0, "sub": 1, "mul": 2, "div": 3, "sqrt": 4, "abs": 5, "exp": 6, "log": 7, "pow": 8, "sin": 9, "cos": 10,
           "sinh": 11, "cosh": 12, "sin_new": 13, "sinh_new": 14, "norm": 15, "arg": 16, "conj": 17, "polar": 18,
           "div_scalar": 19, "scalar_div": 20, "mul_scalar": 21, "scalar_sub": 22, "proj": 23, "log10": 24, "tanh": 25, "tan": 26,
           "asinh": 27, "acosh": 28, "atanh": 29, "asin": 30, "acos": 31, "atan": 32}
    DOP = {"add": 0, "sub": 1, "mul": 2, "div": 3, "sqrt": 4, "abs": 5, "mul_scalar": 6, "div_scalar": 7, "add_scalar": 8,
           "scalar_sub": 9}
    HDOP = {"add": 0, "sub": 1, "mul": 2, "div": 3, "sqrt": 4, "abs": 5, "exp": 6, "log": 7, "sin": 8, "cos": 9, "pow": 10, "f1": 11}
    CSFD = {"mul": 0, "div": 1, "exp": 2, "sin": 3, "pow": 4}

    def _table(self, fn, op, a, b, width, dtype=np.float32, ct=_f32p):
        a = np.ascontiguousarray(a, dtype=dtype).reshape(-1, width)
        b = np.ascontiguousarray(b, dtype=dtype).reshape(-1, width)
        out = np.empty_like(a)
        rc = fn(op, a.shape[0], _p(a, ct), _p(b, ct), _p(out, ct))
        if rc != 0:
            raise ValueError("bad op")
        return out

    def cop(self, name, a, b=None):
        b = a if b is None else b
        return self._table(self._cop, self.COP[name], a, b, 2)

    def cop_f64(self, name, a, b=None):
        b = a if b is None else b
        return self._table(self._cop_f64, self.COP[name], a, b, 2, np.float64, _f64p)

    def dop(self, name, a, b=None):
        b = a if b is None else b
        return self._table(self._dop, self.DOP[name], a, b, 4)

    def hdop(self, name, a, b=None):
        b = a if b is None else b
        return self._table(self._hdop, self.HDOP[name], a, b, 4)

    def csfd_op(self, name, variant, a, b):
        a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1, 2)
        b = np.ascontiguousarray(b, dtype=np.float32).reshape(-1, 2)
        out = np.empty_like(a)
        self._csfd_op(self.CSFD[name], 1 if variant == "our" else 0, a.shape[0], _p(a, _f32p), _p(b, _f32p), _p(out, _f32p))
        return out

    def csfd_chain_rule(self, t0=0.5, h=1e-6):
        out = np.zeros(4, np.float32)
        self._csfd_chain_rule(t0, h, _p(out, _f32p))
        return out

    # ---- kernels (dense, unpitched numpy arrays) ---------------------------
    @staticmethod
    def _res(res):
        return np.ascontiguousarray(res, dtype=np.int32)

    def scale_depth(self, depth_u16):
        d = np.ascontiguousarray(depth_u16, dtype=np.uint16)
        rows, cols = d.shape
        out = np.empty((rows, cols), np.float32)
        self._scale_depth(_p(d, _u16p), cols * 2, rows, cols, _p(out, _f32p), cols * 4)
        return out

    def new_volume(self, res):
        n = int(res[0]) * int(res[1]) * int(res[2])
        return np.zeros(n, np.float32), np.zeros(n, np.int32), np.zeros(n, np.float32)

    def integrate(self, depth_scaled, value, weight, grad, res, tranc_dist, max_weight, Rv2c, tv2c, intr, voxel_size,
                  threshold=0.0, z0=0, z1=None):
        """In-place on value/weight/grad (flat arrays, voxel (x,y,z) at (z*Y+y)*X+x).  Returns U."""
        r = self._res(res)
        ds = np.ascontiguousarray(depth_scaled, dtype=np.float32)
        R = np.ascontiguousarray(Rv2c, dtype=np.float32).reshape(18)
        t = np.ascontiguousarray(tv2c, dtype=np.float32).reshape(6)
        k = np.ascontiguousarray(intr, dtype=np.float32)
        z1 = int(r[2]) if z1 is None else z1
        return self._integrate(_p(ds, _f32p), ds.shape[1] * 4, ds.shape[0], ds.shape[1], _p(value, _f32p), _p(weight, _i32p),
                               _p(grad, _f32p), int(r[0]) * 4, _p(r, _i32p), tranc_dist, max_weight, _p(R, _f32p), _p(t, _f32p),
                               _p(k, _f32p), voxel_size, threshold, z0, z1)

    def raycast(self, intr, Rc2v, tc2v, Rv2w, tv2w, tranc_dist, res, voxel_size, value, grad, rows, cols, vmap=None, nmap=None):
        r = self._res(res)
        k = np.ascontiguousarray(intr, dtype=np.float32)
        a = [np.ascontiguousarray(x, dtype=np.float32).reshape(-1) for x in (Rc2v, tc2v, Rv2w, tv2w)]
        vmap = np.zeros((3 * rows, cols, 2), np.float32) if vmap is None else vmap
        nmap = np.zeros((3 * rows, cols, 2), np.float32) if nmap is None else nmap
        hits = self._raycast(_p(k, _f32p), _p(a[0], _f32p), _p(a[1], _f32p), _p(a[2], _f32p), _p(a[3], _f32p), tranc_dist,
                             _p(r, _i32p), voxel_size, _p(value, _f32p), _p(grad, _f32p), int(r[0]) * 4, _p(vmap, _f32p),
                             _p(nmap, _f32p), cols * 8, rows, cols)
        return vmap, nmap, hits

    def raycast_slab(self, intr, Rc2v, tc2v, Rv2w, tv2w, tranc_dist, res, voxel_size, value, grad, rows, cols, stored, owned):
        """value / grad are full-size arrays of which only planes stored[0]..stored[1] may be read.
        Returns (vmap, nmap, keys, reads_outside_the_stored_planes)."""
        r = self._res(res)
        k = np.ascontiguousarray(intr, dtype=np.float32)
        a = [np.ascontiguousarray(x, dtype=np.float32).reshape(-1) for x in (Rc2v, tc2v, Rv2w, tv2w)]
        vmap = np.zeros((3 * rows, cols, 2), np.float32)
        nmap = np.zeros((3 * rows, cols, 2), np.float32)
        keys = np.zeros(rows * cols, np.int32)
        bad = self._raycast_slab(_p(k, _f32p), _p(a[0], _f32p), _p(a[1], _f32p), _p(a[2], _f32p), _p(a[3], _f32p), tranc_dist,
                                 _p(r, _i32p), voxel_size, _p(value, _f32p), _p(grad, _f32p), int(r[0]) * 4, stored[0], stored[1],
                                 owned[0], owned[1], _p(vmap, _f32p), _p(nmap, _f32p), cols * 8, rows, cols, _p(keys, _i32p))
        return vmap, nmap, keys, bad

    def bilateral(self, depth_u16):
        d = np.ascontiguousarray(depth_u16, dtype=np.uint16)
        rows, cols = d.shape
        out = np.zeros((rows, cols, 2), np.float32)
        self._bilateral(_p(d, _u16p), cols * 2, rows, cols, _p(out, _f32p), cols * 8)
        return out

    def pyr_down(self, src):
        src = np.ascontiguousarray(src, dtype=np.float32)
        rows, cols = src.shape[:2]
        out = np.zeros((rows // 2, cols // 2, 2), np.float32)
        self._pyr_down(_p(src, _f32p), cols * 8, rows, cols, _p(out, _f32p), (cols // 2) * 8)
        return out

    def create_vmap(self, intr, depth_c):
        d = np.ascontiguousarray(depth_c, dtype=np.float32)
        rows, cols = d.shape[:2]
        k = np.ascontiguousarray(intr, dtype=np.float32)
        out = np.zeros((3 * rows, cols, 2), np.float32)
        self._create_vmap(_p(k, _f32p), _p(d, _f32p), cols * 8, rows, cols, _p(out, _f32p), cols * 8)
        return out

    def create_nmap(self, vmap):
        v = np.ascontiguousarray(vmap, dtype=np.float32)
        rows, cols = v.shape[0] // 3, v.shape[1]
        out = np.zeros_like(v)
        self._create_nmap(rows, cols, _p(v, _f32p), _p(out, _f32p), cols * 8)
        return out

    def resize_map(self, m, normalize):
        m = np.ascontiguousarray(m, dtype=np.float32)
        srows, scols = m.shape[0] // 3, m.shape[1]
        out = np.zeros((3 * (srows // 2), scols // 2, 2), np.float32)
        self._resize_map(1 if normalize else 0, srows, scols, _p(m, _f32p), scols * 8, _p(out, _f32p), (scols // 2) * 8)
        return out

    def tsdf_gn_terms(self, depth_scaled, res, voxel_size, Rv2c6, tv2c6, tranc_dist, intr, gt, z0=0, z1=None):
        """29 Gauss-Newton sums for six seeded complex poses (Rv2c6: [6, 3, 3, 2], tv2c6: [6, 3, 2]); gt holds planes [z0, z1)."""
        res = self._res(res)
        ds = np.ascontiguousarray(depth_scaled, dtype=np.float32)
        R = np.ascontiguousarray(Rv2c6, dtype=np.float32).reshape(108)
        t = np.ascontiguousarray(tv2c6, dtype=np.float32).reshape(36)
        k = np.ascontiguousarray(intr, dtype=np.float32).reshape(4)
        gt = np.ascontiguousarray(gt, dtype=np.float32)
        z1 = int(res[2]) if z1 is None else z1
        out = np.zeros(29, np.float64)
        self._tsdf_gn_terms(_p(ds, _f32p), ds.shape[1] * 4, ds.shape[0], ds.shape[1], _p(res, _i32p), voxel_size, _p(R, _f32p), _p(t, _f32p),
                            tranc_dist, _p(k, _f32p), _p(gt, _f32p), z0, z1, _p(out, _f64p))
        return out

    def extract_points(self, value, res, voxel_size, z0=0, z1=None, capacity=None, zs0=0):
        """ExtractPointCloud.cu extractPoints on a dense value volume [(Z-zs0)*Y, X]: (points [n, 3], found)."""
        res = self._res(res)
        X = int(res[0])
        value = np.ascontiguousarray(value, dtype=np.float32)
        z1 = int(res[2]) - 1 if z1 is None else z1
        cap = int(capacity) if capacity is not None else 3 * value.size
        out = np.zeros((max(cap, 1), 3), np.float32)
        n = self._extract_points(_p(value, _f32p), X * 4, _p(res, _i32p), voxel_size, zs0, z0, z1, _p(out, _f32p), cap)
        return out[:min(n, cap)], int(n)

    def extract_normals(self, value, res, voxel_size, points):
        res = self._res(res)
        value = np.ascontiguousarray(value, dtype=np.float32)
        points = np.ascontiguousarray(points, dtype=np.float32)
        out = np.zeros_like(points)
        self._extract_normals(_p(value, _f32p), int(res[0]) * 4, _p(res, _i32p), voxel_size, _p(points, _f32p), points.shape[0], _p(out, _f32p))
        return out

    def icp_combined(self, Rcurr, tcurr, vmap_curr, nmap_curr, Rprev_inv, tprev, intr, vmap_g_prev, nmap_g_prev, distThres,
                     angleThres, y0=0, y1=None):
        rows, cols = vmap_curr.shape[0] // 3, vmap_curr.shape[1]
        y1 = rows if y1 is None else y1
        a = [np.ascontiguousarray(x, dtype=np.float32).reshape(-1) for x in (Rcurr, tcurr, Rprev_inv, tprev, intr)]
        maps = [np.ascontiguousarray(x, dtype=np.float32) for x in (vmap_curr, nmap_curr, vmap_g_prev, nmap_g_prev)]
        sums = np.zeros(54, np.float64)
        A = np.zeros(72, np.float64)
        b = np.zeros(12, np.float64)
        inl = self._icp_combined(_p(a[0], _f32p), _p(a[1], _f32p), _p(maps[0], _f32p), _p(maps[1], _f32p), _p(a[2], _f32p),
                                 _p(a[3], _f32p), _p(a[4], _f32p), _p(maps[2], _f32p), _p(maps[3], _f32p), cols * 8, rows, cols,
                                 distThres, angleThres, y0, y1, _p(sums, _f64p), _p(A, _f64p), _p(b, _f64p))
        return sums, A, b, inl

    def tsdf_hessian(self, depth_scaled, res, voxel_size, Rv2c, tv2c, tranc_dist, intr, gt, want_volumes=False, z0=0, z1=None):
        r = self._res(res)
        ds = np.ascontiguousarray(depth_scaled, dtype=np.float32)
        R = np.ascontiguousarray(Rv2c, dtype=np.float32).reshape(36)
        t = np.ascontiguousarray(tv2c, dtype=np.float32).reshape(12)
        k = np.ascontiguousarray(intr, dtype=np.float32)
        gt = np.ascontiguousarray(gt, dtype=np.float32)
        n = gt.size
        vols = [np.zeros(n, np.float32) for _ in range(3)] + [np.zeros(n, np.int32)] if want_volumes else [None] * 4
        out = np.zeros(4, np.float64)
        z1 = int(r[2]) if z1 is None else z1
        gp = _p(gt, _f32p)
        plane = int(r[0]) * int(r[1])
        if gt.size == (z1 - z0) * plane and gt.size != plane * int(r[2]):
            # gt holds the planes [z0, z1) only (as the C ABI takes a slab); the kernel indexes absolute planes and reads
            # none outside [z0, z1): hand it the address plane 0 would have
            assert not want_volumes
            gp = C.cast(C.c_void_p(gt.ctypes.data - z0 * plane * 4), _f32p)
        else:
            assert gt.size == plane * int(r[2]), "gt must hold the whole volume or exactly the planes [z0, z1)"
        self._tsdf_hessian(_p(ds, _f32p), ds.shape[1] * 4, ds.shape[0], ds.shape[1], _p(r, _i32p), voxel_size, _p(R, _f32p),
                           _p(t, _f32p), tranc_dist, _p(k, _f32p), gp, _p(vols[0], _f32p), _p(vols[1], _f32p),
                           _p(vols[2], _f32p), _p(vols[3], _i32p), z0, z1, _p(out, _f64p))
        return (out, vols) if want_volumes else out

    def tsdf_loss(self, depth_scaled, res, voxel_size, Rv2c, tv2c, tranc_dist, intr, gt, z0=0, z1=None):
        r = self._res(res)
        ds = np.ascontiguousarray(depth_scaled, dtype=np.float32)
        R = np.ascontiguousarray(Rv2c, dtype=np.float32).reshape(9)
        t = np.ascontiguousarray(tv2c, dtype=np.float32).reshape(3)
        k = np.ascontiguousarray(intr, dtype=np.float32)
        gt = np.ascontiguousarray(gt, dtype=np.float32)
        out = np.zeros(2, np.float64)
        z1 = int(r[2]) if z1 is None else z1
        self._tsdf_loss(_p(ds, _f32p), ds.shape[1] * 4, ds.shape[0], ds.shape[1], _p(r, _i32p), voxel_size, _p(R, _f32p),
                        _p(t, _f32p), tranc_dist, _p(k, _f32p), _p(gt, _f32p), None, None, z0, z1, _p(out, _f64p))
        return out

    # ---- host algebra ------------------------------------------------------
    def m4_inverse(self, m):
        m = np.ascontiguousarray(m, dtype=np.float32).reshape(32)
        out = np.zeros(32, np.float32)
        self._m4_inverse(_p(m, _f32p), _p(out, _f32p))
        return out.reshape(4, 4, 2)

    def m4_mul(self, a, b):
        a = np.ascontiguousarray(a, dtype=np.float32).reshape(32)
        b = np.ascontiguousarray(b, dtype=np.float32).reshape(32)
        out = np.zeros(32, np.float32)
        self._m4_mul(_p(a, _f32p), _p(b, _f32p), _p(out, _f32p))
        return out.reshape(4, 4, 2)

    def m3_inverse(self, m):
        m = np.ascontiguousarray(m, dtype=np.float32).reshape(18)
        out = np.zeros(18, np.float32)
        self._m3_inverse(_p(m, _f32p), _p(out, _f32p))
        return out.reshape(3, 3, 2)

    def det6_real(self, A):
        A = np.ascontiguousarray(A, dtype=np.float64).reshape(72)
        return self._det6_real(_p(A, _f64p))

    def llt_solve6(self, A, b):
        A = np.ascontiguousarray(A, dtype=np.float64).reshape(72)
        b = np.ascontiguousarray(b, dtype=np.float64).reshape(12)
        x = np.zeros(12, np.float64)
        self._llt_solve6(_p(A, _f64p), _p(b, _f64p), _p(x, _f64p))
        return x.reshape(6, 2)

    def rinc(self, alpha, beta, gamma):
        a = [np.ascontiguousarray(v, dtype=np.float32).reshape(2) for v in (alpha, beta, gamma)]
        out = np.zeros(18, np.float32)
        self._rinc(_p(a[0], _f32p), _p(a[1], _f32p), _p(a[2], _f32p), _p(out, _f32p))
        return out.reshape(3, 3, 2)


class OracleKinFu:
    """Per-frame pipeline restatement (oc::KinFu)."""

    def __init__(self, oracle, params: KfParams, gt_poses=None):
        self.o = oracle
        self.p = params
        self.h = oracle._kf_create(C.byref(params))
        if gt_poses is not None:
            g = np.ascontiguousarray(gt_poses, dtype=np.float32).reshape(-1, 32)
            oracle._kf_set_gt_poses(self.h, g.shape[0], _p(g, _f32p))

    def close(self):
        if self.h:
            self.o._kf_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def process_frame(self, depth_u16):
        d = np.ascontiguousarray(depth_u16, dtype=np.uint16)
        return self.o._kf_process_frame(self.h, _p(d, _u16p))

    @property
    def frame_id(self):
        return self.o._kf_frame_id(self.h)

    def world2camera(self, idx=-1):
        out = np.zeros(32, np.float32)
        self.o._kf_get_world2camera(self.h, idx, _p(out, _f32p))
        return out.reshape(4, 4, 2)

    def num_poses(self):
        return self.o._kf_num_poses(self.h)

    def tranc_dist(self):
        return self.o._kf_tranc_dist(self.h)

    def last_U(self):
        return self.o._kf_last_U(self.h)

    def last_hits(self):
        return self.o._kf_last_hits(self.h)

    def icp_log(self):
        n = self.o._kf_icp_log_size(self.h)
        out = np.zeros(n, np.float64)
        if n:
            self.o._kf_icp_log(self.h, _p(out, _f64p))
        return out.reshape(-1, 55)

    def _nvox(self):
        s = self.p.tsdf_size
        return s[0] * s[1] * s[2]

    def volume(self):
        n = self._nvox()
        v = np.ctypeslib.as_array(self.o._kf_value(self.h), (n,)).copy()
        g = np.ctypeslib.as_array(self.o._kf_grad(self.h), (n,)).copy()
        w = np.ctypeslib.as_array(self.o._kf_weight(self.h), (n,)).copy()
        return v, w, g

    def map(self, which, level):
        idx = {"depths_curr": 0, "vmaps_curr": 1, "nmaps_curr": 2, "vmaps_g_prev": 3, "nmaps_g_prev": 4}[which]
        rows, cols = self.p.depth_height >> level, self.p.depth_width >> level
        planes = 1 if idx == 0 else 3
        ptr = self.o._kf_map(self.h, idx, level)
        return np.ctypeslib.as_array(ptr, (planes * rows, cols, 2)).copy()


def params_from_dict(d) -> KfParams:
    """Reference YAML keys (KinectFusionReconstruction.cpp:12-72) + csfd_seed_* -> KfParams."""
    p = KfParams()
    p.tsdf_size[:] = [int(d["tsdf_size_x"]), int(d["tsdf_size_y"]), int(d["tsdf_size_z"])]
    p.tsdf_voxel_size = d["tsdf_voxel_size"]
    p.max_integration_weight = int(d["max_integration_weight"])
    p.thres_range = d["thres_range"]
    p.init[:] = [d["init_x"], d["init_y"], d["init_z"]]
    p.r_deg[:] = [d["r_x"], d["r_y"], d["r_z"]]
    p.depth_width, p.depth_height = int(d["depth_width"]), int(d["depth_height"])
    p.fx, p.fy, p.cx, p.cy = d["fx"], d["fy"], d["cx"], d["cy"]
    p.num_levels = int(d["num_levels"])
    p.distThres = d["distThres"]
    p.angleThres_deg = d["angleThres"]
    p.biInterpolate_threshold = d["biInterpolate_threshold"]
    p.trunc_logistic_k = d.get("trunc_logistic_k", 0.0)
    p.flag_use_gtPose = 1 if d.get("flag_use_gtPose", False) else 0
    p.frame_step = int(d.get("frame_step", 1))
    p.seed_row = int(d.get("csfd_seed_row", -1))
    p.seed_col = int(d.get("csfd_seed_col", -1))
    p.seed_h = d.get("csfd_seed_h", 1e-7)
    return p
