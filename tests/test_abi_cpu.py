"""CPU-only: the C-ABI shared libraries load and export every symbol the headers under
include/ declare (no compute calls without a GPU), and the ctypes tables bind them all."""
import ctypes
import importlib
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(xs_[a-z0-9_]+)\s*\(", text)))


@pytest.mark.parametrize("header,lib", [("xslam_amd.h", "libxslam_hip.so"), ("xslam_amd_pipeline.h", "libxslam_host.so"),
                                        ("xslam_amd_rccl.h", "libxslam_rccl.so")])
def test_every_declared_symbol_is_exported(header, lib):
    path = os.path.join(ROOT, "x-slam_amd", lib)
    assert os.path.exists(path), f"{lib} not built: run __graft_entry__.build()"
    if lib == "libxslam_host.so":
        ctypes.CDLL(os.path.join(ROOT, "x-slam_amd", "libxslam_hip.so"), mode=ctypes.RTLD_GLOBAL)
    h = ctypes.CDLL(path)
    names = declared(header)
    assert len(names) >= (10 if header == "xslam_amd_rccl.h" else 20)
    for n in names:
        assert hasattr(h, n), f"{n} declared in include/{header} but not exported by {lib}"


def test_ctypes_tables_cover_the_headers():
    capi = importlib.import_module("x-slam_amd.capi")
    pl = importlib.import_module("x-slam_amd.pipeline")
    assert set(declared("xslam_amd.h")) == set(capi._SIGS)
    assert set(declared("xslam_amd_pipeline.h")) == set(pl._SIGS)
    assert capi.abi_version() == 2
    assert capi.icp_workspace_bytes() > 0 and capi.tsdf_reduce_workspace_bytes() > 0
    assert capi.integrate_workspace_bytes([512, 512, 512]) == 256 + 8192 * 4 + 8 * 128 * 256 * 4 + 8 * 128 * 256 * 4 * 4 + (1 << 20) + 8 * 128 * 256 * 4   # header + update counts (a word per workgroup) + brick list + box classes (a word per wave-sized box) + the call's own depth-tile table + the list in the order it is taken
    assert capi.depth_tiles_bytes(480, 640) == (60 * 80 + 15 * 10) * 8
    # host-only entry point: unpacking the 27 sums into the symmetric system (ICP.cu:419-428)
    import numpy as np
    s = np.arange(54, dtype=np.float64)
    A, b = capi.icp_unpack(s)
    A = A.reshape(6, 6, 2)
    assert np.array_equal(A, A.transpose(1, 0, 2))
    assert A[0, 0, 0] == 0 and A[0, 1, 0] == 2 and b[0] == 12 and b[1] == 13 and A[5, 5, 0] == 50 and b[10] == 52


def test_pose_mailbox_layout_host_side():
    """xs_icp_post_pose is host code: the 128-byte mailbox it writes — four 32-byte sectors, each led by the sequence number, the
    command in word 1, the 18 floats of Rcurr then the 6 of tcurr in words 2..7, 9..15, 18..23 and 25..29 (csrc/xs_mailbox.h) — is what the
    posted kernels read; an abandon command carries no pose.  Both an aligned mailbox (MOVDIR64B direct stores where the CPU has them) and
    an unaligned one (the fenced form)."""
    import ctypes as C
    import numpy as np
    capi = importlib.import_module("x-slam_amd.capi")
    assert capi.icp_mailbox_bytes() == 128
    R = (np.arange(18, dtype=np.float32) + 1).reshape(3, 3, 2)
    t = (np.arange(6, dtype=np.float32) + 101).reshape(3, 2)
    f = np.concatenate([R.reshape(-1), t.reshape(-1)]).view(np.uint32)
    words = [2 + i for i in range(6)] + [9 + i for i in range(7)] + [18 + i for i in range(6)] + [25 + i for i in range(5)]
    raw = np.full(40 + 32, 0xDEADBEEF, np.uint32)                 # 32 words + guard, placed at a 64-byte aligned and at a 4-byte aligned address
    base = (-raw.ctypes.data // 4) % 16
    for off in (base, base + 1):
        raw[:] = 0xDEADBEEF
        buf = raw[off:off + 40]
        assert (buf.ctypes.data % 64 == 0) == (off == base)
        capi.icp_post_pose(buf.ctypes.data, R, t, 77, cmd=0)
        assert all(buf[w] == 77 for w in (0, 8, 16, 24)) and buf[1] == 0 and buf[17] == 0 and buf[30] == 0 and buf[31] == 0
        assert np.array_equal(buf[words], f)
        assert np.all(buf[32:] == 0xDEADBEEF) and np.all(raw[:off] == 0xDEADBEEF)
        capi.icp_post_pose(buf.ctypes.data, None, None, 78, cmd=1)
        assert all(buf[w] == 78 for w in (0, 8, 16, 24)) and buf[1] == 1 and not buf[words].any()


def test_host_fold_of_records_host_side():
    """xs_icp_sum_records is host code: records of 56 doubles (54 sums, count, sequence word) added in index order once each
    carries the launch's sequence number; a record that never arrives ends the wait (-1), one marked as given up returns 1."""
    import numpy as np
    capi = importlib.import_module("x-slam_amd.capi")
    rng = np.random.default_rng(5)
    n, seq = 7, 123456789
    rec = np.zeros((n, 56), np.float64)
    rec[:, :55] = rng.normal(size=(n, 55)) * 10.0 ** rng.integers(-8, 8, size=(n, 55))
    rec[:, 55] = np.array([seq] * n, np.uint64).view(np.float64)
    rc, sums = capi.icp_sum_records(rec.ctypes.data, n, seq, max_spins=10)
    want = np.zeros(55)
    for i in range(n):
        want += rec[i, :55]          # the same association: record 0 first
    assert rc == 0 and np.array_equal(sums, want)
    late = rec.copy(); late[4, 55] = np.array([seq - 1], np.uint64).view(np.float64)[0]
    assert capi.icp_sum_records(late.ctypes.data, n, seq, max_spins=1000)[0] == -1
    gone = rec.copy(); gone[2, 55] = np.array([seq | (1 << 63)], np.uint64).view(np.float64)[0]
    assert capi.icp_sum_records(gone.ctypes.data, n, seq, max_spins=1000)[0] == 1


def test_brick_list_coverage_host_side():
    """xs_integrate_list_covers is host code: a list classified for a pose with doubled slack covers the same pose, poses a
    last-ICP-update away, and not poses centimetres or a hundredth of a radian away."""
    import numpy as np
    capi = importlib.import_module("x-slam_amd.capi")
    synth = importlib.import_module("x-slam_amd.synth")
    prm = synth.s1_params(512)
    T = synth.s1_transforms(7, prm)
    res, vs, k4 = [512, 512, 512], prm["tsdf_voxel_size"], synth.intr_of(prm)
    R = np.array(T["Rv2c"], np.float32).reshape(3, 3, 2); t = np.array(T["tv2c"], np.float32).reshape(3, 2)
    cov = lambda R2, t2, s=2.0: capi.integrate_list_covers(synth.HEIGHT, synth.WIDTH, k4, res, vs, R, t, s, R2, t2)
    assert cov(R, t) and not cov(R, t, 1.0 - 1e-3)
    t2 = t.copy(); t2[:, 0] += [2e-4, -1e-4, 3e-4]
    assert cov(R, t2)
    t3 = t.copy(); t3[2, 0] += 0.08
    assert not cov(R, t3)
    a = 0.01
    Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], np.float32)
    R4 = R.copy(); R4[..., 0] = Rz @ R[..., 0]
    assert not cov(R4, t)


def test_product_never_imports_the_oracle():
    """The product package must not include, import, link or load anything under oracle/
    (the checker is test infrastructure; comments may mention it)."""
    pkg = os.path.join(ROOT, "x-slam_amd")
    bad = re.compile(r'#\s*include\s*["<][^">]*oracle|^\s*(from|import)\s+oracle|liboracle|oc_[a-z]+\.hpp', re.M)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hpp", ".hip", ".cpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not bad.search(text), os.path.join(dirpath, f)
    for lib in ("libxslam_hip.so", "libxslam_host.so"):
        blob = open(os.path.join(pkg, lib), "rb").read()
        assert b"liboracle" not in blob and b"orc_" not in blob


def test_host_double_complex_class_against_oracle(oracle):
    """x-slam_amd/host/DoubleComplex.h (the product's host dual-complex class, CPU code) against the
    oracle's restatement of DeviceArray/src/DoubleComplex.cpp and the test_CSFD known answers."""
    import json
    import numpy as np
    from conftest import GOLDEN, load_golden, ulp_diff
    pl = importlib.import_module("x-slam_amd.pipeline")
    t = load_golden("scalar_tables.npz")
    a, b, pos = t["d_a"], t["d_b"], t["d_pos"]
    for op in ("add", "sub", "mul"):
        assert ulp_diff(pl.host_double_complex(op, a, b), oracle.hdop(op, a, b)).max() == 0, op
    got, want = pl.host_double_complex("div", a, b), oracle.hdop("div", a, b)  # std::complex '/' is lowered differently by clang and gcc
    assert np.allclose(got[:, :3], want[:, :3], rtol=2e-5, atol=1e-12) and np.allclose(got[:, 3], want[:, 3], rtol=2e-4, atol=1e-18)
    for op in ("sqrt", "abs", "exp", "log", "sin", "cos"):
        got, want = pl.host_double_complex(op, pos), oracle.hdop(op, pos)
        assert np.allclose(got, want, rtol=2e-6, atol=1e-12), op  # clang vs gcc complex lowering: an ulp
    y = np.zeros_like(pos); y[:, 0] = 2.5
    assert np.allclose(pl.host_double_complex("pow", pos, y), oracle.hdop("pow", pos, y), rtol=5e-6, atol=1e-12)
    assert ulp_diff(pl.host_double_complex("f1", a, b), oracle.hdop("f1", a, b)).max() == 0
    c = pl.host_double_complex("cmp", a, b)
    assert np.array_equal(c[:, 0] == 1, a[:, 0] > b[:, 0]) and np.array_equal(c[:, 1] == 1, a[:, 0] < b[:, 0])
    # DCSFD of f(t) = (t^2 + sin t)^2 at t = 0.5 (test_CSFD part 2): gradient 2.73911, second derivative 9.26892
    ka = json.load(open(os.path.join(GOLDEN, "test_csfd_known_answers.json")))
    h = 1e-6
    tt = np.array([[0.5, h, h, 0.0]], np.float32)
    x = pl.host_double_complex("mul", tt, tt)
    yv = pl.host_double_complex("sin", tt)
    loss = pl.host_double_complex("f1", x, yv)[0]
    assert abs(loss[1] / h - ka["dcsfd_gradient"]) < 2e-4 and abs(loss[3] / h / h - ka["dcsfd_second"]) < 2e-2


def test_product_complex_header_on_the_host_against_reference_tables():
    """x-slam_amd/csrc/xs_complex.h compiled for the host (g++, glibc) against the tables the reference's own
    cuda_complex.hpp produced (tests/golden/scalar_tables.npz): every operator and function of the header,
    bit for bit — including the divide's and the square root's short forms and every special-value branch."""
    import numpy as np
    from conftest import load_golden, ulp_diff
    from oracle.oracle import Oracle
    pl = importlib.import_module("x-slam_amd.pipeline")
    t = load_golden("scalar_tables.npz")
    for tag in ("csfd", "gen", "wide"):
        for op in ("add", "sub", "mul", "div", "div_scalar", "scalar_div", "mul_scalar", "scalar_sub"):
            assert ulp_diff(pl.host_complex(Oracle.COP[op], t[f"{tag}_a"], t[f"{tag}_b"]), t[f"c_{tag}_{op}"]).max() == 0, (tag, op)
    for tag in ("csfd", "gen", "pos"):
        for op in ("sqrt", "abs", "exp", "sin", "cos", "sinh", "cosh", "sin_new", "sinh_new", "norm", "arg", "conj"):
            assert ulp_diff(pl.host_complex(Oracle.COP[op], t[f"{tag}_a"]), t[f"c_{tag}_{op}"]).max() == 0, (tag, op)
    assert ulp_diff(pl.host_complex(Oracle.COP["log"], t["pos_a"]), t["c_pos_log"]).max() == 0
    assert ulp_diff(pl.host_complex(Oracle.COP["log"], t["gen_a"]), t["c_gen_log"]).max() == 0
    assert ulp_diff(pl.host_complex(Oracle.COP["pow"], t["pos_a"], t["csfd_b"]), t["c_pos_pow"]).max() == 0
    assert ulp_diff(pl.host_complex(Oracle.COP["polar"], np.abs(t["gen_a"]), t["gen_b"]), t["c_gen_polar"]).max() == 0
    for op in ("proj", "log10", "tanh", "tan", "asinh", "acosh", "atanh", "asin", "acos", "atan"):
        for tag, key in (("ext", "ext_a"), ("csfdx", "csfd_a"), ("spec", "spec_a")):
            assert ulp_diff(pl.host_complex(Oracle.COP[op], t[key]), t[f"c_{tag}_{op}"]).max() == 0, (op, tag)
    with pytest.raises(ValueError):
        pl.host_complex(99, t["gen_a"])


# the reference's declarations of the launcher API (signatures as in XKinectFusion/include/TsdfVolume.h:16,
# TsdfFusion.h:40-45, RayCaster.h:21-25, ICP.h:24-31, Map.h:16-54), restated over this repository's argument types, and
# a caller of each: what the shim documented in INTEGRATION.md has to satisfy at link time
_REFERENCE_DECLARATIONS = r"""
#pragma once
#include "xs_types.hpp"
void initVolume(PtrStep<short2> volume, PtrStep<float> value_volume, PtrStep<int> weight_volume, PtrStep<float> grad_volume,
                const int3 &volume_resolution);
void integrateTsdfVolume(const PtrStepSz<ushort> &depth, const Intr &intr, int max_weight, const int3 &volume_resolution,
                         float voxel_size, const MatS33 &Rv2c, const devComplex3 &tv2c, const devComplex3 &tc2v, float tranc_dist,
                         PtrStep<float> value_volume, PtrStep<int> weight_volume, PtrStep<float> grad_volume,
                         DeviceArray2D<float> &depthScaled, int frame_id, float threshold = 0.0f, float k = 0.0f);
void raycast(const Intr &intr, const MatS33 &Rc2v, const devComplex3 &tc2v, const MatS33 &Rv2w, const devComplex3 &tv2w,
             float tranc_dist, const int3 &volume_resolution, float voxel_size, const PtrStep<float> &value_volume,
             const PtrStep<float> &grad_volume, MapArr &vmap, MapArr &nmap);
void estimateCombined(const MatS33 &Rcurr, const devComplex3 &tcurr, const MapArr &vmap_curr, const MapArr &nmap_curr,
                      const MatS33 &Rprev_inv, const devComplex3 &tprev, const Intr &intr, const MapArr &vmap_g_prev,
                      const MapArr &nmap_g_prev, float distThres, float angleThres, DeviceArray2D<devComplexICP> &gbuf,
                      DeviceArray<devComplexICP> &mbuf, hostComplexICP *matrixA_host, hostComplexICP *vectorB_host);
void bilateralFilter(const DeviceArray2D<ushort> &src, MapArr &dst);
void pyrDown(const MapArr &src, MapArr &dst);
void createVMap(const Intr &intr, const MapArr &depth, MapArr &vmap);
void createNMap(const MapArr &vmap, MapArr &nmap);
void resizeVMap(const MapArr &input, MapArr &output);
void resizeNMap(const MapArr &input, MapArr &output);
"""
_CALLER = r"""
#include "CudaFunctions.h"
void call_every_launcher() {
    DeviceArray2D<ushort> depth; DeviceArray2D<float> scaled, value, grad; DeviceArray2D<int> weight; DeviceArray2D<short2> vol;
    MapArr a, b, c, d; Intr k(1, 1, 0, 0); MatS33 R{}; devComplex3 t{}; int3 res{64, 64, 64};
    DeviceArray2D<devComplexICP> gbuf; DeviceArray<devComplexICP> mbuf; hostComplexICP A[36], B[6];
    initVolume(vol, value, weight, grad, res);
    integrateTsdfVolume(depth, k, 100, res, 0.1f, R, t, t, 0.3f, value, weight, grad, scaled, 0);
    raycast(k, R, t, R, t, 0.3f, res, 0.1f, value, grad, a, b);
    estimateCombined(R, t, a, b, R, t, k, c, d, 0.1f, 0.2f, gbuf, mbuf, A, B);
    bilateralFilter(depth, a); pyrDown(a, b); createVMap(k, a, b); createNMap(b, c); resizeVMap(a, b); resizeNMap(a, b);
}
"""


def test_integration_md_shim_compiles_and_links(tmp_path):
    """The reference-side binding shown in INTEGRATION.md section 2 is compiled as written against include/xslam_amd.h and linked,
    with --no-undefined, together with a caller of every function the reference declares: a stale argument list in the
    document (round 1: xs_raycast's workspace argument was missing) or a definition whose signature differs from the
    reference's declaration fails here.  No GPU call is made."""
    import subprocess
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```cpp\n(.*?)```", text, flags=re.S)
    shim = [b for b in blocks if "HipLaunchers.cpp" in b]
    assert len(shim) == 1
    (tmp_path / "CudaFunctions.h").write_text(_REFERENCE_DECLARATIONS)
    (tmp_path / "HipLaunchers.cpp").write_text(shim[0])
    (tmp_path / "caller.cpp").write_text(_CALLER)
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = ["g++", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Werror=return-type", "-D__HIP_PLATFORM_AMD__", f"-I{tmp_path}",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "x-slam_amd", "host"), f"-I{rocm}/include",
           str(tmp_path / "HipLaunchers.cpp"), str(tmp_path / "caller.cpp"), "-o", str(tmp_path / "libshim.so"), "-Wl,--no-undefined",
           "-L" + os.path.join(ROOT, "x-slam_amd"), "-lxslam_hip", f"-L{rocm}/lib", "-lamdhip64",
           "-Wl,-rpath," + os.path.join(ROOT, "x-slam_amd")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    # the options-struct form shown in "Additions of round 5" compiles and links as written too (wrapped into a function that declares its names)
    opts = [b for b in blocks if "xs_integrate_scaled_ex2(" in b and "xs_raycast_ex(" in b and "integrate_done" in b]
    assert len(opts) == 1
    (tmp_path / "OptsCaller.cpp").write_text(
        '#include "xslam_amd.h"\n'
        "void opts_caller(const float *scaled, size_t scaled_pitch, int rows, int cols, const float *intr4, int max_weight, const int *res, float voxel_size,\n"
        "                 const float *Rv2c18, const float *tv2c6, float tranc_dist, float *value, int *weight, float *grad, size_t vol_pitch, float threshold,\n"
        "                 unsigned long long *updated_dev, const float *depth_max_dev, void *workspace, void *stream, const void *tiles, void *sign_map,\n"
        "                 void *integrate_done, int shift, float *vmap, float *nmap, size_t map_pitch, float *vmap1, float *nmap1, size_t pitch1, float *vmap2,\n"
        "                 float *nmap2, size_t pitch2, const float *Rc2v18, const float *tc2v6, const float *Rv2w18, const float *tv2w6, unsigned long long *hits_dev,\n"
        "                 float *ray_ws) {\n"
        + opts[0] + "}\n")
    r = subprocess.run(cmd[:cmd.index(str(tmp_path / "HipLaunchers.cpp"))] + [str(tmp_path / "OptsCaller.cpp"), "-o", str(tmp_path / "libopts.so"), "-Wl,--no-undefined",
                       "-L" + os.path.join(ROOT, "x-slam_amd"), "-lxslam_hip", f"-L{rocm}/lib", "-lamdhip64"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    # and the argument count the document passes to every xs_ function equals the header's
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "xslam_amd.h")).read(), flags=re.S)
    nargs = {m.group(1): m.group(2).count(",") + 1 for m in re.finditer(r"\b(xs_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr)}
    assert nargs["xs_raycast"] == 19


def test_flat_yaml_reads_the_reference_config_format():
    """The reference's shipped config (Experiments/test_xkinect_fusion/configs/ICL_traj2.yaml) is flat
    key: value with comments, quoted strings and blank lines; all 34 keys must come through."""
    pl = importlib.import_module("x-slam_amd.pipeline")
    text = """# -------------------------------------Dataset------------------------------------------------
dataset_format: ICL
dataset_dir: "../dataset/ICL/traj2/"
output_dir: "../output/ICL/traj2/"   # trailing comment
start_frame: 0
end_frame: 300
frame_step: 1
factor: 1
is_flip: false

# ------------------------------ Output Setting -------------------------------
log_slam_pose: true
log_gt_pose: true
draw_pcd: true
# note: big threshold may influence the result
biInterpolate_threshold: 0.00
trunc_logistic_k: 0
flag_use_gtPose: false

# tsdf volume
tsdf_size_x: 256
tsdf_size_y: 256
tsdf_size_z: 256
tsdf_voxel_size: 0.03
max_integration_weight: 100
thres_range: 3
init_x: 3.2
init_y: 3.2
init_z: 3.2
r_x: 0
r_y: 0
r_z: 0
depth_width: 640
depth_height: 480
fx: 481.20
fy: -480.00
cx: 319.50
cy: 239.50

# ICP
num_levels: 3
distThres: 0.10
angleThres: 15
"""
    want = {"dataset_format": "ICL", "dataset_dir": "../dataset/ICL/traj2/", "output_dir": "../output/ICL/traj2/", "start_frame": "0",
            "end_frame": "300", "frame_step": "1", "factor": "1", "is_flip": "false", "log_slam_pose": "true", "log_gt_pose": "true",
            "draw_pcd": "true", "biInterpolate_threshold": "0.00", "trunc_logistic_k": "0", "flag_use_gtPose": "false",
            "tsdf_size_x": "256", "tsdf_size_y": "256", "tsdf_size_z": "256", "tsdf_voxel_size": "0.03", "max_integration_weight": "100",
            "thres_range": "3", "init_x": "3.2", "init_y": "3.2", "init_z": "3.2", "r_x": "0", "r_y": "0", "r_z": "0",
            "depth_width": "640", "depth_height": "480", "fx": "481.20", "fy": "-480.00", "cx": "319.50", "cy": "239.50",
            "num_levels": "3", "distThres": "0.10", "angleThres": "15"}
    assert len(want) == 35
    for k, v in want.items():
        assert pl.flat_yaml_get(text, k) == v, k
    assert pl.flat_yaml_get(text, "csfd_seed_row") is None
    # the text produced from a parameter dict round-trips
    synth = importlib.import_module("x-slam_amd.synth")
    prm = synth.s1_params(512)
    t2 = pl.yaml_text(prm)
    assert float(pl.flat_yaml_get(t2, "tsdf_voxel_size")) == prm["tsdf_voxel_size"]
    assert pl.flat_yaml_get(t2, "flag_use_gtPose") == "false" and int(pl.flat_yaml_get(t2, "csfd_seed_col")) == 3


def test_sign_map_host_side_sizes_and_shifts():
    """Host-only entry points of the sign map (csrc/xs_signmap.h): buffer sizes, and the finest brick shift the march can use — the bricks
    must outgrow a wave's 8 x 8 pixel tile at 5 m (dt + (5.2 + dt) * delta <= 0.9 edges with at most 128 samples), the march at most 320 steps."""
    capi = importlib.import_module("x-slam_amd.capi")
    synth = importlib.import_module("x-slam_amd.synth")
    head = 64 + 320 * 4
    assert capi.signmap_bytes([512, 512, 512], 3) == head + 2 * 64 ** 3
    assert capi.signmap_bytes([90, 70, 83], 3) == head + 2 * ((12 * 9 * 11 + 255) // 256 * 256)      # overhanging last bricks, padded to 256
    assert capi.signmap_bytes([512, 512, 512], 1) == 0 and capi.signmap_bytes([512, 512, 512], 7) == 0 and capi.signmap_bytes([0, 1, 1], 3) == 0
    for n, want in ((128, 2), (256, 2), (512, 3), (1024, 4)):
        prm = synth.s1_params(n)
        assert capi.raycast_signmap_shift(synth.intr_of(prm), prm["tsdf_voxel_size"], synth.tranc_dist(prm)) == want, n
    prm = synth.s1_params(512)
    assert capi.raycast_signmap_shift(synth.intr_of(prm), prm["tsdf_voxel_size"], 0.01) == 0       # 600 march steps: more than the table holds
    assert capi.raycast_signmap_shift(synth.intr_of(prm), 1e-4, 3e-4) == 0                          # 16 000 steps
    wide = synth.intr_of(prm) / 8.0                                                                 # a very wide lens: the tile outgrows the bricks
    assert capi.raycast_signmap_shift(wide, prm["tsdf_voxel_size"], synth.tranc_dist(prm)) in (0, 5, 6)


def test_no_per_thread_setter_is_exported():
    """ABI 2: libxslam_hip.so exports no xs_*_set_* function — every option travels in a struct that is an argument."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "x-slam_amd", "libxslam_hip.so")], capture_output=True, text=True, check=True).stdout
    names = [line.split()[-1] for line in out.splitlines() if " T " in line]
    assert len([n for n in names if n.startswith("xs_")]) > 70
    assert [n for n in names if "_set_" in n and n.startswith("xs_") and n != "xs_set_error"] == []    # (xs_set_error: the error plumbing, no setter)


def test_bench_fails_fast_without_a_gpu_per_rank():
    """bench.py --gpus N on a node with fewer than N GPUs (here: none) exits at once, non-zero, with a line that says why — before any rank is
    started, without initialising a device."""
    import subprocess
    import sys
    import time
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 3, (r.returncode, r.stderr[-400:])
    assert "needs 2 visible GPUs" in r.stderr and r.stdout.strip() == ""
    assert time.perf_counter() - t0 < 120


def test_host_code_runs_clean_under_address_and_ub_sanitizers(tmp_path):
    """The header-only host code that needs no GPU (flat YAML reader on well-formed and hostile text, the fixed-size complex algebra, the 6x6
    solvers, DoubleComplex) compiled with -fsanitize=address,undefined -fno-sanitize-recover and run: tests/cxx/host_selftest.cpp.  (GPU
    sanitizers do not exist on this pool; this is the CPU build the task allows them on.)"""
    import subprocess
    exe = str(tmp_path / "host_selftest")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-ffp-contract=off", "-Wall", "-Werror",
           "-I" + os.path.join(ROOT, "x-slam_amd", "host"), "-I" + os.path.join(ROOT, "x-slam_amd", "csrc"), "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cxx", "host_selftest.cpp"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1"))
    assert r.returncode == 0 and "all checks held" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
