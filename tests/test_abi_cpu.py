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


@pytest.mark.parametrize("header,lib", [("xslam_amd.h", "libxslam_hip.so"), ("xslam_amd_pipeline.h", "libxslam_host.so")])
def test_every_declared_symbol_is_exported(header, lib):
    path = os.path.join(ROOT, "x-slam_amd", lib)
    assert os.path.exists(path), f"{lib} not built: run __graft_entry__.build()"
    if lib == "libxslam_host.so":
        ctypes.CDLL(os.path.join(ROOT, "x-slam_amd", "libxslam_hip.so"), mode=ctypes.RTLD_GLOBAL)
    h = ctypes.CDLL(path)
    names = declared(header)
    assert len(names) >= 20 if header == "xslam_amd.h" else len(names) >= 20
    for n in names:
        assert hasattr(h, n), f"{n} declared in include/{header} but not exported by {lib}"


def test_ctypes_tables_cover_the_headers():
    capi = importlib.import_module("x-slam_amd.capi")
    pl = importlib.import_module("x-slam_amd.pipeline")
    assert set(declared("xslam_amd.h")) == set(capi._SIGS)
    assert set(declared("xslam_amd_pipeline.h")) == set(pl._SIGS)
    assert capi.abi_version() == 1
    assert capi.icp_workspace_bytes() > 0 and capi.tsdf_reduce_workspace_bytes() > 0
    assert capi.integrate_workspace_bytes([512, 512, 512]) == 256 + 8 * 128 * 256 * 4
    # host-only entry point: unpacking the 27 sums into the symmetric system (ICP.cu:419-428)
    import numpy as np
    s = np.arange(54, dtype=np.float64)
    A, b = capi.icp_unpack(s)
    A = A.reshape(6, 6, 2)
    assert np.array_equal(A, A.transpose(1, 0, 2))
    assert A[0, 0, 0] == 0 and A[0, 1, 0] == 2 and b[0] == 12 and b[1] == 13 and A[5, 5, 0] == 50 and b[10] == 52


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
