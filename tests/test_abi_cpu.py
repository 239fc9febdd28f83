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
