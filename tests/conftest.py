import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU restatement (oracle/liboracle.so), built on first use.  Test infrastructure only."""
    from oracle import oracle as orc
    orc.build(ref=False)
    return orc.Oracle()


@pytest.fixture(scope="session")
def oracle_ref():
    """oracle/_ref (the reference's own complex class); only where it was built."""
    from oracle import oracle as orc
    if os.path.isdir("/root/reference/DeviceArray/include"):
        orc.build(ref=True)
    try:
        return orc.Oracle(ref=True)
    except (FileNotFoundError, OSError):
        pytest.skip("oracle/_ref not built (needs /root/reference)")


@pytest.fixture(scope="session")
def synth():
    return importlib.import_module("x-slam_amd.synth")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


def ulp_diff(a, b):
    """Distance in float32 units-in-the-last-place, NaN == NaN."""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    d = np.abs(ia - ib)
    both_nan = np.isnan(a) & np.isnan(b)
    return np.where(both_nan, 0, d)
