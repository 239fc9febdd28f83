"""Pins the oracle's scalar arithmetic (oracle/oc_complex.hpp, oc_csfd.hpp):
  * against tables produced by the reference's own cuda_complex.hpp (oracle/_ref), committed
    as tests/golden/scalar_tables.npz;
  * against the known answers the reference's test_CSFD demo prints.
CPU only."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_golden, ulp_diff

BINARY = ("add", "sub", "mul", "div", "div_scalar", "scalar_div", "mul_scalar", "scalar_sub")
UNARY = ("sqrt", "abs", "exp", "sin", "cos", "sinh", "cosh", "sin_new", "sinh_new", "norm", "arg", "conj")


@pytest.fixture(scope="module")
def tables():
    return load_golden("scalar_tables.npz")


@pytest.mark.parametrize("tag", ["csfd", "gen", "wide"])
@pytest.mark.parametrize("op", BINARY)
def test_complex_binary_bit_exact(oracle, tables, tag, op):
    got = oracle.cop(op, tables[f"{tag}_a"], tables[f"{tag}_b"])
    # same formulas, same libm, no FMA contraction on either side: bit-exact
    assert ulp_diff(got, tables[f"c_{tag}_{op}"]).max() == 0


@pytest.mark.parametrize("tag", ["csfd", "gen", "pos"])
@pytest.mark.parametrize("op", UNARY)
def test_complex_unary_bit_exact(oracle, tables, tag, op):
    got = oracle.cop(op, tables[f"{tag}_a"])
    assert ulp_diff(got, tables[f"c_{tag}_{op}"]).max() == 0


def test_complex_log_pow_polar(oracle, tables):
    assert ulp_diff(oracle.cop("log", tables["pos_a"]), tables["c_pos_log"]).max() == 0
    assert ulp_diff(oracle.cop("log", tables["gen_a"]), tables["c_gen_log"]).max() == 0
    assert ulp_diff(oracle.cop("pow", tables["pos_a"], tables["csfd_b"]), tables["c_pos_pow"]).max() == 0
    assert ulp_diff(oracle.cop("polar", np.abs(tables["gen_a"]), tables["gen_b"]), tables["c_gen_polar"]).max() == 0


EXT = ("proj", "log10", "tanh", "tan", "asinh", "acosh", "atanh", "asin", "acos", "atan")


@pytest.mark.parametrize("tag,key", [("ext", "ext_a"), ("csfdx", "csfd_a"), ("spec", "spec_a")])
@pytest.mark.parametrize("op", EXT)
def test_complex_remaining_functions_bit_exact(oracle, tables, tag, key, op):
    """cuda_complex.hpp:506-516, 570-577, 640-723, 770-841, 873-881 — general operands, CSFD-regime operands and the
    cross product of {0, -0, 1, -1, 0.5, inf, -inf, nan} (every special-value branch), NaN == NaN."""
    got = oracle.cop(op, tables[key])
    assert ulp_diff(got, tables[f"c_{tag}_{op}"]).max() == 0


@pytest.mark.parametrize("op", ["add", "sub", "mul", "div", "sqrt"])
def test_complex_double_bit_exact(oracle, tables, op):
    got = oracle.cop_f64(op, tables["f64_a"], tables["f64_b"])
    assert np.array_equal(got, tables[f"c64_{op}"])


@pytest.mark.parametrize("op", ["add", "sub", "mul", "div", "mul_scalar", "div_scalar", "add_scalar", "scalar_sub"])
def test_dual_complex_binary(oracle, tables, op):
    got = oracle.dop(op, tables["d_a"], tables["d_b"])
    assert ulp_diff(got, tables[f"d_{op}"]).max() == 0


def test_dual_complex_sqrt_abs(oracle, tables):
    assert ulp_diff(oracle.dop("sqrt", tables["d_pos"]), tables["d_sqrt"]).max() == 0
    assert ulp_diff(oracle.dop("abs", tables["d_pos"]), tables["d_abs"]).max() == 0


def test_live_against_reference_header(oracle, oracle_ref):
    """Where oracle/_ref exists (build container), compare live on fresh operands."""
    rng = np.random.default_rng(7)
    a = np.stack([rng.uniform(0.01, 5, 4096), rng.uniform(-1e-6, 1e-6, 4096)], -1).astype(np.float32)
    b = np.stack([rng.uniform(-5, 5, 4096), rng.uniform(-1e-6, 1e-6, 4096)], -1).astype(np.float32)
    for op in BINARY:
        assert ulp_diff(oracle.cop(op, a, b), oracle_ref.cop(op, a, b)).max() == 0, op
    for op in UNARY + ("log",):
        assert ulp_diff(oracle.cop(op, a), oracle_ref.cop(op, a)).max() == 0, op


# ---- test_CSFD known answers (6 significant digits as printed) ---------------
def _close6(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return np.all(np.abs(got - want) <= 5.1e-6 * np.maximum(np.abs(want), 1e-300) + 1e-12 * (want == 0))


def test_csfd_known_answers(oracle):
    ka = json.load(open(os.path.join(GOLDEN, "test_csfd_known_answers.json")))
    a = np.array([ka["inputs"]["a"]], np.float32)
    b = np.array([ka["inputs"]["b"]], np.float32)
    for name in ("mul", "div", "exp", "sin", "pow"):
        ours = oracle.csfd_op(name, "our", a, b)[0]
        assert _close6(ours, ka[f"{name}_our"]), (name, ours)
    # the "std" column is std::complex arithmetic; raw forms agree with it to print precision
    for name in ("mul", "div", "exp", "sin"):
        raw = oracle.csfd_op(name, "raw", a, b)[0]
        assert _close6(raw, ka[f"{name}_std"]), (name, raw)
    # the same values through the restated complex<T> operators (cuda_complex.hpp)
    assert _close6(oracle.cop("mul", a, b)[0], ka["mul_std"])
    assert _close6(oracle.cop("div", a, b)[0], ka["div_std"])
    s = oracle.cop("add", a, b)
    assert _close6(oracle.cop("exp", s)[0], ka["exp_std"])
    assert _close6(oracle.cop("sin", s)[0], ka["sin_std"])
    three = np.array([[3.0, 0.0]], np.float32)
    # the demo's "std" pow is libstdc++'s integer-power product (-1, 6e-06); exp(y log x) of
    # cuda_complex.hpp:625-640 carries float(pi)'s rounding into the imaginary part instead
    got = oracle.cop("pow", s, three)[0]
    assert abs(got[0] + 1.0) < 1e-6 and abs(got[1] - 6e-06) < 4e-7


def test_dcsfd_chain_rule_known_answers(oracle):
    ka = json.load(open(os.path.join(GOLDEN, "test_csfd_known_answers.json")))
    out = oracle.csfd_chain_rule(ka["inputs"]["t0"], ka["inputs"]["h"])
    assert _close6(out[0], ka["dcsfd_gradient"])
    assert _close6(out[1], ka["dcsfd_second"])
    assert _close6(out[2], ka["chain_gradient"])
    assert _close6(out[3], ka["chain_second"])
    # analytic: f = (t^2 + sin t)^2
    t = 0.5
    u, du, ddu = t * t + np.sin(t), 2 * t + np.cos(t), 2 - np.sin(t)
    assert abs(out[0] - 2 * u * du) < 1e-4
    assert abs(out[1] - (2 * du * du + 2 * u * ddu)) < 1e-2


def test_host_double_complex_matches_device_dual_complex(oracle, tables):
    """DoubleComplex (std::complex) and d_complex restate the same algebra (+ - * /)."""
    for op in ("add", "sub", "mul", "div"):
        h = oracle.hdop(op, tables["d_a"], tables["d_b"])
        d = oracle.dop(op, tables["d_a"], tables["d_b"])
        if op == "div":
            # std::complex '/' (libgcc __divsc3) and the libc++-style quotient round differently
            # by an ulp; the second-order slot is a difference of such terms
            assert np.allclose(h[:, :3], d[:, :3], rtol=2e-5, atol=1e-12)
            assert np.allclose(h[:, 3], d[:, 3], rtol=2e-4, atol=1e-18)
        else:
            assert ulp_diff(h, d).max() <= 2, op
