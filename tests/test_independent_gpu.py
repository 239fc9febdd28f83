"""GPU: the HIP kernels, through the C ABI, against the independently written float64 model and its central differences
(tests/independent_f64.py; no text shared with oracle/): kernel values to float32 rounding and — the point of the
product — the imaginary parts / h against finite-difference derivatives of a real-valued model, per kernel: integrate
voxel update, raycast vertex + normal, ICP rows + 27 sums, dual-complex residual (loss, gradient, second derivative).
Measured figures are written to gpurun_out/independent_f64.json when that directory exists."""
import importlib
import json
import os

import numpy as np
import pytest

import independent_cases as ic

pytestmark = pytest.mark.gpu
CASES = [("s3", (2, 3), 0.0), ("s1", (0, 3), 0.05)]
LOG = {}


@pytest.fixture(scope="module")
def be():
    import torch
    assert torch.cuda.is_available()
    capi = importlib.import_module("x-slam_amd.capi")
    yield ic.GpuBackend(torch, capi, _m3_inverse)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out) and LOG:
        json.dump(LOG, open(os.path.join(out, "independent_f64.json"), "w"), indent=1)


def _m3_inverse(R):
    """complex 3x3 inverse of an input transform (a test input, computed in float64 and rounded once)."""
    R = np.asarray(R, np.float64).reshape(3, 3, 2)
    inv = np.linalg.inv(R[..., 0] + 1j * R[..., 1])
    out = np.zeros((3, 3, 2), np.float32)
    out[..., 0], out[..., 1] = inv.real, inv.imag
    return out


@pytest.mark.parametrize("n", [128, 256])
@pytest.mark.parametrize("scene,seed,threshold", CASES)
def test_integrate_values_and_derivatives(be, scene, seed, threshold, n):
    r = ic.check_integrate(be, n=n, scene=scene, seed=seed, threshold=threshold)
    LOG[f"integrate_{scene}_{n}"] = r
    assert r["n_band"] > 5000 and r["deriv_scale"] > 1.0
    assert r["written_disagree"] <= 1e-3
    assert r["value_bad"] <= 1e-3 and r["value_err_p999"] <= 2e-5
    assert r["deriv_bad"] <= 1e-3 and r["deriv_err_p999_rel"] <= 1e-4
    assert (r["bilinear_share"] > 0.9) == (threshold > 0)


@pytest.mark.parametrize("n", [128, 256])
@pytest.mark.parametrize("scene,seed,threshold", CASES)
def test_raycast_values_and_derivatives(be, scene, seed, threshold, n):
    r = ic.check_raycast(be, n=n, scene=scene, seed=seed, threshold=threshold)
    LOG[f"raycast_{scene}_{n}"] = r
    assert r["n_hit"] > 10000 and r["dvertex_scale"] > 1.0
    assert r["hit_disagree"] <= 2e-3 and r["normal_disagree"] <= 2e-3
    assert r["vertex_bad"] <= 2e-3 and r["vertex_err_p99"] <= 1e-5
    assert r["normal_bad"] <= 5e-3 and r["normal_err_p99"] <= 2e-4
    assert r["dvertex_bad"] <= 5e-3 and r["dvertex_err_p99_rel"] <= 2e-4
    assert r["dnormal_bad"] <= 5e-3 and r["dnormal_err_p99_rel"] <= 2e-4


@pytest.mark.parametrize("level", [0, 1, 2])
@pytest.mark.parametrize("scene,seed,threshold", CASES)
def test_icp_sums_and_derivatives(be, scene, seed, threshold, level):
    r = ic.check_icp(be, n=128, scene=scene, seed=seed, threshold=threshold, level=level)
    LOG[f"icp_{scene}_level{level}"] = r
    assert r["inliers"] > 0.5 * (480 >> level) * (640 >> level)
    assert abs(r["inliers"] - r["inliers_model"]) <= max(3, 5e-5 * r["inliers"])
    assert r["value_err_rel"] <= 2e-5 and r["deriv_err_rel"] <= 1e-4 and r["deriv_scale"] > 0


@pytest.mark.parametrize("scene,n", [("s3", 128), ("s1", 128), ("s3", 256)])
def test_hessian_loss_gradient_and_second_derivative(be, scene, n):
    r = ic.check_hessian(be, n=n, scene=scene)
    LOG[f"hessian_{scene}_{n}"] = r
    assert r["count"] > 1000 and abs(r["count"] - r["count_model"]) <= max(2, 1e-4 * r["count_model"])
    assert abs(r["loss"] - r["loss_model"]) <= 5e-4 * abs(r["loss_model"])
    # the gradient is a sum of terms of both signs (near a minimum they cancel: S1 at 128^3 has |grad| = 122 where the terms'
    # scale sqrt(loss * hessian) is 6 100): tolerance relative to the larger of the two
    assert abs(r["grad"] - r["grad_model"]) <= 5e-4 * max(abs(r["grad_model"]), (r["loss_model"] * abs(r["hess_model"])) ** 0.5)
    assert abs(r["hess"] - r["hess_model"]) <= 5e-4 * abs(r["hess_model"])


@pytest.mark.parametrize("n", [256, 512])
def test_survey_reference_kernel_figures_through_the_kernels(be, n):
    """The counts SURVEY.md section 6 recorded from the reference's own kernel bodies on scene S1 frame 0, through the HIP
    kernels alone (C ABI, no orchestrator): voxels written by integrate, rays hit by raycast, ICP inliers at level 0
    (tests/golden/survey_reference_kernel_figures.json; tolerances stated there)."""
    from helpers import intr_of, s1_transforms, synth
    fig = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "survey_reference_kernel_figures.json")))
    prm = synth.s1_params(n)
    T = s1_transforms(0, prm)
    d0 = synth.s1_frame(0)
    zero = (np.zeros(n ** 3, np.float32), np.zeros(n ** 3, np.int32), np.zeros(n ** 3, np.float32))
    state = be.integrate(zero, be.scale_depth(d0), prm, T, 0.0)
    U = int((state[1] != 0).sum())
    pv, pn, hits = be.raycast(state, prm, T)
    cv, cn = be.current_maps(d0, prm, 0)
    angle = float(np.sin(np.float32(15.0) / np.float32(180.0) * np.pi))
    _, inl = be.icp(T["Rc2w"], T["tc2w"], cv, cn, be.m3_inverse(T["Rc2w"]), T["tc2w"], intr_of(prm), pv, pn, 0.10, angle)
    LOG[f"survey_figures_{n}"] = dict(U=U, hits=hits, inliers=inl)
    assert abs(U - fig["integrate_U"][str(n)]) <= max(2, 1e-4 * fig["integrate_U"][str(n)])
    assert abs(hits - fig["raycast_hits"][str(n)]) <= 2
    assert abs(inl - fig["icp_inliers_level0"][str(n)]) <= 0.003 * fig["icp_inliers_level0"][str(n)]
