"""CPU: the oracle (oracle/, complex<float> restatement) against the independently written float64 model and its
central differences (tests/independent_f64.py) — a pin of the oracle's kernels that does not come from the oracle's own
text: values to float32 rounding, CSFD derivatives (imaginary parts / h) against finite differences of a real-valued
model.  The same cases run against the HIP kernels in tests/test_independent_gpu.py."""
import pytest

import independent_cases as ic

# (scene, seeded entry of world2camera, biInterpolate_threshold): the axial seed with nearest-pixel depth and the lateral
# seed with the bilinear depth lookup (TsdfFusion.cu:133-141), the two ways a derivative enters the volume
CASES = [("s3", (2, 3), 0.0), ("s1", (0, 3), 0.05)]


@pytest.fixture(scope="module")
def be(oracle):
    return ic.OracleBackend(oracle)


@pytest.mark.parametrize("scene,seed,threshold", CASES)
def test_integrate_values_and_derivatives(be, scene, seed, threshold):
    r = ic.check_integrate(be, n=64, scene=scene, seed=seed, threshold=threshold)
    assert r["n_band"] > 1500 and r["deriv_scale"] > 1.0
    assert r["written_disagree"] <= 1e-3                      # same voxels written
    assert r["value_bad"] <= 1e-3 and r["value_err_p999"] <= 1e-5
    assert r["deriv_bad"] <= 1e-3 and r["deriv_err_p999_rel"] <= 5e-5
    assert (r["bilinear_share"] > 0.9) == (threshold > 0)


@pytest.mark.parametrize("scene,seed,threshold", CASES)
def test_raycast_values_and_derivatives(be, scene, seed, threshold):
    r = ic.check_raycast(be, n=64, scene=scene, seed=seed, threshold=threshold)
    assert r["n_hit"] > 10000 and r["dvertex_scale"] > 1.0
    assert r["hit_disagree"] <= 2e-3 and r["normal_disagree"] <= 2e-3
    assert r["vertex_bad"] <= 2e-3 and r["vertex_err_p99"] <= 1e-5
    assert r["normal_bad"] <= 2e-3 and r["normal_err_p99"] <= 1e-4
    assert r["dvertex_bad"] <= 2e-3 and r["dvertex_err_p99_rel"] <= 1e-4
    assert r["dnormal_bad"] <= 2e-3 and r["dnormal_err_p99_rel"] <= 1e-4


@pytest.mark.parametrize("level", [0, 1, 2])
@pytest.mark.parametrize("scene,seed,threshold", CASES)
def test_icp_sums_and_derivatives(be, scene, seed, threshold, level):
    r = ic.check_icp(be, n=64, scene=scene, seed=seed, threshold=threshold, level=level)
    assert r["inliers"] > 0.5 * (480 >> level) * (640 >> level)
    assert abs(r["inliers"] - r["inliers_model"]) <= max(3, 5e-5 * r["inliers"])
    assert r["value_err_rel"] <= 2e-5 and r["deriv_err_rel"] <= 1e-4 and r["deriv_scale"] > 0


@pytest.mark.parametrize("scene", ["s3", "s1"])
def test_hessian_loss_gradient_and_second_derivative(be, scene):
    r = ic.check_hessian(be, n=64, scene=scene)
    assert r["count"] > 1000 and abs(r["count"] - r["count_model"]) <= 2
    assert abs(r["loss"] - r["loss_model"]) <= 2e-4 * abs(r["loss_model"])
    assert abs(r["grad"] - r["grad_model"]) <= 5e-4 * max(abs(r["grad_model"]), (r["loss_model"] * abs(r["hess_model"])) ** 0.5)
    assert abs(r["hess"] - r["hess_model"]) <= 5e-4 * abs(r["hess_model"])


def test_hessian_on_a_slab_of_planes(be):
    """The slab form (gt holds planes [z0, z1) only, as xs_compute_local_tsdf_hessian takes a z-shard) against the model."""
    _, states = ic.two_frames(be, 64, "s3", (0, 3), 0.0)
    r = ic.check_hessian(be, n=64, scene="s3", z0=20, z1=40, gt=states[1][0])
    assert r["count"] > 50 and r["count"] == r["count_model"]
    assert abs(r["loss"] - r["loss_model"]) <= 5e-4 * abs(r["loss_model"])
    assert abs(r["hess"] - r["hess_model"]) <= 5e-4 * abs(r["hess_model"])
