"""GPU: whole-trajectory parity at BASELINE's sizes.  The HIP pipeline (C++ orchestrator + kernels through the C ABI)
and the CPU oracle pipeline process the same 30 frames of scene S1 (SURVEY's tracking scene) and S3 (the box room, every
degree of freedom constrained) at 256^3 and 512^3, side by side; every frame's pose, pose derivative, voxels written,
rays hit and ICP inlier counts are compared, then the fused volumes on 400 000 seeded voxels.

Stated envelope (measured figures: DESIGN.md "Parity status" and docs/DESIGN_rounds1-3.md; the per-frame log goes to gpurun_out/trajectory_parity.json).
Scene S3 constrains all six degrees of freedom.  There the two implementations produce IDENTICAL BITS for the first 15
frames at both sizes (every pose entry, real and imaginary), then differ in the last digits once a bilateral-filter pixel
(expf ulp) or a flipped discrete decision enters:
  S3   pose entries |d| <= 5e-6 (measured <= 1.2e-6), pose derivative within 2e-3 of its largest entry (measured <= 4e-4),
       voxels written / rays hit within 10 of the oracle's counts (measured <= 5), ICP inliers within 20, every frame;
       fused volume after 30 frames: weights equal on all sampled voxels but FLIPS, value within 1e-3 there.
Scene S1 (SURVEY's wall + sphere; the bench's scene) leaves sliding along the wall to the sphere alone: its 6x6 system is
nearly singular, and a last-digit difference entering at frame 3-5 is amplified ~100x per frame until it saturates at the
size of the scene's free motion (millimetres; DESIGN.md section 6 — the oracle run against itself with one depth pixel
changed by 1 mm departs just as far: measured here as "sensitivity").  So for S1:
  frames 0-3    pose |d| <= 1e-6, derivative within 1e-4 — identical inputs, identical bits expected and measured
  frames 4-29   both sides keep tracking; pose |d| <= 3e-2 (measured <= 1.8e-2) and no larger than 10x what the one-pixel
                perturbation of the GPU pipeline itself produces; voxels written / rays hit within 0.2 % of the oracle's."""
import importlib
import json
import os

import numpy as np
import pytest

from trajectory_cases import side_by_side

pytestmark = pytest.mark.gpu
FRAMES = 30
FLIPS = 2e-3
LOG = {}


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available()
    yield torch, importlib.import_module("x-slam_amd.pipeline")
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out) and LOG:
        json.dump(LOG, open(os.path.join(out, "trajectory_parity.json"), "w"), indent=1)


@pytest.mark.parametrize("n", [256, 512])
def test_thirty_frames_constrained_scene_side_by_side_with_the_oracle(dev, oracle, n):
    torch, pl = dev
    r = side_by_side(torch, pl, oracle, "s3", n, FRAMES, seed=(2, 3), threshold=0.0)
    LOG[f"s3_{n}"] = r
    dpose, dder = np.array(r["dpose"]), np.array(r["dderiv_rel"])
    assert dpose[:10].max() == 0.0 and dder[:10].max() <= 1e-6, (dpose[:10], dder[:10])   # identical bits while no decision has flipped
    assert dpose.max() <= 5e-6, dpose.tolist()
    assert dder.max() <= 2e-3, dder.tolist()
    assert min(r["deriv_scale"]) >= 0.9                                         # d pose(2,3) / d seed stays ~1: the derivative is alive
    assert max(r["dU"]) <= 10 and max(r["dhits"]) <= 10 and 0 <= min(r["dinliers"]) and max(r["dinliers"]) <= 20, (r["dU"], r["dhits"], r["dinliers"])
    v = r["voxels"]
    assert v["touched"] > 1000 and v["weight_mismatch"] <= 1e-4 and v["value_max"] <= 1e-3 and v["grad_bad"] <= 1e-4, v


@pytest.mark.parametrize("n", [256, 512])
def test_thirty_frames_bench_scene_side_by_side_with_the_oracle(dev, oracle, n):
    torch, pl = dev
    r = side_by_side(torch, pl, oracle, "s1", n, FRAMES, seed=(0, 3), threshold=0.0, sensitivity=True)
    LOG[f"s1_{n}"] = r
    dpose, dder = np.array(r["dpose"]), np.array(r["dderiv_rel"])
    assert dpose[:4].max() <= 1e-6 and dder[:4].max() <= 1e-4, (dpose[:4], dder[:4])
    assert dpose.max() <= 3e-2, dpose.tolist()
    sens = np.array(r["sensitivity_dpose"])
    assert dpose.max() <= 10 * max(sens.max(), 1e-3), (dpose.max(), sens.max())  # no worse than the scene's own conditioning
    assert np.all(np.array(r["dU"]) <= np.maximum(3, 2e-3 * np.array(r["U"]))), r["dU"]
    assert np.all(np.array(r["dhits"]) <= np.maximum(3, 2e-3 * np.array(r["hits"]))), r["dhits"]
    assert min(r["dinliers"]) >= 0 and max(r["dinliers"]) <= 0.01 * 640 * 480
    # the pose derivative, every frame (round 3): in units of d pose / d seed.  On this scene the lateral seed dies within a few frames
    # (deriv_scale falls to ~1e-3: nearest-pixel depth), so a relative figure means nothing after frame 3; what is asserted
    # is that the two pipelines' derivatives differ by no more than 10x what the one-pixel perturbation does to the GPU pipeline's own
    # derivative (the scene's conditioning), with a floor of 1e-4 of the seed's initial unit derivative
    dabs, sens_d = np.array(r["dderiv_abs"]), np.array(r["sensitivity_dderiv_abs"])
    assert dabs[:4].max() <= 1e-4, dabs[:4]
    assert dabs[4:].max() <= 10 * max(sens_d.max(), 1e-4), (dabs[4:].max(), sens_d.max())
    v = r["voxels"]
    assert v["touched"] > 1000 and v["weight_mismatch"] <= 5e-3, v


def test_thirty_frames_bilinear_branch_lateral_seed_512(dev, oracle):
    """The branch on which a lateral seed survives: scene S3 at 512^3 with biInterpolate_threshold 0.05 (bilinear depth lookup wherever the
    four neighbouring depths agree to 5 cm: TsdfFusion.cu:135-140) and the bench's seed, world2camera(0,3).  Thirty frames side by side
    with the oracle, S3's envelope: pose entries within 5e-6, pose derivative within 2e-3 of its largest entry — and that derivative is
    alive (d pose(0,3) / d seed stays of order 1), unlike with nearest-pixel depth."""
    torch, pl = dev
    r = side_by_side(torch, pl, oracle, "s3", 512, FRAMES, seed=(0, 3), threshold=0.05)
    LOG["s3_512_bilinear_seed03"] = r
    dpose, dder = np.array(r["dpose"]), np.array(r["dderiv_rel"])
    assert dpose.max() <= 5e-6, dpose.tolist()
    assert dder.max() <= 2e-3, dder.tolist()
    assert min(r["deriv_scale"]) >= 0.5 and max(r["deriv_scale"]) <= 5.0, r["deriv_scale"]   # alive, and not running away
    assert max(r["dU"]) <= 10 and max(r["dhits"]) <= 10 and 0 <= min(r["dinliers"]) and max(r["dinliers"]) <= 20, (r["dU"], r["dhits"], r["dinliers"])
    v = r["voxels"]
    assert v["touched"] > 1000 and v["weight_mismatch"] <= 1e-4 and v["value_max"] <= 1e-3 and v["grad_bad"] <= 1e-4, v
