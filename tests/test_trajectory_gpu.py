"""GPU: whole-trajectory parity at BASELINE's sizes.  The HIP pipeline (C++ orchestrator + kernels through the C ABI)
and the CPU oracle pipeline process the same 30 frames of scene S1 (SURVEY's tracking scene) and S3 (the box room, every
degree of freedom constrained) at 256^3 and 512^3, side by side; every frame's pose, pose derivative, voxels written,
rays hit and ICP inlier counts are compared, then the fused volumes on 400 000 seeded voxels.

Stated envelope (measured figures: DESIGN.md "Parity status"; the per-frame log is written to gpurun_out/trajectory_parity.json):
the two sides differ by a couple of 1 mm pixels after the bilateral filter (expf ulp) and by flipped discrete decisions
(pixel picks, zero crossings, ICP gates), so agreement cannot be bit-exact after frame 0; it must stay inside
  pose entries          |d| <= POSE_TOL[scene]           (metres / rotation-matrix entries), every frame
  pose derivative       |d Im| <= DERIV_REL * max|Im|     every frame
  voxels written, hits  within COUNT_REL of the oracle's
  fused volume          weights equal on all but FLIPS of the sampled voxels; value / grad within 1e-4 / 1e-3 of scale there."""
import importlib
import json
import os

import numpy as np
import pytest

from trajectory_cases import side_by_side

pytestmark = pytest.mark.gpu
FRAMES = 30
POSE_TOL = {"s1": 2e-4, "s3": 2e-5}
DERIV_REL = {"s1": 5e-2, "s3": 5e-3}
COUNT_REL = 2e-3
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
@pytest.mark.parametrize("scene,seed,threshold", [("s1", (0, 3), 0.0), ("s3", (2, 3), 0.0)])
def test_thirty_frames_side_by_side_with_the_oracle(dev, oracle, scene, seed, threshold, n):
    torch, pl = dev
    r = side_by_side(torch, pl, oracle, scene, n, FRAMES, seed=seed, threshold=threshold)
    LOG[f"{scene}_{n}"] = r
    dpose, dder = np.array(r["dpose"]), np.array(r["dderiv_rel"])
    assert dpose[0] == 0.0 and dder[0] == 0.0                                  # frame 0: the given pose
    assert dpose[1] <= 1e-6 and dder[1] <= 1e-5, (dpose[1], dder[1])           # first tracked frame
    assert dpose.max() <= POSE_TOL[scene], dpose.tolist()
    assert dder.max() <= DERIV_REL[scene], dder.tolist()
    assert np.all(np.array(r["dU"]) <= np.maximum(3, COUNT_REL * np.array(r["U"]))), r["dU"]
    assert np.all(np.array(r["dhits"]) <= np.maximum(3, COUNT_REL * np.array(r["hits"]))), r["dhits"]
    assert min(r["dinliers"]) >= 0 and max(r["dinliers"]) <= 0.002 * 640 * 480
    v = r["voxels"]
    assert v["touched"] > 1000
    assert v["weight_mismatch"] <= FLIPS and v["value_bad"] <= FLIPS and v["grad_bad"] <= FLIPS, v
