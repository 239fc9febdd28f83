"""GPU: the record hand-offs under repetition.  k_icp and the TSDF residual kernels publish one record per workgroup with relaxed
agent-scope (write-through) stores + s_waitcnt vmcnt(0) + a relaxed ticket, and the last workgroup adds the records after an acquire
(csrc/xs_icp.hip, csrc/xs_tsdf.hip: block_fold_and_finish) — a protocol that leans on what gfx942 / gfx950 do with such stores, outside
the HIP memory model.  A stale record would show as a sum that differs from launch to launch: thousands of back-to-back launches on
fixed inputs must give identical bits.  And the posted-pose loop (a resident launch polling a mailbox the host writes) must neither time
out nor change a pose when the host is slow."""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import intr_of, s1_transforms, synth, tranc_dist

pytestmark = pytest.mark.gpu
H, W = synth.HEIGHT, synth.WIDTH
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    return torch, importlib.import_module("x-slam_amd.capi"), importlib.import_module("x-slam_amd.pipeline")


def icp_stress(launches):
    """`launches` level-0 reductions on fixed inputs; returns the number of distinct results (1 = every launch gave the same bits)."""
    import torch
    capi = importlib.import_module("x-slam_amd.capi")
    from oracle.oracle import Oracle
    o = Oracle()
    n = 96
    prm = synth.s1_params(n)
    res = [n, n, n]
    v, w, g = o.new_volume(res)
    T0 = s1_transforms(0, prm)
    o.integrate(o.scale_depth(synth.s1_frame(0)), v, w, g, res, tranc_dist(prm), 100, T0["Rv2c"], T0["tv2c"], intr_of(prm), prm["tsdf_voxel_size"])
    pv, pn, _ = o.raycast(intr_of(prm), T0["Rc2v"], T0["tc2v"], T0["Rv2w"], T0["tv2w"], tranc_dist(prm), res, prm["tsdf_voxel_size"], v, g, H, W)
    cv = o.create_vmap(intr_of(prm), o.bilateral(synth.s1_frame(1)))
    cn = o.create_nmap(cv)
    dv = [torch.from_numpy(np.ascontiguousarray(x)).cuda() for x in (cv, cn, pv, pn)]
    Rprev_inv = o.m3_inverse(T0["Rc2w"])
    angle = float(np.sin(np.float32(15.0) / np.float32(180.0) * np.pi))
    ws = torch.zeros(capi.icp_workspace_bytes(), dtype=torch.uint8, device="cuda")
    out = torch.zeros((launches, 56), dtype=torch.float64, device="cuda")
    k = intr_of(prm)
    for i in range(launches):
        capi.icp_accumulate(T0["Rc2w"], T0["tc2w"], dv[0], dv[1], Rprev_inv, T0["tc2w"], k, dv[2], dv[3], W * 8, H, W, 0.10, angle, ws, out[i])
    torch.cuda.synchronize()
    r = out[:, :55].cpu().numpy().view(np.int64)
    assert r[0, 54] != 0 and float(out[0, 54]) > 100000          # level 0 of a 640 x 480 frame: the 16-wave (or, XS_ICP_BALANCED=1, 8-wave) balanced instance
    return int(len(np.unique(r, axis=0)))


def test_icp_reduction_5000_launches_sixteen_wave_instance(dev):
    assert icp_stress(5000) == 1


def test_icp_reduction_5000_launches_eight_wave_instance():
    """XS_ICP_BALANCED=1 selects the 512 x 8-wave balanced instance (read once per process: a child)."""
    env = dict(os.environ, XS_ICP_BALANCED="1")
    code = "import sys; sys.path.insert(0, 'tests'); import test_publish_stress_gpu as t; n = t.icp_stress(5000); print('distinct', n); sys.exit(0 if n == 1 else 1)"
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_icp_pairs_publication_read_as_it_arrives_4000_launches(dev):
    """XS_ICP_PUBLISH_PAIRS under repetition: every launch's 55 sums are read by the host the moment all 55 pairs carry the launch's number —
    no stream synchronisation, no completion word — at levels 0 and 2 alternating (sixteen- and eight-wave instances, 256 and 45 workgroups).
    A pair seen in halves, or a number that overtook its sum, would show as a result that differs from the first of its level."""
    import ctypes as C
    torch, capi, pl = dev
    from oracle.oracle import Oracle
    o = Oracle()
    n = 96
    prm = synth.s1_params(n)
    res = [n, n, n]
    v, w, g = o.new_volume(res)
    T0 = s1_transforms(0, prm)
    o.integrate(o.scale_depth(synth.s1_frame(0)), v, w, g, res, tranc_dist(prm), 100, T0["Rv2c"], T0["tv2c"], intr_of(prm), prm["tsdf_voxel_size"])
    pv, pn, _ = o.raycast(intr_of(prm), T0["Rc2v"], T0["tc2v"], T0["Rv2w"], T0["tv2w"], tranc_dist(prm), res, prm["tsdf_voxel_size"], v, g, H, W)
    d = o.bilateral(synth.s1_frame(1))
    levels = []
    for level in range(3):
        if level:
            pv, pn, d = o.resize_map(pv, False), o.resize_map(pn, True), o.pyr_down(d)
        k = intr_of(prm, level)
        cv = o.create_vmap(k, d); cn = o.create_nmap(cv)
        levels.append((k, cv.shape[0] // 3, cv.shape[1], [torch.from_numpy(np.ascontiguousarray(x)).cuda() for x in (cv, cn, pv, pn)]))
    Rprev_inv = o.m3_inverse(T0["Rc2w"])
    angle = float(np.sin(np.float32(15.0) / np.float32(180.0) * np.pi))
    ws = torch.zeros(capi.icp_workspace_bytes(), dtype=torch.uint8, device="cuda")
    hip = C.CDLL("libamdhip64.so")
    pairs = C.c_void_p()
    assert hip.hipHostMalloc(C.byref(pairs), C.c_size_t(1024), C.c_uint(0x40000000 | 0x2)) == 0
    C.memset(pairs, 0, 1024)
    try:
        first = {}
        for i in range(4000):
            level = 0 if i % 2 == 0 else 2
            k, rows, cols, dv = levels[level]
            capi.icp_accumulate(T0["Rc2w"], T0["tc2w"], dv[0], dv[1], Rprev_inv, T0["tc2w"], k, dv[2], dv[3], cols * 8, rows, cols, 0.10, angle, ws,
                                pairs.value, done_flag=capi.ICP_PUBLISH_PAIRS, done_seq=i + 1)
            rc, got = capi.icp_wait_pairs(pairs.value, i + 1)
            assert rc == 0
            if level not in first:
                first[level] = got.copy()
                assert got[54] > 0.3 * rows * cols
            assert np.array_equal(got.view(np.int64), first[level].view(np.int64)), (i, level)
    finally:
        torch.cuda.synchronize()
        hip.hipHostFree(pairs)


def test_tsdf_residual_kernels_thousands_of_launches_512(dev):
    """2 000 launches of xs_compute_local_tsdf_hessian and 1 000 of xs_tsdf_gauss_newton_terms over a 512^3 map, fixed inputs: every
    result identical to the first."""
    torch, capi, pl = dev
    from independent_cases import dual_pose
    n = 512
    prm = synth.s1_params(n)
    res = [n, n, n]
    value = torch.empty((n * n, n), dtype=torch.float32, device="cuda")
    weight = torch.empty((n * n, n), dtype=torch.int32, device="cuda")
    grad = torch.empty((n * n, n), dtype=torch.float32, device="cuda")
    capi.init_volume(value, weight, grad, n * 4, res)
    scaled = torch.empty((H, W), dtype=torch.float32, device="cuda")
    for k in range(3):
        T = s1_transforms(k, prm)
        capi.integrate_tsdf_volume(torch.from_numpy(synth.s3_frame(k).view(np.int16)).cuda(), W * 2, H, W, intr_of(prm), 100, res, prm["tsdf_voxel_size"],
                                   T["Rv2c"], T["tv2c"], tranc_dist(prm), value, weight, grad, n * 4, scaled, W * 4)
    del weight, grad
    capi.scale_depth(torch.from_numpy(synth.s3_frame(3).view(np.int16)).cuda(), W * 2, H, W, scaled, W * 4)
    Rd, td = dual_pose(prm, 3, 1e-6)
    ws = torch.zeros(capi.tsdf_reduce_workspace_bytes(), dtype=torch.uint8, device="cuda")
    gt = value.reshape(-1)
    out = torch.zeros((2000, 4), dtype=torch.float64, device="cuda")
    for i in range(2000):
        capi.compute_local_tsdf_hessian(scaled, W * 4, H, W, intr_of(prm), res, prm["tsdf_voxel_size"], Rd, td, tranc_dist(prm), gt, ws, out[i])
    torch.cuda.synchronize()
    r = out.cpu().numpy()
    assert r[0, 3] > 100000 and len(np.unique(r.view(np.int64), axis=0)) == 1
    # the six-pose first-order kernel: real pose of frame 3 with the six unit seeds the orchestrator builds
    from test_gauss_newton_gpu import seeded_poses
    T3 = s1_transforms(3, prm, seed=None)
    v2c = np.eye(4)
    v2c[:3, :3] = np.asarray(T3["Rv2c"], np.float64).reshape(3, 3, 2)[..., 0]
    v2c[:3, 3] = np.asarray(T3["tv2c"], np.float64).reshape(3, 2)[:, 0]
    Rs, ts = seeded_poses(np.linalg.inv(v2c))
    out29 = torch.zeros((1000, 32), dtype=torch.float64, device="cuda")
    for i in range(1000):
        capi.tsdf_gauss_newton_terms(scaled, W * 4, H, W, intr_of(prm), res, prm["tsdf_voxel_size"], Rs, ts, tranc_dist(prm), gt, ws, out29[i])
    torch.cuda.synchronize()
    r = out29[:, :29].cpu().numpy()
    assert r[0, 28] > 100000 and len(np.unique(np.ascontiguousarray(r).view(np.int64), axis=0)) == 1


def test_posted_pose_loop_with_a_slow_host(dev):
    """The tracker with a random 50-500 us host sleep in front of every pose post (xs_kf_debug_post_delay): 40 frames of scene S3,
    no launch gives up waiting (no timeout, every frame tracked) and every pose is the one the undelayed run computes, bit for bit."""
    torch, capi, pl = dev
    n = 256
    prm = synth.s1_params(n)
    frames = [torch.from_numpy(synth.s3_frame(k).view(np.int16)).cuda() for k in range(40)]

    def run(delay):
        kf = pl.KinectFusion(prm)
        try:
            kf.debug_post_delay(*delay)
            poses = []
            for f in frames:
                assert kf.process_frame(f) == 1
                poses.append(kf.world2camera().copy())
            torch.cuda.synchronize()
            return np.stack(poses)
        finally:
            kf.close()
    plain = run((0, 0))
    slow = run((50, 500))          # (a launch that gave up on its pose fails its frame: process_frame != 1 above)
    assert np.array_equal(plain.view(np.int32), slow.view(np.int32))
