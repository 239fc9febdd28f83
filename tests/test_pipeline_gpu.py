"""GPU parity of the whole XKinectFusion pipeline (C++ orchestrator + HIP kernels, through the
C ABI) against (a) the committed fixtures generated with the reference's complex class and
(b) the live CPU oracle, on scene S1.  Tolerances (north_star: trajectory and per-voxel TSDF
within a stated float tolerance, CSFD derivatives within 1e-6 relative):
  * poses: |d| <= 1e-6 on every real entry (rotation entries and metres);
  * pose derivatives (imaginary parts / h): 1e-6 relative to the largest derivative entry of that
    pose, plus the propagated effect of flipped discrete decisions (<= 1e-4 relative by frame 4);
  * voxels: bit-exact except a flip budget of 2e-5 of the sampled voxels."""
import importlib
import os

import numpy as np
import pytest

from conftest import load_golden
from helpers import mismatch_fraction, synth

pytestmark = pytest.mark.gpu
H, W = synth.HEIGHT, synth.WIDTH


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available()
    return torch, importlib.import_module("x-slam_amd.pipeline")


FLIPS = 2e-3  # share of sampled voxels / pixels allowed to sit on the other side of a discrete decision


def frac_bad(ok_mask):
    ok_mask = np.asarray(ok_mask)
    return 0.0 if ok_mask.size == 0 else 1.0 - ok_mask.mean()


def upload(torch, d):
    return torch.from_numpy(d.view(np.int16)).cuda()


def pose_close(got, want, value_tol=1e-6, deriv_rel=1e-6):
    assert np.all(np.abs(got[..., 0] - want[..., 0]) <= value_tol), np.abs(got[..., 0] - want[..., 0]).max()
    scale = max(np.abs(want[..., 1]).max(), 1e-30)
    assert np.all(np.abs(got[..., 1] - want[..., 1]) <= deriv_rel * scale), (np.abs(got[..., 1] - want[..., 1]).max() / scale)


@pytest.mark.parametrize("solve_on_device", [True, False], ids=["device_solve", "host_solve"])
@pytest.mark.parametrize("name", ["pipeline_s1_n64.npz", "pipeline_s1_n96.npz"])
def test_pipeline_against_committed_fixture(dev, name, solve_on_device):
    torch, pl = dev
    g = load_golden(name)
    n = int(g["n"])
    kf = pl.KinectFusion(dict(synth.s1_params(n), icp_solve_on_device=solve_on_device))
    vox, pix = g["voxel_index"], g["pixel_index"]
    py, px = pix // W, pix % W
    frames = list(g["frames"])
    for k in range(max(frames) + 1):
        d = synth.s1_frame(k)
        assert int(d.astype(np.uint64).sum()) == int(g["depth_checksums"][k]), "synthetic depth differs from the fixture's"
        assert kf.process_frame(upload(torch, d)) == 1
        sums = g[f"sums_{k}"] if k in frames else None
        if k > 2:
            # Scene S1 at 12 / 8 cm voxels amplifies a last-digit difference ~100x per frame (the sphere alone holds the slide along the
            # wall).  Measured against these fixtures: frames 0-2 identical bits; frame 3: 3.2e-7 / 1.8e-5 (pose / derivative relative to
            # its largest entry) at 64^3, 6.7e-8 / 2.3e-6 at 96^3; frame 4: 1.0e-4 / 3.0e-3 and 1.0e-6 / 9.6e-5.  The bound is ten
            # times the larger of the two; beyond frame 4 see tests/test_trajectory_gpu.py (30 frames beside the oracle, with the
            # scene's own sensitivity measured) and the constrained-scene fixture test below, which is tight for every frame.
            value_tol, deriv_rel = {3: (5e-6, 2e-4), 4: (1e-3, 3e-2)}[k]
            pose_close(kf.world2camera(), g[f"w2c_{k}"], value_tol=value_tol, deriv_rel=deriv_rel)
            if sums is not None:
                assert abs(kf.last_U() - sums[4]) <= 0.02 * sums[4] and abs(kf.last_hits() - sums[5]) <= 0.02 * sums[5]
            continue
        pose_close(kf.world2camera(), g[f"w2c_{k}"], value_tol=1e-6, deriv_rel=1e-6)
        if sums is None:
            continue
        v, w, gr = kf.volume()
        assert abs(kf.last_U() - sums[4]) <= max(2, 1e-4 * sums[4])
        assert abs(kf.last_hits() - sums[5]) <= 3
        assert mismatch_fraction(w[vox], g[f"weight_{k}"]) <= 1e-3
        ok = w[vox] == g[f"weight_{k}"]
        # a flipped pixel pick / truncation test moves a voxel by O(1): budget, not all()
        assert frac_bad(np.abs(v[vox][ok] - g[f"value_{k}"][ok]) <= 1e-6) <= FLIPS
        gs = np.abs(g[f"grad_{k}"]).max()
        assert frac_bad(np.abs(gr[vox][ok] - g[f"grad_{k}"][ok]) <= 1e-6 * gs) <= FLIPS
        assert abs(v.astype(np.float64).sum() - sums[0]) <= 1e-5 * abs(sums[1])
        vm, nm = kf.map("vmaps_g_prev", 0), kf.map("nmaps_g_prev", 0)
        # normals are differences of neighbouring trilinear TSDF samples: a 1e-6 volume
        # perturbation (within the voxel tolerance above) shows up ~10x larger
        for got, want, t0 in ((vm, g[f"vmap_{k}"], 5e-6), (nm, g[f"nmap_{k}"], 5e-5)):
            gx = got[py, px]
            both = ~np.isnan(gx[:, 0]) & ~np.isnan(want[0][:, 0])
            assert (np.isnan(gx[:, 0]) != np.isnan(want[0][:, 0])).mean() <= 5e-3
            for p in range(3):
                gp = got[py + p * H, px]
                assert frac_bad(np.abs(gp[both, 0] - want[p][both, 0]) <= t0) <= 5e-3
        if k > 0:
            il, wl = kf.icp_log(), g[f"icp_{k}"]
            assert il.shape == wl.shape
            assert np.all(np.abs(il[:, 54] - wl[:, 54]) <= np.maximum(3, 2e-4 * wl[:, 54]))
            # first iteration: identical inputs up to the bilateral pixels; later iterations see the
            # pose update of the previous one
            for it, rel in ((0, 1e-6), (-1, 1e-4)):
                assert np.all(np.abs(il[it, 0:54:2] - wl[it, 0:54:2]) <= rel * np.abs(wl[it, 0:54:2]).max())
                assert np.all(np.abs(il[it, 1:54:2] - wl[it, 1:54:2]) <= rel * np.abs(wl[it, 1:54:2]).max())
    kf.close()


@pytest.mark.parametrize("solve_on_device", [True, False], ids=["device_solve", "host_solve"])
def test_pipeline_against_committed_fixture_ten_frames_constrained_scene(dev, solve_on_device):
    """The fixture test above can only be tight for two frames: scene S1 amplifies a last-digit difference ~10x per frame at
    these voxel sizes.  On the box room (every degree of freedom constrained; fixture made with the reference's complex class
    over ten frames, axial CSFD seed) the GPU pipeline must stay with the fixture to the last digits for ALL ten frames:
    poses |d| <= 2e-6, pose derivatives within 1e-3 of their largest entry (measured: identical bits for most frames), voxel
    and hit counts within 3, sampled voxels equal up to the flip budget, d pose(2,3) / d seed alive (0.95 ... 1.25)."""
    torch, pl = dev
    g = load_golden("pipeline_s3_n96.npz")
    n = int(g["n"])
    kf = pl.KinectFusion(dict(synth.s1_params(n, seed=(2, 3)), icp_solve_on_device=solve_on_device))
    vox = g["voxel_index"]
    for k in range(10):
        d = synth.s3_frame(k)
        assert int(d.astype(np.uint64).sum()) == int(g["depth_checksums"][k]), "synthetic depth differs from the fixture's"
        assert kf.process_frame(upload(torch, d)) == 1
        pose_close(kf.world2camera(), g[f"w2c_{k}"], value_tol=2e-6, deriv_rel=1e-3)
        assert 0.95 <= kf.world2camera()[2, 3, 1] / np.float32(1e-7) <= 1.25
        if k in (0, 1, 4, 9):
            sums = g[f"sums_{k}"]
            assert abs(kf.last_U() - sums[4]) <= 3 and abs(kf.last_hits() - sums[5]) <= 3
            v, w, gr = kf.volume()
            assert mismatch_fraction(w[vox], g[f"weight_{k}"]) <= 1e-3
            ok = w[vox] == g[f"weight_{k}"]
            assert frac_bad(np.abs(v[vox][ok] - g[f"value_{k}"][ok]) <= 1e-5) <= FLIPS
            gs = np.abs(g[f"grad_{k}"]).max()
            assert frac_bad(np.abs(gr[vox][ok] - g[f"grad_{k}"][ok]) <= 1e-4 * gs) <= FLIPS
    kf.close()


def test_pipeline_against_live_oracle_128(dev, oracle):
    """Whole volumes and every map level against the oracle pipeline run side by side."""
    torch, pl = dev
    from oracle.oracle import OracleKinFu, params_from_dict
    n = 128
    prm = synth.s1_params(n)
    kf = pl.KinectFusion(prm)
    ok_ = OracleKinFu(oracle, params_from_dict(prm))
    for k in range(3):
        d = synth.s1_frame(k)
        assert kf.process_frame(upload(torch, d)) == 1 and ok_.process_frame(d) == 1
        pose_close(kf.world2camera(), ok_.world2camera(), value_tol=1e-6 if k <= 1 else 2e-5, deriv_rel=1e-6 if k <= 1 else 1e-3)
        assert abs(kf.last_U() - ok_.last_U()) <= max(3, 1e-4 * ok_.last_U())
    v, w, g = kf.volume()
    ov, ow, og = ok_.volume()
    assert mismatch_fraction(w, ow) <= 1e-4
    same = w == ow
    assert frac_bad(np.abs(v[same] - ov[same]) <= 1e-4) <= 1e-4
    assert frac_bad(np.abs(g[same] - og[same]) <= 1e-3 * np.abs(og).max()) <= 1e-4
    for level in range(3):
        for which in ("depths_curr", "vmaps_curr", "nmaps_curr"):
            a, b = kf.map(which, level), ok_.map(which, level)
            rows = H >> level  # the sentinel lives in the x plane only; y/z planes hold stale data there
            nan_a, nan_b = np.isnan(a[:rows, :, 0]), np.isnan(b[:rows, :, 0])
            assert (nan_a != nan_b).mean() <= 1e-4
    kf.close()


def test_pipeline_quarter_size_sensor_against_live_oracle(dev, oracle):
    """A 320 x 240 sensor (pyramid 320 / 160 / 80 columns: 2.5 and 1.25 tiles per row at the coarse levels) through the whole
    pipeline, side by side with the oracle pipeline."""
    torch, pl = dev
    from oracle.oracle import OracleKinFu, params_from_dict
    w2, h2 = W // 2, H // 2
    cam = dict(width=w2, height=h2, fx=synth.FX / 2, fy=synth.FY / 2, cx=synth.CX / 2, cy=synth.CY / 2)
    prm = dict(synth.s1_params(96), depth_width=w2, depth_height=h2, fx=cam["fx"], fy=cam["fy"], cx=cam["cx"], cy=cam["cy"])
    kf = pl.KinectFusion(prm)
    ok_ = OracleKinFu(oracle, params_from_dict(prm))
    for k in range(4):
        d = synth.s1_frame(k, **cam)
        assert d.shape == (h2, w2)
        assert kf.process_frame(upload(torch, d)) == 1 and ok_.process_frame(d) == 1
        pose_close(kf.world2camera(), ok_.world2camera(), value_tol=1e-6 if k <= 1 else 5e-5, deriv_rel=1e-6 if k <= 1 else 3e-3)
        assert abs(kf.last_U() - ok_.last_U()) <= max(3, 1e-4 * ok_.last_U())
        if k >= 1:
            la, lb = kf.icp_log(), ok_.icp_log()
            assert la.shape == lb.shape and la.shape[0] == 12
            assert abs(la[0, 54] - lb[0, 54]) <= max(2, 1e-4 * lb[0, 54]) and lb[0, 54] > 0.5 * (w2 // 4) * (h2 // 4)
    v, w, g = kf.volume()
    ov, ow, og = ok_.volume()
    assert mismatch_fraction(w, ow) <= 1e-4
    kf.close()


@pytest.mark.parametrize("sensor", ["640x480", "320x240", "600x452"])
def test_model_map_pyramid_built_inside_the_raycast_launch(dev, sensor):
    """raycast_builds_pyramid (default): the one-launch raycast also halves its own pixel tiles twice — levels 1 and 2 of the model
    vertex / normal maps — against the pyramid as a launch of its own (false): poses, counts and all three levels of both maps, bit for
    bit, over six frames; a sensor whose width and height are no multiples of the 16 x 16 workgroup tile among them."""
    torch, pl = dev
    w_, h_ = (int(v) for v in sensor.split("x"))
    sx, sy = w_ / W, h_ / H
    cam = dict(width=w_, height=h_, fx=synth.FX * sx, fy=synth.FY * sy, cx=synth.CX * sx, cy=synth.CY * sy)
    prm = dict(synth.s1_params(128), depth_width=w_, depth_height=h_, fx=cam["fx"], fy=cam["fy"], cx=cam["cx"], cy=cam["cy"])
    a, b = pl.KinectFusion(dict(prm, raycast_builds_pyramid=True)), pl.KinectFusion(dict(prm, raycast_builds_pyramid=False))
    for k in range(6):
        d = upload(torch, synth.s1_frame(k, **cam) if sensor != "640x480" else synth.s1_frame(k))
        assert a.process_frame(d) == 1 and b.process_frame(d) == 1
        assert np.array_equal(a.world2camera(), b.world2camera()) and a.last_hits() == b.last_hits() and a.last_U() == b.last_U()
        for which in ("vmaps_g_prev", "nmaps_g_prev"):
            for level in range(3):
                x, y = a.map(which, level), b.map(which, level)
                rows_ = x.shape[0] // 3
                assert rows_ == h_ >> level and x.shape[1] == w_ >> level
                valid = np.isfinite(y[:rows_, :, 0])
                assert np.array_equal(valid, np.isfinite(x[:rows_, :, 0])), (k, which, level)
                assert valid.mean() > 0.5
                for p_ in range(3):
                    assert np.array_equal(x[p_ * rows_:(p_ + 1) * rows_][valid].view(np.int32), y[p_ * rows_:(p_ + 1) * rows_][valid].view(np.int32)), (k, which, level, p_)
    a.close(); b.close()


@pytest.mark.parametrize("levels", [1, 2])
def test_pipeline_fewer_pyramid_levels_against_live_oracle(dev, oracle, levels):
    """num_levels 1 and 2 (the model-map pyramid then goes through the separate resize launches, the ICP runs 5 or 5 + 4
    iterations) side by side with the oracle pipeline."""
    torch, pl = dev
    from oracle.oracle import OracleKinFu, params_from_dict
    prm = dict(synth.s1_params(96), num_levels=levels)
    kf = pl.KinectFusion(prm)
    ok_ = OracleKinFu(oracle, params_from_dict(prm))
    for k in range(4):
        d = synth.s1_frame(k)
        assert kf.process_frame(upload(torch, d)) == 1 and ok_.process_frame(d) == 1
        pose_close(kf.world2camera(), ok_.world2camera(), value_tol=1e-6 if k <= 1 else 5e-5, deriv_rel=1e-6 if k <= 1 else 3e-3)
        assert abs(kf.last_U() - ok_.last_U()) <= max(3, 1e-4 * ok_.last_U())
        if k >= 1:
            assert kf.icp_log().shape == ok_.icp_log().shape == ((5, 9)[levels - 1], 55)
    kf.close()


def test_brick_list_classified_ahead_changes_nothing(dev):
    """integrate_classify_ahead (the integrate call's brick list built behind the last ICP launch, from the pose that launch
    starts from) against the plain order and against a list slack of 1 — with which no final pose is ever covered, so every frame
    takes the fall-back (header cleared again, classification repeated); and integrate_post_pose (the integrate kernel itself
    enqueued behind the classification and handed the final pose through its mailbox), covered and — slack 1 — never covered (the
    posted launch told to leave every frame): the same poses, counts and volume, bit for bit.  Also the two scheduling switches that
    were measured and left off: the classification on the auxiliary stream beside the ICP launch (integrate_classify_beside_icp: the
    integrate launch waits for its completion event) and one / two ICP iterations early (integrate_classify_early: a pose that many
    more updates old — some frames then classify again, some decide only the boxes again); and the classification at the frame's start
    for a predicted pose (integrate_classify_predicted), at the usual slack and at a wide one."""
    torch, pl = dev
    prm = synth.s1_params(128)
    runs = [pl.KinectFusion(dict(prm, integrate_classify_ahead=False)), pl.KinectFusion(dict(prm, integrate_classify_ahead=True)),
            pl.KinectFusion(dict(prm, integrate_classify_ahead=True, integrate_classify_slack=1.0)),
            pl.KinectFusion(dict(prm, integrate_classify_ahead=True, integrate_post_pose=True)),
            pl.KinectFusion(dict(prm, integrate_classify_ahead=True, integrate_post_pose=True, integrate_classify_slack=1.0)),
            pl.KinectFusion(dict(prm, integrate_classify_ahead=True, integrate_classify_beside_icp=True)),
            pl.KinectFusion(dict(prm, integrate_classify_ahead=True, integrate_classify_beside_icp=True, integrate_classify_early=1)),
            pl.KinectFusion(dict(prm, integrate_classify_ahead=True, integrate_classify_early=2)),
            pl.KinectFusion(dict(prm, integrate_classify_ahead=True, integrate_classify_beside_icp=True, integrate_post_pose=True)),
            pl.KinectFusion(dict(prm, integrate_classify_ahead=True, integrate_classify_predicted=True)),
            pl.KinectFusion(dict(prm, integrate_classify_ahead=True, integrate_classify_predicted=True, integrate_classify_slack=6.0))]
    blank = upload(torch, np.zeros_like(synth.s1_frame(0)))
    for k in list(range(6)) + ["blank", 6, 7]:
        d = blank if k == "blank" else upload(torch, synth.s1_frame(k))
        rcs = [r.process_frame(d) for r in runs]
        assert all(rc == (0 if k == "blank" else 1) for rc in rcs)
        for r in runs[1:]:
            assert np.array_equal(r.world2camera(), runs[0].world2camera())
            assert r.last_U() == runs[0].last_U() and r.last_hits() == runs[0].last_hits()
    v0, w0, g0 = runs[0].volume()
    for r in runs[1:]:
        v, w, g = r.volume()
        assert np.array_equal(w, w0) and np.array_equal(v, v0) and np.array_equal(g, g0)
    for r in runs:
        r.close()


@pytest.mark.parametrize("n,scene", [(128, "s1"), (512, "s1"), (256, "s3")])
def test_ray_march_from_the_sign_map_changes_nothing(dev, tmp_path, n, scene):
    """raycast_sign_map (the integrate kernel marks the bricks it writes negative values into, the march starts each ray at the first
    step that can end it: csrc/xs_signmap.h) against the march of every step, with bricks of 8^3 and of 16^3 voxels: the same poses,
    counts, model maps and volume, bit for bit, over 12 frames — a blank frame among them — and after a checkpoint is loaded into a
    fresh pipeline (whose map is rebuilt from the restored volume)."""
    torch, pl = dev
    prm = synth.s1_params(n)    # (S3 — the box room — shares S1's placement and camera path)
    frame = synth.s1_frame if scene == "s1" else synth.s3_frame
    runs = [pl.KinectFusion(dict(prm, raycast_sign_map=False)), pl.KinectFusion(dict(prm, raycast_sign_map=True)),
            pl.KinectFusion(dict(prm, raycast_sign_map=True, raycast_sign_map_shift=4))]
    blank = upload(torch, np.zeros_like(frame(0)))

    def maps_same(r):
        # every ray that hit, all three planes (the y and z planes of a ray that missed are not written: they keep what the buffer held)
        H_ = synth.HEIGHT
        for which in ("vmaps_g_prev", "nmaps_g_prev"):
            a_, b_ = r.map(which, 0), runs[0].map(which, 0)
            valid = np.isfinite(b_[:H_, :, 0])
            assert np.array_equal(valid, np.isfinite(a_[:H_, :, 0]))
            for p_ in range(3):
                assert np.array_equal(a_[p_ * H_:(p_ + 1) * H_][valid].view(np.int32), b_[p_ * H_:(p_ + 1) * H_][valid].view(np.int32))

    def same(r):
        assert np.array_equal(r.world2camera(), runs[0].world2camera())
        assert r.last_U() == runs[0].last_U() and r.last_hits() == runs[0].last_hits()
        maps_same(r)

    for k in list(range(5)) + ["blank"] + list(range(5, 11)):
        d = blank if k == "blank" else upload(torch, frame(k))
        rcs = [r.process_frame(d) for r in runs]
        assert all(rc == (0 if k == "blank" else 1) for rc in rcs), (k, rcs)
        for r in runs[1:]:
            same(r)
    assert runs[0].last_hits() > 0.5 * synth.HEIGHT * synth.WIDTH
    path = str(tmp_path / "vol.ckpt")
    runs[0].save_checkpoint(path)
    fresh = pl.KinectFusion(dict(prm, raycast_sign_map=True))
    assert fresh.load_checkpoint(path)
    runs.append(fresh)
    maps_same(fresh)   # the maps the load regenerated from the restored volume
    for k in (11, 12):
        d = upload(torch, frame(k))
        assert all(r.process_frame(d) == 1 for r in runs)
        for r in runs[1:]:
            same(r)
    v0, w0, g0 = runs[0].volume()
    for r in runs[1:]:
        v, w, g = r.volume()
        assert np.array_equal(w, w0) and np.array_equal(v, v0) and np.array_equal(g, g0)
    for r in runs:
        r.close()


def test_volume_written_through_its_pointer_invalidates_the_sign_map(dev):
    """A caller that writes the value array through volume_ptr (xs_kf_volume_ptr) need not know about the sign map: asking for the
    pointer marks the map stale and the next raycast rebuilds it from the volume.  A slab of negative values is written into free space
    in front of the camera of two pipelines — one with the map, one marching every step — and the next frames give the same poses,
    counts and model maps (the rays now stop at the slab)."""
    torch, pl = dev
    n = 128
    prm = synth.s1_params(n)
    runs = [pl.KinectFusion(dict(prm, raycast_sign_map=False)), pl.KinectFusion(dict(prm, raycast_sign_map=True))]
    for k in range(3):
        d = upload(torch, synth.s1_frame(k))
        assert all(r.process_frame(d) == 1 for r in runs)
    hits_before = runs[0].last_hits()
    for r in runs:
        ptr, step = r.volume_ptr("value")
        assert step == n * 4

        class View:
            __cuda_array_interface__ = {"shape": (n, n, n), "typestr": "<f4", "data": (int(ptr), False), "version": 2}
        vol = torch.as_tensor(View(), device="cuda")
        cz = int(prm["init_z"] / prm["tsdf_voxel_size"])
        vol[cz + 10:cz + 12, 40:90, 30:100] = -0.5        # a negative slab 0.6 m in front of the camera, in bricks the integrate kernels never marked
        torch.cuda.synchronize()
    for k in (3, 4):
        d = upload(torch, synth.s1_frame(k))
        rcs = [r.process_frame(d) for r in runs]
        assert rcs[0] == rcs[1]
        assert np.array_equal(runs[0].world2camera(), runs[1].world2camera())
        assert runs[0].last_hits() == runs[1].last_hits() and runs[0].last_U() == runs[1].last_U()
        H_ = synth.HEIGHT
        a_, b_ = runs[1].map("vmaps_g_prev", 0), runs[0].map("vmaps_g_prev", 0)
        valid = np.isfinite(b_[:H_, :, 0])
        assert np.array_equal(valid, np.isfinite(a_[:H_, :, 0]))
        assert np.array_equal(a_[:H_][valid].view(np.int32), b_[:H_][valid].view(np.int32))
    assert runs[0].last_hits() != hits_before            # the slab changed what the rays see
    for r in runs:
        r.close()


def test_sign_map_over_a_long_trajectory(dev):
    """Sixty frames of the box room at 256^3 — the camera sweeps, surfaces enter and leave the view, bricks marked early stay marked —
    with and without the sign map: every pose, count and the final volume identical, bit for bit."""
    torch, pl = dev
    prm = synth.s1_params(256)
    a, b = pl.KinectFusion(dict(prm, raycast_sign_map=False)), pl.KinectFusion(dict(prm, raycast_sign_map=True))
    for k in range(60):
        d = upload(torch, synth.s3_frame(k))
        assert a.process_frame(d) == 1 and b.process_frame(d) == 1, k
        assert np.array_equal(a.world2camera(), b.world2camera()), k
        assert a.last_U() == b.last_U() and a.last_hits() == b.last_hits(), k
    for x, y in zip(a.volume(), b.volume()):
        assert np.array_equal(x, y)
    a.close(); b.close()


def test_announcing_the_next_frame_changes_nothing(dev):
    """hint_next_frame (the next frame's bilateral filter and depth pyramid built during this frame's ICP loop) against the plain order: the
    same poses, counts, current-frame maps and volume, bit for bit — also when the announced frame is not the one that comes (the pipeline
    then prepares its own maps), when a frame is announced twice, when tracking fails in between (a blank frame), and for the frame after
    the last announcement."""
    torch, pl = dev
    prm = synth.s1_params(128)
    a, b = pl.KinectFusion(prm), pl.KinectFusion(prm)
    frames = [upload(torch, synth.s1_frame(k)) for k in range(12)]
    blank = upload(torch, np.zeros_like(synth.s1_frame(0)))
    order = [0, 1, 2, 3, "blank", 4, 5, 6, 7, 8, 9, 10, 11]
    tens = lambda k: blank if k == "blank" else frames[k]
    for n, k in enumerate(order):
        nxt = order[n + 1] if n + 1 < len(order) else None
        if k == 5:
            b.hint_next_frame(frames[9])            # a wrong announcement: frame 6 comes
        elif k == 8:
            b.hint_next_frame(tens(nxt)); b.hint_next_frame(tens(nxt))
        elif nxt is not None and k != 10:           # (nothing announced while frame 10 is processed: frame 11 prepares itself)
            b.hint_next_frame(tens(nxt))
        ra, rb = a.process_frame(tens(k)), b.process_frame(tens(k))
        assert ra == rb == (0 if k == "blank" else 1), (k, ra, rb)
        assert np.array_equal(a.world2camera(), b.world2camera()), k
        assert a.last_U() == b.last_U() and a.last_hits() == b.last_hits(), k
        for level in range(3):
            assert np.array_equal(a.map("depths_curr", level).view(np.int32), b.map("depths_curr", level).view(np.int32)), (k, level)
            for which in ("vmaps_curr", "nmaps_curr"):   # (an invalid pixel has NaN in its x plane; its y and z planes are not written)
                ma, mb = a.map(which, level), b.map(which, level)
                rows = ma.shape[0] // 3
                valid = np.isfinite(ma[:rows, :, 0])
                assert np.array_equal(valid, np.isfinite(mb[:rows, :, 0])), (k, which, level)
                for p_ in range(3):
                    assert np.array_equal(ma[p_ * rows:(p_ + 1) * rows][valid].view(np.int32), mb[p_ * rows:(p_ + 1) * rows][valid].view(np.int32)), (k, which, level)
    for x, y in zip(a.volume(), b.volume()):
        assert np.array_equal(x, y)
    a.close(); b.close()


def test_device_pose_solve_matches_host_solve(dev):
    """The two shapes of the ICP loop — pose update on the device, one host wait per frame (default) and
    the reference's one host solve per iteration — on the same frames: identical first-iteration sums,
    poses equal to float rounding of three sin / cos pairs per iteration."""
    torch, pl = dev
    prm = synth.s1_params(128)
    a = pl.KinectFusion(dict(prm, icp_solve_on_device=True))
    b = pl.KinectFusion(dict(prm, icp_solve_on_device=False))
    for k in range(4):
        d = upload(torch, synth.s1_frame(k))
        assert a.process_frame(d) == 1 and b.process_frame(d) == 1
        la, lb = a.icp_log(), b.icp_log()
        assert la.shape == lb.shape
        if k == 1:
            assert np.array_equal(la[0], lb[0])   # same maps, same starting pose: same bits
            assert np.all(np.abs(la[-1, :54] - lb[-1, :54]) <= 1e-5 * np.abs(lb[-1, :54]).max())
        if k >= 1:
            assert la.shape[0] == 12
            pose_close(a.world2camera(), b.world2camera(), value_tol=1e-6 if k == 1 else 1e-5, deriv_rel=1e-6 if k == 1 else 1e-3)
    assert abs(a.last_U() - b.last_U()) <= max(3, 1e-4 * b.last_U())
    a.close(); b.close()


@pytest.mark.parametrize("lookahead", [1, 4])
def test_posted_pose_loop_is_bit_identical_to_launch_after_solve(dev, lookahead):
    """icp_post_pose (default: every iteration's launch enqueued before its pose is known — one launch ahead, or a
    queue of icp_lookahead of them, which a lost frame's abandon command must empty in one post —, pose delivered
    through the BAR mailbox) against the reference's order (launch after the solve): the same kernel
    arithmetic on the same inputs — every sum of every iteration, every pose and the volume must be the
    same bits; and the frame after a lost frame (all-zero depth: no inliers, singular system, ProcessFrame
    returns 0 while a posted launch is still in flight) is processed normally."""
    torch, pl = dev
    prm = synth.s1_params(96)
    a = pl.KinectFusion(dict(prm, icp_post_pose=True, icp_lookahead=lookahead))
    b = pl.KinectFusion(dict(prm, icp_post_pose=False))
    for k in range(5):
        d = upload(torch, synth.s1_frame(k))
        assert a.process_frame(d) == 1 and b.process_frame(d) == 1
        assert np.array_equal(a.icp_log(), b.icp_log())
        assert np.array_equal(a.world2camera(), b.world2camera())
    blank = upload(torch, np.zeros_like(synth.s1_frame(0)))
    assert a.process_frame(blank) == 0 and b.process_frame(blank) == 0
    assert a.icp_log().shape[0] == 1 and np.array_equal(a.icp_log(), b.icp_log())
    for k in range(5, 7):
        d = upload(torch, synth.s1_frame(k))
        assert a.process_frame(d) == 1 and b.process_frame(d) == 1
        assert np.array_equal(a.icp_log(), b.icp_log())
        assert np.array_equal(a.world2camera(), b.world2camera())
    va, wa, ga = a.volume()
    vb, wb, gb = b.volume()
    assert np.array_equal(wa, wb) and np.array_equal(va, vb) and np.array_equal(ga, gb)
    assert a.last_U() == b.last_U() > 0
    a.close(); b.close()


def test_posted_pose_numbers_cross_the_2_to_32_wrap(dev):
    """The mailbox sequence number of a posted launch is the low word of its completion number, and a poller accepts any post at or
    after its own number in the modulo-2^32 order: the numbers must stay monotone across the wrap, with 0 (the mailbox's initial word)
    skipped.  A run whose launch numbers start 30 below 2^32 — the wrap falls inside frame 3's twelve launches — gives the same sums,
    poses and volume as a run numbered from 0, and no launch times out."""
    torch, pl = dev
    prm = synth.s1_params(96)
    a = pl.KinectFusion(dict(prm, icp_post_pose=True))
    b = pl.KinectFusion(dict(prm, icp_post_pose=True))
    d0 = upload(torch, synth.s1_frame(0))
    assert a.process_frame(d0) == 1 and b.process_frame(d0) == 1
    a.debug_set_icp_sequence((1 << 32) - 30)
    for k in range(1, 7):
        d = upload(torch, synth.s1_frame(k))
        assert a.process_frame(d) == 1 and b.process_frame(d) == 1, k
        assert np.array_equal(a.icp_log(), b.icp_log()), k
        assert np.array_equal(a.world2camera(), b.world2camera()), k
    for x, y in zip(a.volume(), b.volume()):
        assert np.array_equal(x, y)
    times = a.icp_iteration_times()
    assert [times[lv][1] for lv in range(4)] == [6 * 5, 6 * 4, 6 * 2, 6] and all(0.0 < times[lv][0] / times[lv][1] < 5e4 for lv in range(4))
    a.close(); b.close()


def test_alignment_failure_after_the_bricks_were_classified_ahead(dev):
    """The brick classification of the integrate call is enqueued behind the last ICP launch; if that last iteration then fails (forced
    here: the determinant gate of iteration 11), IntegrateFrame never runs and the list must not survive into the retried frame — whose
    map preparation clears the workspace header and rewrites the depth maximum on the auxiliary stream.  The retry and the following
    frames give the same poses, counts and volume as a run that never failed."""
    torch, pl = dev
    prm = synth.s1_params(128)
    a = pl.KinectFusion(dict(prm, integrate_classify_ahead=True, integrate_post_pose=True))   # (the failure path of the posted launch too)
    b = pl.KinectFusion(dict(prm, integrate_classify_ahead=True))
    for k in range(3):
        d = upload(torch, synth.s1_frame(k))
        assert a.process_frame(d) == 1 and b.process_frame(d) == 1
    for k in range(3, 7):
        d = upload(torch, synth.s1_frame(k))
        if k in (3, 5):
            a.debug_fail_icp_iteration(11)
            assert a.process_frame(d) == 0 and a.frame_id == k     # lost: the same frame is offered again
        assert a.process_frame(d) == 1 and b.process_frame(d) == 1
        assert np.array_equal(a.world2camera(), b.world2camera()), k
        assert a.last_U() == b.last_U() and a.last_hits() == b.last_hits(), k
    for x, y in zip(a.volume(), b.volume()):
        assert np.array_equal(x, y)
    a.close(); b.close()


def test_pipeline_gt_pose_mode_s2(dev, oracle):
    """flag_use_gtPose: only surface measure + integrate + raycast run (scene S2 at a small size)."""
    torch, pl = dev
    from oracle.oracle import OracleKinFu, params_from_dict
    prm = synth.s2_params(64)
    gt = np.zeros((2, 4, 4, 2), np.float32)
    gt[:, [0, 1, 2, 3], [0, 1, 2, 3], 0] = 1.0
    gt[:, 0, 3, 1] = 1e-7  # a CSFD seed riding on the given pose
    kf = pl.KinectFusion(prm, gt_poses=gt)
    ok_ = OracleKinFu(oracle, params_from_dict(prm), gt_poses=gt)
    d = synth.render_s2()
    for k in range(2):
        assert kf.process_frame(upload(torch, d)) == 1 and ok_.process_frame(d) == 1
    assert kf.last_U() == ok_.last_U() and kf.last_U() > 0.2 * 64 ** 3
    v, w, g = kf.volume()
    ov, ow, og = ok_.volume()
    assert np.array_equal(w, ow) and mismatch_fraction(v, ov) <= 2e-5 and mismatch_fraction(g, og) <= 2e-5
    kf.close()


def test_checkpoint_roundtrip(dev, tmp_path):
    torch, pl = dev
    prm = synth.s1_params(64)
    a = pl.KinectFusion(prm)
    for k in range(3):
        assert a.process_frame(upload(torch, synth.s1_frame(k))) == 1
    path = str(tmp_path / "vol.ckpt")
    a.save_checkpoint(path)
    a.save_tsdf_volume(str(tmp_path / "tsdf.bin"))
    assert os.path.getsize(str(tmp_path / "tsdf.bin")) == 64 ** 3 * 4  # X*Y*Z floats
    b = pl.KinectFusion(prm)
    assert b.load_checkpoint(path)
    assert b.frame_id == 3 and b.num_poses() == a.num_poses()
    for x, y in zip(a.volume(), b.volume()):
        assert np.array_equal(x, y)
    # both continue identically from the restored state
    d = upload(torch, synth.s1_frame(3))
    assert a.process_frame(d) == 1 and b.process_frame(d) == 1
    assert np.array_equal(a.world2camera(), b.world2camera())
    for x, y in zip(a.volume(), b.volume()):
        assert np.array_equal(x, y)
    a.close(); b.close()


def test_checkpoint_rejects_bad_files_without_touching_state(dev, tmp_path):
    """loadCheckpoint validates the whole file before it changes anything: a wrong magic, another resolution / voxel size,
    a zero, negative or absurd pose count, a truncated file and trailing bytes all return false and leave the pipeline
    exactly where it was (the next frame gives the same bits as an instance that never saw the bad file)."""
    import struct
    torch, pl = dev
    prm = synth.s1_params(64)
    a, ref = pl.KinectFusion(prm), pl.KinectFusion(prm)
    for k in range(3):
        d = upload(torch, synth.s1_frame(k))
        assert a.process_frame(d) == 1 and ref.process_frame(d) == 1
    good = str(tmp_path / "good.ckpt")
    a.save_checkpoint(good)
    blob = open(good, "rb").read()
    hdr = struct.Struct("<8s3iff2i4i")            # magic, res[3], voxel_size, tranc_dist, frame_id, n_poses, zs0, zs1, rank, count
    f = list(hdr.unpack_from(blob))
    assert f[0].rstrip(b"\0") == b"XSTSDF2" and f[1:4] == [64, 64, 64] and f[7] == 3 and f[8:10] == [0, 64] and f[10:12] == [0, 1]

    def variant(name, **kw):
        g = list(f)
        idx = dict(magic=0, res0=1, voxel_size=4, tranc_dist=5, frame_id=6, n_poses=7, zs0=8, zs1=9, rank=10, count=11)
        tail = kw.pop("tail", None)
        cut = kw.pop("cut", None)
        for k_, v in kw.items():
            g[idx[k_]] = v
        data = hdr.pack(*g) + blob[hdr.size:]
        if cut is not None:
            data = data[:cut]
        if tail is not None:
            data += tail
        path = str(tmp_path / f"{name}.ckpt")
        open(path, "wb").write(data)
        return path
    bad = [variant("magic", magic=b"XSTSDF1\0"), variant("res", res0=32), variant("voxel", voxel_size=0.5), variant("trunc", tranc_dist=1.0),
           variant("poses0", n_poses=0), variant("posesneg", n_poses=-5), variant("poseshuge", n_poses=2 ** 30), variant("frameneg", frame_id=-1),
           variant("planes", zs1=32), variant("rank", rank=1, count=2), variant("short", cut=len(blob) - 4096), variant("header_only", cut=hdr.size),
           variant("tiny", cut=10), variant("trailing", tail=b"\0" * 16), str(tmp_path / "does_not_exist.ckpt")]
    poses_before, frame_before = a.num_poses(), a.frame_id
    for path in bad:
        assert not a.load_checkpoint(path), path
        assert a.num_poses() == poses_before and a.frame_id == frame_before
    d = upload(torch, synth.s1_frame(3))
    assert a.process_frame(d) == 1 and ref.process_frame(d) == 1
    assert np.array_equal(a.world2camera(), ref.world2camera())
    for x, y in zip(a.volume(), ref.volume()):
        assert np.array_equal(x, y)
    # and the good file still loads into a fresh instance
    b = pl.KinectFusion(prm)
    assert b.load_checkpoint(good) and b.frame_id == 3
    a.close(); b.close(); ref.close()


def test_host_and_device_depth_entry_points_agree(dev):
    torch, pl = dev
    prm = synth.s1_params(64)
    a, b = pl.KinectFusion(prm), pl.KinectFusion(prm)
    for k in range(2):
        d = synth.s1_frame(k)
        assert a.process_frame(upload(torch, d)) == 1 and b.process_frame_host(d) == 1
    assert np.array_equal(a.world2camera(), b.world2camera())
    a.close(); b.close()


def test_pinned_ingest_buffers(dev):
    """Frames written into the pipeline's own pinned staging buffers (no staging copy, asynchronous upload on
    the second stream, two buffers alternating) give the same poses and volume as device-resident frames."""
    torch, pl = dev
    prm = synth.s1_params(64)
    a, b = pl.KinectFusion(prm), pl.KinectFusion(prm)
    seen = set()
    for k in range(5):
        d = synth.s1_frame(k)
        buf = b.ingest_buffer()
        seen.add(buf.ctypes.data)
        buf[...] = d
        assert a.process_frame(upload(torch, d)) == 1 and b.process_frame_host(buf) == 1
    assert len(seen) == 2
    assert np.array_equal(a.world2camera(), b.world2camera())
    for x, y in zip(a.volume(), b.volume()):
        assert np.array_equal(x, y)
    a.close(); b.close()


def test_alignment_failure_does_not_advance(dev):
    """A frame with no valid depth gives a singular 6x6 system: ProcessFrame returns 0 and frame_id
    stays (KinectFusionReconstruction.cpp:151-157, :203-210); the next good frame still tracks."""
    torch, pl = dev
    prm = synth.s1_params(64)
    kf = pl.KinectFusion(prm)
    assert kf.process_frame(upload(torch, synth.s1_frame(0))) == 1 and kf.frame_id == 1
    blank = np.zeros((H, W), np.uint16)
    poses = kf.num_poses()
    assert kf.process_frame(upload(torch, blank)) == 0
    assert kf.frame_id == 1 and kf.num_poses() == poses
    assert kf.process_frame(upload(torch, synth.s1_frame(1))) == 1 and kf.frame_id == 2
    kf.close()


@pytest.mark.parametrize("levels,width,height", [(2, 640, 480), (3, 320, 240), (1, 640, 480)])
def test_other_pyramid_depths_and_image_sizes(dev, oracle, levels, width, height):
    torch, pl = dev
    from oracle.oracle import OracleKinFu, params_from_dict
    prm = synth.s1_params(64)
    s = width / 640.0
    prm.update(num_levels=levels, depth_width=width, depth_height=height, fx=synth.FX * s, fy=synth.FY * s,
               cx=(synth.CX + 0.5) * s - 0.5, cy=(synth.CY + 0.5) * s - 0.5)
    kf = pl.KinectFusion(prm)
    ok_ = OracleKinFu(oracle, params_from_dict(prm))
    for k in range(2):
        d = synth.render_s1(synth.s1_pose(k), width=width, height=height, fx=prm["fx"], fy=prm["fy"], cx=prm["cx"], cy=prm["cy"])
        assert kf.process_frame(upload(torch, d)) == 1 and ok_.process_frame(d) == 1
        pose_close(kf.world2camera(), ok_.world2camera(), value_tol=1e-6, deriv_rel=1e-6)
    assert abs(kf.last_U() - ok_.last_U()) <= max(3, 1e-4 * ok_.last_U())
    v, w, g = kf.volume()
    ov, ow, og = ok_.volume()
    assert mismatch_fraction(w, ow) <= 1e-4
    kf.close()


def test_seven_scenes_intrinsics_against_live_oracle(dev, oracle):
    """BASELINE configs 3 and 5 name 7-Scenes: its Kinect intrinsics (585, 585, 320, 240 — fy positive, unlike the ICL file's
    -480) on the box room rendered through those intrinsics, GPU pipeline beside the oracle pipeline for six frames: same
    tolerances as the ICL-intrinsics runs, trajectory within a third of a voxel of the ground truth, derivative of the axial seed alive."""
    torch, pl = dev
    from oracle.oracle import OracleKinFu, params_from_dict
    prm = dict(synth.s1_params(128, seed=(2, 3)), **synth.SEVEN_SCENES)
    kf = pl.KinectFusion(prm)
    ok_ = OracleKinFu(oracle, params_from_dict(prm))
    for k in range(6):
        d = synth.s3_frame(k, **synth.SEVEN_SCENES)
        assert kf.process_frame(upload(torch, d)) == 1 and ok_.process_frame(d) == 1
        pose_close(kf.world2camera(), ok_.world2camera(), value_tol=2e-6, deriv_rel=1e-3)
        assert abs(kf.last_U() - ok_.last_U()) <= 3 and abs(kf.last_hits() - ok_.last_hits()) <= 3
    c2w = np.linalg.inv(kf.world2camera()[..., 0].astype(np.float64))
    gt = np.linalg.inv(synth.s1_pose(0)) @ synth.s1_pose(5)
    assert np.linalg.norm(c2w[:3, 3] - gt[:3, 3]) <= 0.02      # 6 cm voxels at 128^3: a third of a voxel
    assert 0.9 <= kf.world2camera()[2, 3, 1] / np.float32(1e-7) <= 1.3
    kf.close(); ok_.close()


@pytest.mark.parametrize("n", [256, 512])
def test_full_size_first_frame_figures(dev, n):
    """BASELINE configs 2 and 3 at their full sizes: after frame 0 of scene S1 the voxels written and the rays that
    found a surface equal the figures recorded from the reference's own kernel bodies, and a
    second pipeline fed the same frames reproduces poses, counters and maps bit for bit (determinism)."""
    import json, os
    torch, pl = dev
    fig = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "survey_reference_kernel_figures.json")))
    prm = synth.s1_params(n)
    a, b = pl.KinectFusion(prm), pl.KinectFusion(prm)
    d0 = upload(torch, synth.s1_frame(0))
    assert a.process_frame(d0) == 1 and b.process_frame(d0) == 1
    assert abs(a.last_U() - fig["integrate_U"][str(n)]) <= max(2, 1e-4 * fig["integrate_U"][str(n)])
    assert abs(a.last_hits() - fig["raycast_hits"][str(n)]) <= 2
    for k in (1, 2):
        d = upload(torch, synth.s1_frame(k))
        assert a.process_frame(d) == 1 and b.process_frame(d) == 1
    assert np.array_equal(a.world2camera(), b.world2camera())
    assert a.last_U() == b.last_U() and a.last_hits() == b.last_hits()
    ma, mb = a.map("vmaps_g_prev", 0), b.map("vmaps_g_prev", 0)
    na = np.isnan(ma[:H, :, 0])
    assert np.array_equal(na, np.isnan(mb[:H, :, 0])) and np.array_equal(ma[:H][~na], mb[:H][~na])
    a.close(); b.close()


def test_full_size_1024_tracks_like_512(dev):
    """The largest volume BASELINE names (1024^3: 4 GiB per array — exactly what 32-bit byte offsets reach, so the raycast
    still takes its 32-bit-offset kernels; the 64-bit ones are exercised by tests/test_large_volume_gpu.py on a volume of
    1024 x 1024 x 1040): four frames of scene S1 track, the camera ends within millimetres of where the
    512^3 run puts it (the scene constrains sliding along its wall only through the sphere, and the two
    discretisations slide differently: 4 mm apart after 27 mm of motion) and has moved as far as the scene's ground
    truth says; the rays find the same surface, and the voxel count scales with the resolution (a truncation band
    of fixed metric width: eight times the voxels at twice the resolution)."""
    torch, pl = dev
    big, ref = pl.KinectFusion(synth.s1_params(1024)), pl.KinectFusion(synth.s1_params(512))
    position = lambda kf: np.linalg.inv(kf.world2camera()[..., 0].astype(np.float64))[:3, 3]
    start = None
    for k in range(4):
        d = upload(torch, synth.s1_frame(k))
        assert big.process_frame(d) == 1 and ref.process_frame(d) == 1
        if k == 0:
            start = position(big)
    apart = np.linalg.norm(position(big) - position(ref))
    assert apart <= 0.006, f"1024^3 and 512^3 camera positions differ by {apart * 1e3:.2f} mm"
    gt_step = np.linalg.norm(synth.s1_pose(3, 300)[:3, 3] - synth.s1_pose(0, 300)[:3, 3])
    moved = np.linalg.norm(position(big) - start)
    assert gt_step > 0.02 and abs(moved - gt_step) <= 0.003, f"moved {moved * 1e3:.2f} mm, ground truth {gt_step * 1e3:.2f} mm"
    assert abs(big.last_hits() - ref.last_hits()) <= 0.02 * ref.last_hits()
    assert 7.0 <= big.last_U() / ref.last_U() <= 9.0
    deriv = big.world2camera()[..., 1]
    assert np.isfinite(deriv).all() and np.abs(deriv).max() > 0     # the CSFD seed's derivative rides along
    big.close(); ref.close()


def test_trajectory_against_ground_truth_on_a_constrained_scene(dev):
    """Scene S3 (the inside of a box room: every degree of freedom is observable) on the S1 camera path: the
    estimated camera motion must follow the path the frames were rendered from over 40 frames and 27 cm — position
    to a third of a voxel (30 mm voxels at 256^3; measured 4.4 mm, and 1.1-1.9 mm over three laps of the closed path at 512^3), orientation to a fraction of a degree.  (Scene
    S1, a wall and a sphere, leaves sliding along the wall to the sphere alone and drifts by millimetres per frame
    at every resolution: its trajectories are compared between implementations, not with the ground truth.)"""
    torch, pl = dev
    kf = pl.KinectFusion(synth.s1_params(256))
    c2w = lambda: np.linalg.inv(kf.world2camera()[..., 0].astype(np.float64))
    start = None
    worst_t, worst_r = 0.0, 0.0
    for k in range(41):
        assert kf.process_frame(upload(torch, synth.s3_frame(k))) == 1
        if k == 0:
            start = c2w()
            continue
        rel = np.linalg.inv(start) @ c2w()                      # camera k in the frame of camera 0
        gt = np.linalg.inv(synth.s1_pose(0)) @ synth.s1_pose(k)
        worst_t = max(worst_t, np.linalg.norm(rel[:3, 3] - gt[:3, 3]))
        cosang = (np.trace(rel[:3, :3].T @ gt[:3, :3]) - 1.0) / 2.0
        worst_r = max(worst_r, np.degrees(np.arccos(np.clip(cosang, -1.0, 1.0))))
    travelled = np.linalg.norm(synth.s1_pose(40)[:3, 3] - synth.s1_pose(0)[:3, 3])
    assert travelled > 0.2
    print(f"S3 at 256^3: worst position error {worst_t * 1e3:.2f} mm, worst orientation error {worst_r:.3f} deg over {travelled * 1e3:.0f} mm")
    assert worst_t <= 0.010, f"position error {worst_t * 1e3:.2f} mm over {travelled * 1e3:.0f} mm"
    assert worst_r <= 0.3, f"orientation error {worst_r:.3f} deg"
    kf.close()


def test_csfd_derivative_survives_the_whole_pipeline(dev):
    """A physical check of the complex-step derivative, end to end.  Seed the first camera pose's translation along
    the optical axis (world2camera(2,3) += i h): the first depth image is then fused h further along z, every
    later frame is tracked against that map, so every later world2camera(2,3) must carry d/dseed = 1 — through
    integrate (complex projective SDF), the running average of (value, grad), the raycast's complex zero crossing,
    27 complex sums per ICP iteration and the complex LLT solve, frame after frame.  Measured 1.00 -> 1.12 over
    20 frames of scene S3.  The same seed on a lateral translation, (0,3), must die out within two frames with
    nearest-pixel depth lookups (biInterpolate_threshold = 0: a projective SDF does not see a sideways shift of
    the camera except through the depth lookup) and survive with the bilinear lookup."""
    torch, pl = dev
    def run(seed, threshold, frames=21):
        kf = pl.KinectFusion(synth.s1_params(256, seed=seed, threshold=threshold))
        out = []
        for k in range(frames):
            assert kf.process_frame(upload(torch, synth.s3_frame(k))) == 1
            out.append(kf.world2camera()[:3, 3, 1] / np.float32(1e-7))
        kf.close()
        return np.array(out)
    along = run((2, 3), 0.0)
    assert along[0].tolist() == [0.0, 0.0, 1.0]
    assert np.all(along[1:, 2] >= 0.95) and np.all(along[1:, 2] <= 1.25), along[:, 2]
    lateral = run((0, 3), 0.0, frames=6)
    assert lateral[0, 0] == 1.0 and np.all(np.abs(lateral[3:]) <= 0.02), lateral
    lateral_bilinear = run((0, 3), 0.05, frames=6)
    assert np.all(lateral_bilinear[1:, 0] >= 0.9) and np.all(lateral_bilinear[1:, 0] <= 1.6), lateral_bilinear[:, 0]


@pytest.mark.parametrize("seed,threshold", [((2, 3), 0.0), ((0, 3), 0.05)], ids=["axial_seed_nearest", "lateral_seed_bilinear"])
def test_pipeline_scene_s3_against_live_oracle(dev, oracle, seed, threshold):
    """The box-room scene with the other CSFD seeds and the bilinear depth lookup (TsdfFusion.cu:128-158), side by
    side with the oracle pipeline: poses, their derivatives, the voxel counts and the fused volume."""
    torch, pl = dev
    from oracle.oracle import OracleKinFu, params_from_dict
    prm = synth.s1_params(96, seed=seed, threshold=threshold)
    kf = pl.KinectFusion(prm)
    ok_ = OracleKinFu(oracle, params_from_dict(prm))
    for k in range(4):
        d = synth.s3_frame(k)
        assert kf.process_frame(upload(torch, d)) == 1 and ok_.process_frame(d) == 1
        pose_close(kf.world2camera(), ok_.world2camera(), value_tol=1e-6 if k <= 1 else 2e-5, deriv_rel=1e-6 if k <= 1 else 1e-3)
        assert abs(kf.last_U() - ok_.last_U()) <= max(3, 1e-4 * ok_.last_U())
    assert abs(kf.world2camera()[seed[0], seed[1], 1] / np.float32(1e-7) - 1.0) <= 0.5     # the derivative is alive
    v, w, g = kf.volume()
    ov, ow, og = ok_.volume()
    assert mismatch_fraction(w, ow) <= 1e-4
    same = w == ow
    assert frac_bad(np.abs(v[same] - ov[same]) <= 1e-4) <= 1e-4
    assert frac_bad(np.abs(g[same] - og[same]) <= 1e-3 * np.abs(og).max()) <= 1e-4
    kf.close()


@pytest.mark.parametrize("threshold", [0.0, 0.02])
def test_pipeline_on_a_sensor_like_stream_against_live_oracle(dev, oracle, threshold):
    """The box-room stream as a sensor would deliver it — SURVEY 8(d)'s +-2 mm noise, 25 rectangular holes / out-of-range patches and 0.3 % invalid
    speckle per frame — through the whole pipeline (bilateral filter with holes, ICP, integrate with the round-5 box classes for boxes on the
    frustum's side and boxes that see invalid pixels, raycast from the sign map), side by side with the oracle pipeline: poses, their
    derivatives, voxel counts and the fused volume.  The fast paths are otherwise only compared with this build's own per-voxel walk."""
    torch, pl = dev
    from oracle.oracle import OracleKinFu, params_from_dict
    prm = synth.s1_params(128, seed=(2, 3), threshold=threshold)
    kf = pl.KinectFusion(prm)
    ok_ = OracleKinFu(oracle, params_from_dict(prm))
    rng = np.random.default_rng(0x5E75)
    for k in range(5):
        clean = synth.render_s3(synth.s1_pose(k)).astype(np.float64)
        noisy = np.clip(np.rint(clean + 2.0 * (rng.random(clean.shape) * 2 - 1)), 0, 65535).astype(np.uint16)
        d = synth.holed(noisy, rng, n_holes=25, speckle=0.003)
        assert kf.process_frame(upload(torch, d)) == 1 and ok_.process_frame(d) == 1
        # (noise puts many more pixels on the ICP's distance / angle gates and the bilateral filter's range than the clean scenes do: a pixel that
        # flips on a last-digit libm difference moves the 27 sums by one pixel's share — the derivative tolerance of the first frames is 1e-4
        # relative here where the clean scenes hold 1e-6; values hold 1e-6)
        pose_close(kf.world2camera(), ok_.world2camera(), value_tol=1e-6 if k <= 1 else 2e-5, deriv_rel=1e-4 if k <= 1 else 5e-3)
        if k > 0:
            il, wl = kf.icp_log(), ok_.icp_log()
            assert il.shape == wl.shape and np.all(np.abs(il[:, 54] - wl[:, 54]) <= np.maximum(3, 3e-4 * wl[:, 54]))     # inlier counts per iteration
        assert abs(kf.last_U() - ok_.last_U()) <= max(3, 1e-4 * ok_.last_U())
        assert abs(kf.last_hits() - ok_.last_hits()) <= max(3, 2e-4 * ok_.last_hits())
    v, w, g = kf.volume()
    ov, ow, og = ok_.volume()
    assert mismatch_fraction(w, ow) <= 1e-4
    same = w == ow
    assert frac_bad(np.abs(v[same] - ov[same]) <= 1e-4) <= 1e-4
    assert frac_bad(np.abs(g[same] - og[same]) <= 1e-3 * np.abs(og).max()) <= 1e-4
    kf.close()


def test_profiling_level_one_times_every_nth_integrate_launch_and_the_tail_clocks(dev):
    """Profiling level 1 attaches the integrate kernel's start / stop event pair to every profile_integrate_every-th frame only (its start event
    costs the launch call ~5 us of host time on the frame's critical path): the stage counter says how many launches were timed, their
    mean is a plausible kernel time, the voxel counters are complete all the same, and the results do not depend on the setting; level 2 times
    every frame.  xs_kf_tail_host_times / xs_kf_list_cover_counts report something for every tracked frame."""
    torch, pl = dev
    prm = synth.s1_params(128)
    frames = [upload(torch, synth.s1_frame(k)) for k in range(17)]
    runs = {every: pl.KinectFusion(dict(prm, profile_integrate_every=every)) for every in (1, 4)}
    for every, r in runs.items():
        assert r.process_frame(frames[0]) == 1
        r.set_profiling(1)
        r.reset_stage_times()
        for f in frames[1:]:
            assert r.process_frame(f) == 1
        torch.cuda.synchronize()
        st = r.stage_times()
        assert st["integrate"][1] == (16 if every == 1 else 4), (every, st["integrate"])
        assert 0.002 < st["integrate"][0] / st["integrate"][1] < 0.2
        th = r.tail_host_times()
        assert th["frames"] == 16 and 0.0 < th["integrate_launch_call"] < 200.0 and th["sums_seen_to_integrate_entered"] > 0.0
        held = r.list_cover_counts()
        assert held["neither"] + held["list_only"] + held["both"] == 16
        assert r.cumulative_counters()[0] > 16 * 1000
    a, b = runs[1], runs[4]
    assert np.array_equal(a.world2camera(), b.world2camera()) and a.cumulative_counters() == b.cumulative_counters()
    for u, v in zip(a.volume(), b.volume()):
        assert np.array_equal(u, v)
    b.set_profiling(2)
    b.reset_stage_times()
    for f in frames[:4]:
        assert b.process_frame(f) == 1
    torch.cuda.synchronize()
    assert b.stage_times()["integrate"][1] == 4
    for r in runs.values():
        r.close()
