"""Synthetic 640x480 depth streams: no dataset ships with the
reference (its ICL-NUIM loader, XKinectFusion/src/Dataset.cpp:69-124, reads files
that are not in the tree), so the workloads are analytic scenes rendered on the
host.  Depth is u16 millimetres, the convention the reference's loader hands to
``ProcessFrame`` (Dataset.cpp:3-11; valid range 200..5000, TsdfFusion.cu:76-81).

Scene S1 ("ICL-like", tracks):   plane z = 2.5 m + sphere centre (0.2, 0.1, 1.8) r 0.4,
                                 world = camera-0 frame, slow sinusoidal trajectory.
Scene S2 ("frustum-filling"):    static camera on the z = 0 face of a 5.12 m cube
                                 looking at a plane 4.5 m away (HBM-bound integrate).
Scene S3 ("room"):               the inside of a box, walls z = 2.5, x = +-1.0, y = -0.8 / +0.9, on
                                 the S1 trajectory: three orthogonal walls are in view from every
                                 pose, all six degrees of freedom are constrained, so the estimated
                                 trajectory can be held against the ground truth.
"""
import math

import numpy as np

# ICL_traj2.yaml:36-41
FX, FY, CX, CY = 481.20, -480.00, 319.50, 239.50
WIDTH, HEIGHT = 640, 480


def _rot_x(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]], np.float64)


def _rot_y(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], np.float64)


def s1_pose(k, T=300):
    """Camera-to-world pose of frame k: t = (0.30 sin th, 0.15 sin 2th, 0.10 (1 - cos th)),
    R = Ry(0.10 sin th) Rx(0.05 sin 2th), th = 2 pi k / T."""
    th = 2.0 * math.pi * k / T
    R = _rot_y(0.10 * math.sin(th)) @ _rot_x(0.05 * math.sin(2 * th))
    t = np.array([0.30 * math.sin(th), 0.15 * math.sin(2 * th), 0.10 * (1.0 - math.cos(th))], np.float64)
    M = np.eye(4)
    M[:3, :3] = R
    M[:3, 3] = t
    return M


def _hash_noise(frame, n, seed=0xC5FD):
    """Counter-based integer hash (splitmix64 finaliser) -> uniform in [-1, 1)."""
    idx = np.arange(n, dtype=np.uint64) + np.uint64((int(frame) * 0x9E3779B97F4A7C15 + int(seed)) & 0xFFFFFFFFFFFFFFFF)   # (wraps mod 2^64, as the scalar product did)
    z = idx
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) / float(1 << 53) * 2.0 - 1.0


def render_s1(c2w, width=WIDTH, height=HEIGHT, fx=FX, fy=FY, cx=CX, cy=CY, noise_mm=0.0, frame=0):
    """Depth (u16 mm, z-component in the camera frame) of plane + sphere seen from c2w."""
    u = np.arange(width, dtype=np.float64)
    v = np.arange(height, dtype=np.float64)
    dx = (u[None, :] - cx) / fx
    dy = (v[:, None] - cy) / fy
    d = np.stack([np.broadcast_to(dx, (height, width)), np.broadcast_to(dy, (height, width)),
                  np.ones((height, width))], axis=-1)
    R, o = c2w[:3, :3], c2w[:3, 3]
    D = d @ R.T  # world direction; camera-frame z of o + s*D is s
    best = np.full((height, width), np.inf)
    # plane z_w = 2.5
    with np.errstate(divide="ignore", invalid="ignore"):
        s = (2.5 - o[2]) / D[..., 2]
    s = np.where((D[..., 2] > 0) & (s > 0), s, np.inf)
    best = np.minimum(best, s)
    # sphere
    c = np.array([0.2, 0.1, 1.8])
    r = 0.4
    oc = o - c
    a = np.sum(D * D, axis=-1)
    b = 2.0 * (D @ oc)
    cc = float(oc @ oc) - r * r
    disc = b * b - 4 * a * cc
    with np.errstate(invalid="ignore"):
        sq = np.sqrt(np.where(disc >= 0, disc, 0.0))
        s0 = (-b - sq) / (2 * a)
    s0 = np.where((disc >= 0) & (s0 > 0), s0, np.inf)
    best = np.minimum(best, s0)
    mm = 1000.0 * best
    if noise_mm > 0:
        mm = mm + noise_mm * _hash_noise(frame, width * height).reshape(height, width)
    mm = np.where(np.isfinite(mm), mm, 0.0)
    mm = np.clip(np.rint(mm), 0, 65535)
    return mm.astype(np.uint16)


def render_s3(c2w, width=WIDTH, height=HEIGHT, fx=FX, fy=FY, cx=CX, cy=CY):
    """Depth (u16 mm) of the room z = 2.5, x = +-1.0, y = -0.8 / +0.9 seen from inside, at c2w."""
    u = np.arange(width, dtype=np.float64)
    v = np.arange(height, dtype=np.float64)
    d = np.stack([np.broadcast_to((u[None, :] - cx) / fx, (height, width)), np.broadcast_to((v[:, None] - cy) / fy, (height, width)),
                  np.ones((height, width))], axis=-1)
    R, o = c2w[:3, :3], c2w[:3, 3]
    D = d @ R.T
    best = np.full((height, width), np.inf)
    for axis, level in ((2, 2.5), (0, 1.0), (0, -1.0), (1, -0.8), (1, 0.9)):
        with np.errstate(divide="ignore", invalid="ignore"):
            s = (level - o[axis]) / D[..., axis]
        best = np.minimum(best, np.where(np.isfinite(s) & (s > 0), s, np.inf))
    mm = np.where(np.isfinite(best), 1000.0 * best, 0.0)
    return np.clip(np.rint(mm), 0, 65535).astype(np.uint16)


def s3_frame(k, T=300, **kw):
    return render_s3(s1_pose(k, T), **kw)


# 7-Scenes (BASELINE configs 3 and 5): Kinect intrinsics 585 / 585 / 320 / 240, 640 x 480 — the dataset itself is not available
SEVEN_SCENES = dict(fx=585.0, fy=585.0, cx=320.0, cy=240.0)


def s1_frame(k, T=300, noise_mm=0.0, **kw):
    return render_s1(s1_pose(k, T), noise_mm=noise_mm, frame=k, **kw)


def s1_params(n=256, seed=(0, 3), seed_h=1e-7, threshold=0.0):
    """XKinectFusion parameters for scene S1 at volume edge n (7.68 m cube)."""
    return dict(
        tsdf_size_x=n, tsdf_size_y=n, tsdf_size_z=n, tsdf_voxel_size=7.68 / n, max_integration_weight=100,
        thres_range=3.0, init_x=3.2, init_y=3.2, init_z=3.2, r_x=0.0, r_y=0.0, r_z=0.0,
        depth_width=WIDTH, depth_height=HEIGHT, fx=FX, fy=FY, cx=CX, cy=CY, num_levels=3,
        distThres=0.10, angleThres=15.0, biInterpolate_threshold=threshold, trunc_logistic_k=0.0,
        flag_use_gtPose=False, frame_step=1,
        csfd_seed_row=seed[0] if seed else -1, csfd_seed_col=seed[1] if seed else -1, csfd_seed_h=seed_h,
    )


def render_s2(width=WIDTH, height=HEIGHT, noise_mm=0.0, frame=0):
    """Static view of a plane 4.5 m in front of the camera (z-depth constant); noise_mm: SURVEY 8(d)'s uniform sensor noise."""
    mm = np.full((height, width), 4500.0)
    if noise_mm > 0:
        mm = mm + noise_mm * _hash_noise(frame, width * height).reshape(height, width)
    return np.clip(np.rint(mm), 0, 65535).astype(np.uint16)


def holed(d, rng, n_holes=40, speckle=0.002):
    """A depth frame with rectangular holes (0), out-of-range pixels (150 / 6000 mm: outside the 200..5000 gate of TsdfFusion.cu:76-81)
    and speckle (single invalid pixels): what a real sensor's frame looks like to the integrate kernel's box classification."""
    d = d.copy()
    h, w = d.shape
    for _ in range(n_holes):
        y, x = rng.integers(0, h - 30), rng.integers(0, w - 40)
        d[y:y + rng.integers(1, 30), x:x + rng.integers(1, 40)] = rng.choice([0, 150, 6000])
    m = rng.random(d.shape) < speckle
    d[m] = 0
    return d


def s2_params(n=512):
    """Frustum-filling placement: 5.12 m cube, camera at (2.56, 2.56, 0.05) in the
    volume frame looking +z; ground-truth pose, so only integrate + raycast run."""
    p = s1_params(n, seed=(0, 3))
    p.update(tsdf_voxel_size=5.12 / n, init_x=2.56, init_y=2.56, init_z=0.05, flag_use_gtPose=True)
    return p


# ---- kernel inputs for a frame of the synthetic scenes (what the orchestrator hands to the launchers) ----------------
def cmat(real, imag=None):
    """float64 real (+ optional imag) matrix -> float32 array with trailing (re, im)."""
    real = np.asarray(real, np.float64)
    out = np.zeros(real.shape + (2,), np.float32)
    out[..., 0] = real
    if imag is not None:
        out[..., 1] = imag
    return out


def s1_transforms(k, prm, seed=(0, 3), h=1e-7):
    """Complex transforms the orchestrator hands to the kernels for frame k of the S1 / S3 camera path,
    computed in float64 and rounded once (inputs only — every implementation receives the same bits).
    The CSFD seed i*h sits on world2camera(seed) as in KinectFusionReconstruction.cpp:22."""
    c2w = s1_pose(k).astype(np.complex128)
    w2c = np.linalg.inv(c2w)
    if seed is not None:
        w2c[seed] += 1j * h
    c2w = np.linalg.inv(w2c)
    w2v = np.eye(4, dtype=np.complex128)
    w2v[:3, 3] = [prm["init_x"], prm["init_y"], prm["init_z"]]
    c2v = w2v @ c2w
    v2c = np.linalg.inv(c2v)
    v2w = np.linalg.inv(w2v)
    f = lambda m: cmat(m.real, m.imag)
    return dict(Rv2c=f(v2c[:3, :3]), tv2c=f(v2c[:3, 3]), Rc2v=f(c2v[:3, :3]), tc2v=f(c2v[:3, 3]),
                Rv2w=f(v2w[:3, :3]), tv2w=f(v2w[:3, 3]), Rc2w=f(c2w[:3, :3]), tc2w=f(c2w[:3, 3]), c2w=c2w, w2c=w2c)


def tranc_dist(prm):
    """TsdfVolume::setTsdfTruncDist (TsdfVolume.cpp:34-38): max(voxel * thres_range, 2.1 * voxel), in float."""
    vs = np.float32(prm["tsdf_voxel_size"])
    return float(max(np.float32(vs * np.float32(prm["thres_range"])), np.float32(np.float32(2.1) * vs)))


def intr_of(prm, level=0):
    d = np.float32(1 << level)
    return np.array([np.float32(prm["fx"]) / d, np.float32(prm["fy"]) / d, np.float32(prm["cx"]) / d, np.float32(prm["cy"]) / d],
                    np.float32)
