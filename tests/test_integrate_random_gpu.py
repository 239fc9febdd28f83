"""Randomized differential test of the integrate kernel's box classes (FREE / EMPTY / plane by plane, EDGE, SPECKLE: xs_tsdf.hip
classify_box, integrate_edge_column, valid_slot_request) against the per-voxel walk over every listed brick (XS_INTEGRATE_NO_TILES),
which the oracle tests pin (tests/test_integrate_gpu.py).  The classes' shortcuts rest on numeric margins — a reciprocal-based
projection trusted outside `near_margin` of a pixel boundary, an in-image window trusted outside 1/32 px — so hand-picked scenes
are not enough: here the pose, the sensor, the depth image and its invalid pixels are drawn at random.

TRIALS = 600 seeded trials (200 per sensor: 640x480, 1280x960, 1920x1080), each: a random volume (96^3 .. 160^3, 1.5 .. 8 m), a
random camera (inside the volume half of the time — voxels down to c = XS_EDGE_CMIN and behind the camera — or outside it, looking
at a random point of it, rolled by up to +-pi), a random piecewise-planar depth image with +-2 mm noise and 0 .. 10 % invalid
pixels (speckle + rectangles of 0 / out-of-range depths), first-order CSFD imaginary parts on the pose, nearest-pixel and bilinear
depth (both thresholds), two launches per volume (the second on the first one's state).  Volume (value, weight, grad bits) and the
count of written voxels must equal the walk's for (a) classes decided with the launch's own pose and (b) list and classes decided
AHEAD for a nearby pose with slack 2 (xs_integrate_classify_ex + xs_integrate_list_covers, as the orchestrator does).  On 180 of the trials
(60 per sensor) the walk itself is held against the CPU ORACLE on the same random input (bit-exact up to the fixed-scene tests' flip budget):
the random poses, sensors and invalid pixels reach the oracle in one link, the classes in two."""
import numpy as np
import pytest

from helpers import synth

pytestmark = pytest.mark.gpu
TRIALS = 600
SENSORS = ((480, 640), (960, 1280), (1080, 1920))
NO_TILES, COUNT_CLASSES, LIST_IS_READY, HEADER_IS_CLEAR, RECLASSIFY_BOXES = 32, 64, 4, 1, 128


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    return torch, __import__("importlib").import_module("x-slam_amd.capi")


def intrinsics(hh, ww):
    sx, sy = ww / synth.WIDTH, hh / synth.HEIGHT
    return np.array([synth.FX * sx, synth.FY * sy, (ww - 1) / 2.0, (hh - 1) / 2.0], np.float32)


def random_depth(rng, hh, ww, invalid_share):
    """Millimetres, u16: a tilted plane, up to six rectangles at other depths, +-2 mm noise; then invalid pixels — half of the share as
    speckle, half as rectangles of 0 / 150 / 6000 mm (outside the 200 .. 5000 gate of TsdfFusion.cu:76-81)."""
    yy, xx = np.mgrid[0:hh, 0:ww].astype(np.float32)
    d0 = rng.uniform(0.45, 4.7)
    mm = 1000.0 * (d0 + rng.uniform(-0.4, 0.4) * (xx / ww - 0.5) + rng.uniform(-0.4, 0.4) * (yy / hh - 0.5))
    for _ in range(rng.integers(0, 7)):
        y, x = rng.integers(0, hh - 8), rng.integers(0, ww - 8)
        mm[y:y + rng.integers(4, hh // 2), x:x + rng.integers(4, ww // 2)] = 1000.0 * rng.uniform(0.3, 4.9)
    mm += 2.0 * (rng.random((hh, ww), dtype=np.float32) * 2 - 1)
    d = np.clip(np.rint(mm), 0, 65535).astype(np.uint16)
    if invalid_share > 0:
        d[rng.random((hh, ww), dtype=np.float32) < 0.5 * invalid_share] = 0
        area, want = 0, 0.5 * invalid_share * hh * ww
        while area < want:
            h, w = rng.integers(1, max(2, hh // 8)), rng.integers(1, max(2, ww // 8))
            y, x = rng.integers(0, hh - h), rng.integers(0, ww - w)
            d[y:y + h, x:x + w] = rng.choice([0, 150, 6000])
            area += h * w
    return d


def random_pose(rng, extent, h=1e-7):
    """volume-to-camera (R [3, 3, 2], t [3, 2]) of a camera inside the volume (any direction) or outside it (looking at a random point of it),
    rolled by up to +-pi about its axis; imaginary parts ~h."""
    inside = rng.random() < 0.5
    eye = rng.uniform(0.05, 0.95, 3) * extent if inside else extent * (0.5 + rng.choice([-1, 1], 3) * rng.uniform(0.55, 1.6, 3) * (rng.random(3) < 0.7))
    target = rng.uniform(0.1, 0.9, 3) * extent
    if inside and rng.random() < 0.3:
        target = eye + rng.normal(size=3)
    z = target - eye
    z /= np.linalg.norm(z) + 1e-12
    up = rng.normal(size=3)
    x = np.cross(up, z); x /= np.linalg.norm(x) + 1e-12
    y = np.cross(z, x)
    roll = rng.uniform(-np.pi, np.pi)
    x, y = np.cos(roll) * x + np.sin(roll) * y, -np.sin(roll) * x + np.cos(roll) * y
    Rc2v = np.stack([x, y, z], 1)                       # camera axes in volume coordinates
    Rv2c = Rc2v.T
    R = np.zeros((3, 3, 2), np.float32); R[..., 0] = Rv2c; R[..., 1] = rng.normal(size=(3, 3)) * h
    t = np.zeros((3, 2), np.float32); t[:, 0] = -Rv2c @ eye; t[:, 1] = rng.normal(size=3) * h
    return R, t, inside


def nearby(rng, R, t, vs):
    """A pose the ICP's last update could have come from: a rotation of <= 0.4 mrad and a shift of <= 0.3 voxels away."""
    w = rng.normal(size=3) * 2e-4
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    R2, t2 = R.copy(), t.copy()
    R2[..., 0] = (np.eye(3) + K) @ R[..., 0].astype(np.float64)
    t2[:, 0] = (np.eye(3) + K) @ t[:, 0].astype(np.float64) + rng.uniform(-0.3, 0.3, 3) * vs
    return R2, t2


def run_trial(torch, capi, rng, hh, ww, threshold, h=1e-7, n=None, extent=None, oracle=None):
    n = int(n or rng.choice([96, 128, 160]))
    extent = float(extent or rng.uniform(1.5, 8.0))
    vs = extent / n
    trunc = float(rng.uniform(2.1, 5.0)) * vs
    res = [n, n, n]
    k4 = intrinsics(hh, ww)
    invalid_share = 0.0 if rng.random() < 0.15 else float(rng.uniform(0.0, 0.10))
    depth_np = random_depth(rng, hh, ww, invalid_share)
    depth = torch.from_numpy(depth_np.view(np.int16)).cuda()
    scaled = torch.empty((hh, ww), dtype=torch.float32, device="cuda")
    dmax = torch.zeros(1, dtype=torch.float32, device="cuda")
    tiles = torch.zeros(capi.depth_tiles_bytes(hh, ww), dtype=torch.uint8, device="cuda")
    capi.scale_depth_tiles(depth, ww * 2, hh, ww, scaled, ww * 4, dmax, tiles)
    ws = torch.zeros(capi.integrate_workspace_bytes(res), dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    poses = [random_pose(rng, extent, h)[:2], None]
    poses[1] = nearby(rng, *poses[0], vs) if rng.random() < 0.5 else random_pose(rng, extent, h)[:2]   # second launch: the same view again, or another

    def volume():
        v = torch.empty((n * n, n), dtype=torch.float32, device="cuda"); w = torch.empty((n * n, n), dtype=torch.int32, device="cuda")
        g = torch.empty((n * n, n), dtype=torch.float32, device="cuda")
        capi.init_volume(v, w, g, n * 4, res)
        return v, w, g

    def launch(vol, R, t, opts):
        cnt.zero_()
        capi.integrate_scaled_ex2(scaled, ww * 4, hh, ww, k4, 100, res, vs, R, t, trunc, vol[0], vol[1], vol[2], n * 4, opts, threshold=threshold,
                                  updated=cnt, depth_max=dmax, workspace=ws)
        return int(cnt.item())

    ref = volume()
    U = [launch(ref, R, t, capi.integrate_opts(flags=NO_TILES)) for R, t in poses]
    if oracle is not None:
        # the walk itself against the CPU oracle on the same random input (the walk still rests on the brick list and the column clip, which are
        # conservative tests of their own): bit-exact up to the flip budget of the fixed-scene tests (2e-5 of the voxels), counts within 1e-4
        ov, ow, og = oracle.new_volume(res)
        ds = oracle.scale_depth(depth_np)
        UO = [oracle.integrate(ds, ov, ow, og, res, trunc, 100, R, t, k4, vs, threshold) for R, t in poses]
        gv, gw, gg = (x.cpu().numpy().reshape(-1) for x in ref)
        for a, b, what in ((gw, ow, "weight"), (gv, ov, "value"), (gg, og, "grad")):
            bad = float((a != b).mean())
            assert bad <= 2e-5, ("oracle", what, bad)
        for a, b in zip(U, UO):
            assert abs(a - b) <= max(2, 1e-4 * b), ("oracle count", a, b)
    stats = np.zeros(8, np.int64)
    covered = 0
    for ahead in (False, True):
        vol = volume()
        for i, (R, t) in enumerate(poses):
            flags = COUNT_CLASSES
            if ahead:
                Rl, tl = nearby(rng, R, t, vs)
                capi.integrate_classify_ex(hh, ww, k4, res, vs, Rl, tl, trunc, ws, capi.integrate_opts(flags=COUNT_CLASSES, depth_tiles=tiles), slack_scale=2.0,
                                           depth_max=dmax)
                covers = capi.integrate_list_covers(hh, ww, k4, res, vs, Rl, tl, 2.0, R, t)
                if covers:
                    flags |= LIST_IS_READY | HEADER_IS_CLEAR | (0 if covers & 2 else RECLASSIFY_BOXES)
                    covered += 1
                else:
                    capi.integrate_workspace_clear(ws)
            got = launch(vol, R, t, capi.integrate_opts(flags=flags, depth_tiles=tiles))
            assert got == U[i], (ahead, i, got, U[i])
            stats += ws[192:224].view(torch.int32).cpu().numpy()
        for a, b, what in zip(ref, vol, ("value", "weight", "grad")):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (ahead, what, int((a.view(torch.int32) != b.view(torch.int32)).sum().item()))
    return sum(U), stats, covered


def test_randomized_classes_against_the_walk(dev, oracle):
    torch, capi = dev
    rng = np.random.default_rng(0xC1A55E5)
    totals = {s: np.zeros(8, np.int64) for s in SENSORS}
    written, covered = 0, 0
    for trial in range(TRIALS):
        hh, ww = SENSORS[trial % 3]
        threshold = (0.0, 0.02)[(trial // 3) % 2]
        try:
            u, stats, cov = run_trial(torch, capi, rng, hh, ww, threshold, oracle=oracle if trial % 10 < 3 else None)   # (every sensor: trials 0, 1, 2 of ten)
        except AssertionError as e:
            raise AssertionError(f"trial {trial} ({ww}x{hh}, threshold {threshold}): {e}") from e
        totals[(hh, ww)] += stats
        written += u
        covered += cov
    # every class was taken on every sensor: whole boxes free / empty / walked, EDGE planes, SPECKLE planes
    for s, t in totals.items():
        assert t[0] > 1000 and t[1] > 1000 and t[2] > 1000 and t[6] > 1000 and t[7] > 1000, (s, t)
    assert written > 50_000_000 and covered > TRIALS, (written, covered)


@pytest.mark.parametrize("case", ["wide_4096x2048", "too_wide_6000x2400", "seed_1e-3"])
def test_streamed_classes_stop_where_their_margins_are_not_proven(dev, case):
    """stream_margins (xs_tsdf.hip): the nearest-pixel margin is 8 ulp of E = max(cols + |cx|, rows + |cy|) — a 4096 x 2048 sensor (the
    largest the tile room takes) streams its EDGE / SPECKLE planes with 1 / 256 px and stays bit-identical to the walk; with E >= 8192,
    or with imaginary pose parts of 1e-3 (whose products reach the real part of the exact projection), no box is classed EDGE or
    SPECKLE — such boxes walk — and the volume is the walk's all the same."""
    torch, capi = dev
    rng = np.random.default_rng({"wide_4096x2048": 1, "too_wide_6000x2400": 2, "seed_1e-3": 3}[case])
    hh, ww = {"wide_4096x2048": (2048, 4096), "too_wide_6000x2400": (2400, 6000), "seed_1e-3": (480, 640)}[case]
    stats = np.zeros(8, np.int64)
    for trial in range(4):
        _, s, _ = run_trial(torch, capi, rng, hh, ww, (0.0, 0.02)[trial % 2], h=1e-3 if case == "seed_1e-3" else 1e-7, n=128)
        stats += s
    if case == "wide_4096x2048":
        assert stats[6] > 100 and stats[7] > 100, stats
    else:
        assert stats[6] == 0 and stats[7] == 0 and stats[2] > 100, stats
