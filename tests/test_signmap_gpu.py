"""GPU: the sign map (include/xslam_amd.h, csrc/xs_signmap.h) — the ray march started at the first step that can end it.
The reference has no such thing (RayCaster.cu:222-247 walks every step from t = 0.2 m); what is asserted here is therefore equality
with OUR OWN full march, which the oracle tests pin to the reference: the same crossing times, march lengths, vertices, normals and
hit counts, bit for bit, on fused scenes and on adversarial volumes, and that the map is a superset of the bricks holding negatives."""
import importlib

import numpy as np
import pytest

from helpers import intr_of, s1_transforms, synth, tranc_dist

pytestmark = pytest.mark.gpu
H, W = synth.HEIGHT, synth.WIDTH
HEAD = 64 + 320 * 4


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    return torch, importlib.import_module("x-slam_amd.capi")


def fuse(torch, capi, prm, res, frames, signmap=None, shift=3, bricks=True, depth_fn=None, threshold=0.0):
    X, Y, Z = res
    value = torch.zeros((Y * Z, X), dtype=torch.float32, device="cuda")
    weight = torch.zeros((Y * Z, X), dtype=torch.int32, device="cuda")
    grad = torch.zeros((Y * Z, X), dtype=torch.float32, device="cuda")
    scaled = torch.empty((H, W), dtype=torch.float32, device="cuda")
    dmax = torch.zeros(1, dtype=torch.float32, device="cuda")
    ws = torch.zeros(capi.integrate_workspace_bytes(res, Z), dtype=torch.uint8, device="cuda") if bricks else None
    if signmap is not None:
        capi.signmap_reset(signmap, res, shift, tranc_dist(prm))
    for k in frames:
        d = depth_fn(k) if depth_fn else synth.s1_frame(k)
        depth = torch.from_numpy(d.astype(np.int16)).cuda()
        T = s1_transforms(k, prm)
        capi.scale_depth_max(depth, W * 2, H, W, scaled, W * 4, dmax)
        capi.integrate_scaled_ex(scaled, W * 4, H, W, intr_of(prm), prm["max_integration_weight"], res, prm["tsdf_voxel_size"],
                                 T["Rv2c"], T["tv2c"], tranc_dist(prm), value, weight, grad, X * 4, 0, threshold=threshold,
                                 depth_max=dmax, workspace=ws, signmap=signmap)
    torch.cuda.synchronize()
    return value, weight, grad


def cast(torch, capi, prm, res, value, grad, T, signmap=None, shift=3, tranc=None):
    X = res[0]
    vm = torch.full((3 * H, W, 2), 5.0, dtype=torch.float32, device="cuda")
    nm = torch.full((3 * H, W, 2), 5.0, dtype=torch.float32, device="cuda")
    ws = torch.zeros(H * W, dtype=torch.float32, device="cuda")
    steps = torch.zeros(H * W, dtype=torch.int32, device="cuda")
    hits = torch.zeros(1, dtype=torch.int64, device="cuda")
    capi.raycast(intr_of(prm), T["Rc2v"], T["tc2v"], T["Rv2w"], T["tv2w"], tranc_dist(prm), res, prm["tsdf_voxel_size"], value, grad,
                 X * 4, vm, nm, W * 8, H, W, hits=hits, workspace=ws, steps=steps, signmap=signmap, signmap_shift=shift,
                 signmap_tranc_dist=tranc_dist(prm) if tranc is None else tranc)
    torch.cuda.synchronize()
    return [t.cpu().numpy() for t in (vm, nm, ws, steps, hits)]


def same_bits(a, b):
    for x, y in zip(a, b):
        assert np.array_equal(x.view(np.int32) if x.dtype != np.int64 else x, y.view(np.int32) if y.dtype != np.int64 else y)


def bricks_of(torch, signmap, res, shift):
    e = 1 << shift
    nx, ny, nz = [(r + e - 1) >> shift for r in res]
    nb = nx * ny * nz
    pad = (nb + 255) & ~255
    raw = signmap[HEAD:HEAD + nb].view(nz, ny, nx)
    dil = signmap[HEAD + pad:HEAD + pad + nb].view(nz, ny, nx)
    return raw, dil


def negatives_by_brick(torch, value, res, shift):
    X, Y, Z = res
    e = 1 << shift
    neg = (value.view(Z, Y, X) < 0).to(torch.float32)[None, None]
    return torch.nn.functional.max_pool3d(neg, e, ceil_mode=True)[0, 0] > 0


@pytest.mark.parametrize("n,shift", [(128, 3), (256, 3), (256, 4), (512, 3), (512, 4)])
def test_march_from_the_sign_map_gives_the_same_bits(dev, n, shift):
    torch, capi = dev
    prm = synth.s1_params(n)
    res = [n, n, n]
    sm = torch.zeros(capi.signmap_bytes(res, shift), dtype=torch.uint8, device="cuda")
    value, weight, grad = fuse(torch, capi, prm, res, [0, 1, 2], signmap=sm, shift=shift)
    plain = fuse(torch, capi, prm, res, [0, 1, 2])                # marking the map leaves the volume alone
    assert torch.equal(value, plain[0]) and torch.equal(weight, plain[1]) and torch.equal(grad, plain[2])
    raw, dil = bricks_of(torch, sm, res, shift)
    neg = negatives_by_brick(torch, value, res, shift)
    assert int(neg.sum()) > 50 and bool((raw[neg] == 1).all())    # a superset of the bricks that hold a negative voxel
    grown = torch.nn.functional.max_pool3d(raw.to(torch.float32)[None, None], 3, stride=1, padding=1)[0, 0] > 0
    assert bool((dil[grown] == 1).all())                          # ... grown by one brick
    assert bool((dil[0] == 1).all()) and bool((dil[:, :, -1] == 1).all())   # ... and the boundary shell
    for k in (2, 3, 9):
        T = s1_transforms(k, prm)
        full = cast(torch, capi, prm, res, value, grad, T)
        fast = cast(torch, capi, prm, res, value, grad, T, signmap=sm, shift=shift)
        same_bits(full, fast)
        assert int(full[4][0]) > 0.5 * H * W
    # the map rebuilt from the volume alone (a checkpoint was loaded): again the same bits, and no brick more than the marked ones
    sm2 = torch.zeros_like(sm)
    capi.signmap_rebuild(sm2, res, shift, tranc_dist(prm), value, n * 4)
    torch.cuda.synchronize()
    raw2, _ = bricks_of(torch, sm2, res, shift)
    assert bool(((raw2 == 1) == neg).all()) and bool((raw >= raw2).all())
    T = s1_transforms(3, prm)
    same_bits(cast(torch, capi, prm, res, value, grad, T), cast(torch, capi, prm, res, value, grad, T, signmap=sm2, shift=shift))


def test_sign_map_on_adversarial_volumes(dev):
    """Volumes no fusion would produce: isolated negative voxels in free space (a ray must stop at each), negative voxels at brick corners,
    a wholly negative volume, a camera outside the volume, a ragged size (the last brick overhangs)."""
    torch, capi = dev
    n = 96
    prm = synth.s1_params(n)
    rng = np.random.default_rng(11)
    cases = []
    for res in ([96, 96, 96], [90, 70, 83]):
        X, Y, Z = res
        v = np.where(rng.random((Z, Y, X)) < 2e-4, -0.5, rng.random((Z, Y, X))).astype(np.float32)
        cases.append((res, v, "isolated negatives"))
        v = np.full((Z, Y, X), 0.3, np.float32)
        v[7::8, 7::8, 7::8] = -0.2
        v[8::16, 8::16, 8::16] = -0.2
        cases.append((res, v, "brick corners"))
        cases.append((res, np.full((Z, Y, X), -0.4, np.float32), "all negative"))
        cases.append((res, np.zeros((Z, Y, X), np.float32), "never observed"))
    for res, v, name in cases:
        value = torch.from_numpy(v.reshape(res[1] * res[2], res[0])).cuda()
        grad = torch.zeros_like(value)
        for shift in (2, 3, 4):
            sm = torch.zeros(capi.signmap_bytes(res, shift), dtype=torch.uint8, device="cuda")
            capi.signmap_rebuild(sm, res, shift, tranc_dist(prm), value, res[0] * 4)
            for k in (0, 5):
                T = s1_transforms(k, prm)
                for out in (0.0, 4.0):     # 4.0: the camera 4 m further back, outside the volume
                    T2 = dict(T)
                    T2["tc2v"] = T["tc2v"].copy()
                    T2["tc2v"][2, 0] -= out   # Re z of the camera position in the volume frame
                    full = cast(torch, capi, prm, res, value, grad, T2)
                    fast = cast(torch, capi, prm, res, value, grad, T2, signmap=sm, shift=shift)
                    for x, y in zip(full, fast):
                        assert np.array_equal(x.view(np.int32) if x.dtype != np.int64 else x, y.view(np.int32) if y.dtype != np.int64 else y), (name, res, shift, k, out)


@pytest.mark.parametrize("thres_range", [3.0, 6.0, 10.0])
def test_sign_map_when_a_ray_leaves_a_negative_region(dev, thres_range):
    """The march's '- to +' event (RayCaster.cu:243) is decided by the PREVIOUS sample: a ray that starts inside, or passes through, a
    negative region and comes out into free space ends there without a vertex — also when a second surface lies further along.  With a
    long time step (thres_range 6 / 10: 0.8 * tranc_dist exceeds the map's sampling distance) the step that leaves the region has its
    own sample in a clear brick: the map must still have it evaluated.  Negative shells round the camera, thick and thin negative slabs
    across the view with a far surface behind them; every shift, against the full march, bit for bit."""
    torch, capi = dev
    n = 128
    prm = dict(synth.s1_params(n), thres_range=thres_range)
    res = [n, n, n]
    vs = prm["tsdf_voxel_size"]
    zz, yy, xx = np.mgrid[0:n, 0:n, 0:n].astype(np.float32)
    T = s1_transforms(0, prm)
    cam = np.asarray(T["tc2v"], np.float32).reshape(3, 2)[:, 0] / vs        # the camera in voxel units
    r = np.sqrt((xx - cam[0]) ** 2 + (yy - cam[1]) ** 2 + (zz - cam[2]) ** 2)
    cases = []
    for r_in, r_out in ((0.0, 9.0), (0.0, 14.5), (6.0, 11.0), (4.0, 5.5)):
        v = np.full((n, n, n), 0.4, np.float32)
        v[(r >= r_in) & (r < r_out)] = -0.3                                 # the camera inside (or just inside of) a negative shell
        v[zz > cam[2] + 30] = -0.6                                          # a surface further along: a + to - crossing the march must not reach
        cases.append((v, f"shell {r_in}-{r_out}"))
    for z0, thick in ((12, 1), (12, 3), (17, 9), (20, 2)):
        v = np.full((n, n, n), 0.4, np.float32)
        v[(zz >= cam[2] + z0) & (zz < cam[2] + z0 + thick) & (xx > cam[0] - 10)] = -0.3     # a slab across the right part of the view ...
        v[(zz >= cam[2] + z0 - 6) & (zz < cam[2] + z0) & (xx <= cam[0] - 10)] = -0.3        # ... and one nearer across the left
        v[zz > cam[2] + 40] = -0.6
        cases.append((v, f"slab at {z0} x {thick}"))
    ok_shift = misses = 0
    for v, name in cases:
        value = torch.from_numpy(v.reshape(n * n, n)).cuda()
        grad = torch.zeros_like(value)
        full = cast(torch, capi, prm, res, value, grad, T)
        misses += int(full[4][0]) < H * W                                     # (rays that end without a vertex)
        for shift in (2, 3, 4):
            nbytes = capi.signmap_bytes(res, shift)
            if nbytes == 0:
                continue
            sm = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
            try:
                capi.signmap_rebuild(sm, res, shift, tranc_dist(prm), value, n * 4)
            except RuntimeError:
                continue          # this spacing cannot serve this time step: the library refuses (and the orchestrator marches in full)
            try:
                fast = cast(torch, capi, prm, res, value, grad, T, signmap=sm, shift=shift)
            except RuntimeError:
                continue
            ok_shift += 1
            for x, y in zip(full, fast):
                assert np.array_equal(x.view(np.int32) if x.dtype != np.int64 else x, y.view(np.int32) if y.dtype != np.int64 else y), (name, shift)
    assert ok_shift >= len(cases) and misses >= 4


def test_sign_map_of_the_column_walk_and_the_bilinear_branch(dev):
    """The integrate paths other than the brick list mark the map too: the column walk (no workspace) and the bilinear depth lookup."""
    torch, capi = dev
    n = 128
    prm = synth.s1_params(n)
    res = [n, n, n]
    for bricks, threshold in ((False, 0.0), (True, 0.05), (False, 0.05)):
        sm = torch.zeros(capi.signmap_bytes(res, 3), dtype=torch.uint8, device="cuda")
        value, weight, grad = fuse(torch, capi, prm, res, [0, 1], signmap=sm, bricks=bricks, threshold=threshold)
        raw, dil = bricks_of(torch, sm, res, 3)
        neg = negatives_by_brick(torch, value, res, 3)
        assert int(neg.sum()) > 50 and bool((raw[neg] == 1).all())
        T = s1_transforms(2, prm)
        same_bits(cast(torch, capi, prm, res, value, grad, T), cast(torch, capi, prm, res, value, grad, T, signmap=sm))


def test_sign_map_misuse_is_refused(dev):
    torch, capi = dev
    n = 64
    prm = synth.s1_params(n)
    res = [n, n, n]
    assert capi.signmap_bytes(res, 1) == 0 and capi.signmap_bytes(res, 7) == 0 and capi.signmap_bytes([0, 4, 4], 3) == 0
    sm = torch.zeros(capi.signmap_bytes(res, 3), dtype=torch.uint8, device="cuda")
    capi.signmap_reset(sm, res, 3, tranc_dist(prm))
    value = torch.zeros((n * n, n), dtype=torch.float32, device="cuda")
    T = s1_transforms(0, prm)
    with pytest.raises(RuntimeError, match="truncation"):      # the time table belongs to another truncation distance
        cast(torch, capi, prm, res, value, value, T, signmap=sm, tranc=2 * tranc_dist(prm))
    with pytest.raises(RuntimeError, match="time table"):      # more march steps than the table holds
        capi.signmap_reset(sm, res, 3, 1e-4)
    # slab launches mark the bricks of their own planes (a rank of a sharded volume)
    prm2 = synth.s1_params(96)
    res2 = [96, 96, 96]
    marks = []
    for bounds in ([0, 96], [0, 40, 96]):
        sm2 = torch.zeros(capi.signmap_bytes(res2, 3), dtype=torch.uint8, device="cuda")
        capi.signmap_reset(sm2, res2, 3, tranc_dist(prm2))
        v = torch.zeros((96 * 96, 96), dtype=torch.float32, device="cuda")
        w = torch.zeros((96 * 96, 96), dtype=torch.int32, device="cuda")
        g = torch.zeros_like(v)
        scaled2 = torch.empty((H, W), dtype=torch.float32, device="cuda")
        depth = torch.from_numpy(synth.s1_frame(0).astype(np.int16)).cuda()
        capi.scale_depth(depth, W * 2, H, W, scaled2, W * 4)
        T2 = s1_transforms(0, prm2)
        for z0, z1 in zip(bounds[:-1], bounds[1:]):
            off = z0 * 96
            capi.integrate_scaled_ex(scaled2, W * 4, H, W, intr_of(prm2), 100, res2, prm2["tsdf_voxel_size"], T2["Rv2c"], T2["tv2c"], tranc_dist(prm2),
                                     v[off:], w[off:], g[off:], 96 * 4, 0, z0=z0, z1=z1, signmap=sm2)
        torch.cuda.synchronize()
        raw2 = bricks_of(torch, sm2, res2, 3)[0]
        neg2 = negatives_by_brick(torch, v, res2, 3)
        assert int(neg2.sum()) > 20 and bool((raw2[neg2] == 1).all())     # a superset either way (the column walk marks a column's whole span)
        marks.append(v.clone())
    assert torch.equal(marks[0], marks[1])


def test_randomized_poses_march_from_the_sign_map_like_the_full_march(dev, oracle):
    """The sign-map march rests on a geometric argument (samples every dt along the ray of a wave's pixel tile cannot miss a brick that holds a
    negative voxel: xs_raycast.hip sign_map_spacing) — so, beside the hand-picked poses above, RANDOM ones (round 6): a 256^3 map of the box room
    (walls in every orientation, fused with the map marked), 360 cameras (60 per sensor and brick shift) drawn within the march's reach of a random surface point (inside the
    volume or not, from in front of the surface or from behind it), looking at it, rolled by up to +-pi, intrinsics of three sensors scaled to the 640 x 480 maps (fx 481 / 585 / 350: the pixel tile's half diagonal changes the sample
    spacing), both usable brick shifts: vertex map, normal map, crossing times, per-ray step counts and hit count against the march that evaluates
    every step, bit for bit; the full march of 18 of the views against the CPU oracle (vertices: test_raycast's rule; normals to 2 ulp)."""
    torch, capi = dev
    n = 256
    prm = synth.s1_params(n)
    res = [n, n, n]
    rng = np.random.default_rng(0x51617)
    td = tranc_dist(prm)
    for fx, fy in ((synth.FX, synth.FY), (585.0, -585.0), (350.0, -350.0)):
        intr = np.array([fx, fy, synth.CX, synth.CY], np.float32)
        finest = capi.raycast_signmap_shift(intr, prm["tsdf_voxel_size"], td)
        assert finest
        for shift in sorted({finest, min(finest + 1, 6)}):
            sm = torch.zeros(capi.signmap_bytes(res, shift), dtype=torch.uint8, device="cuda")
            value, weight, grad = fuse(torch, capi, prm, res, [0, 3, 6, 9], signmap=sm, shift=shift, depth_fn=synth.s3_frame)
            vs = prm["tsdf_voxel_size"]
            v_host, g_host = value.cpu().numpy().reshape(-1), grad.cpu().numpy().reshape(-1)
            neg = torch.nonzero(value.view(n, n, n) < 0).cpu().numpy()      # (z, y, x) of the voxels behind a surface: what a camera can look at
            assert len(neg) > 1000
            hit_total = 0
            for trial in range(60):
                zyx = neg[rng.integers(len(neg))]
                target = (zyx[::-1] + 0.5) * vs
                d = rng.normal(size=3); d /= np.linalg.norm(d) + 1e-12
                eye = target + d * rng.uniform(0.4, 4.5)                    # anywhere within the march's 5 m of it: inside the volume or not
                z = target - eye
                z /= np.linalg.norm(z) + 1e-12
                x = np.cross(rng.normal(size=3), z); x /= np.linalg.norm(x) + 1e-12
                y = np.cross(z, x)
                roll = rng.uniform(-np.pi, np.pi)
                x, y = np.cos(roll) * x + np.sin(roll) * y, -np.sin(roll) * x + np.cos(roll) * y
                Rc2v = np.zeros((3, 3, 2), np.float32); Rc2v[..., 0] = np.stack([x, y, z], 1); Rc2v[..., 1] = rng.normal(size=(3, 3)) * 1e-7
                tc2v = np.zeros((3, 2), np.float32); tc2v[:, 0] = eye; tc2v[:, 1] = rng.normal(size=3) * 1e-7
                I = np.zeros((3, 3, 2), np.float32); I[[0, 1, 2], [0, 1, 2], 0] = 1
                T = {"Rc2v": Rc2v, "tc2v": tc2v, "Rv2w": I, "tv2w": np.zeros((3, 2), np.float32)}
                p2 = dict(prm, fx=float(fx), fy=float(fy))
                full = cast(torch, capi, p2, res, value, grad, T)
                fast = cast(torch, capi, p2, res, value, grad, T, signmap=sm, shift=shift)
                same_bits(full, fast)
                hit_total += int(full[4][0])
                if trial < 3:   # ... and the full march itself against the CPU oracle on these random views (18 of them): test_raycast's tolerances
                    from test_kernels_gpu import cmap_close
                    ov, on, ohits = oracle.raycast(intr, Rc2v, tc2v, I, T["tv2w"], td, res, vs, v_host, g_host, H, W)
                    assert abs(int(full[4][0]) - ohits) <= max(3, 1e-4 * ohits), (fx, shift, trial, int(full[4][0]), ohits)
                    if ohits > 1000:
                        # vertices: test_raycast's rule (measured on these views: identical bits in all 18).  Normals: the squared length of a small
                        # gradient can leave the CSFD cone in which sqrt(z) has its libm-free form, and device and host libm then differ
                        # by an ulp — measured: up to 8 % of a view's normals differ, by <= 2 ulp of 1 in the value and <= 1.1e-7 in the derivative part
                        cmap_close(full[0], ov, H, budget=2e-4)
                        gg, ww = full[1].reshape(3, H, W, 2), on.reshape(3, H, W, 2)
                        gn_, wn_ = np.isnan(gg[0, ..., 0]), np.isnan(ww[0, ..., 0])
                        assert (gn_ != wn_).mean() <= 2e-4
                        okk = ~gn_ & ~wn_
                        for q in range(3):
                            assert np.abs(gg[q][okk][:, 0] - ww[q][okk][:, 0]).max() <= 5e-7
                            assert np.abs(gg[q][okk][:, 1] - ww[q][okk][:, 1]).max() <= 2e-7 + 1e-5 * np.abs(ww[q][okk][:, 1]).max()
            assert hit_total > 60 * 0.05 * H * W, (fx, shift, hit_total)
