"""GPU parity for the remaining hot-path kernels (map prep, raycast, ICP normal equations,
dual-complex Hessian / loss, DeviceArray scalar tables), each through the C ABI against the
CPU oracle on the same seeded inputs."""
import importlib

import numpy as np
import pytest

from conftest import load_golden, ulp_diff
from helpers import intr_of, mismatch_fraction, s1_transforms, synth, tranc_dist

pytestmark = pytest.mark.gpu
H, W = synth.HEIGHT, synth.WIDTH


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available()
    return torch, importlib.import_module("x-slam_amd.capi")


def to_dev(torch, a):
    a = np.ascontiguousarray(a)
    if a.dtype == np.uint16:
        a = a.view(np.int16)
    return torch.from_numpy(a).cuda()


def cmap_close(got, want, planes_rows, rel=1e-6, budget=2e-5, deriv_scale=None):
    """Maps with the NaN sentinel in the x plane: sentinel sets must agree (up to a flip
    budget); y/z planes are compared only where x is valid (they hold stale data elsewhere)."""
    got = got.reshape(3, planes_rows, -1, 2)
    want = want.reshape(3, planes_rows, -1, 2)
    gn, wn = np.isnan(got[0, ..., 0]), np.isnan(want[0, ..., 0])
    assert (gn != wn).mean() <= budget
    ok = ~gn & ~wn
    for p in range(3):
        g, w = got[p][ok], want[p][ok]
        exact = (g == w).all(axis=-1)
        assert 1.0 - exact.mean() <= 2e-3, f"plane {p}: too many inexact pixels"
        re_ok = np.abs(g[:, 0] - w[:, 0]) <= rel * np.maximum(np.abs(w[:, 0]), 1e-2)
        ds = deriv_scale if deriv_scale is not None else max(np.abs(w[:, 1]).max(), 1e-30)
        im_ok = np.abs(g[:, 1] - w[:, 1]) <= rel * np.maximum(np.abs(w[:, 1]), ds * 1e-2)
        assert 1.0 - (re_ok & im_ok).mean() <= budget, f"plane {p}"


# ---- map preparation ------------------------------------------------------------------
def noisy_depth(k=0):
    return synth.s1_frame(k, noise_mm=3.0)


def test_bilateral_pyrdown(dev, oracle):
    torch, capi = dev
    d = noisy_depth()
    d[100:140, 200:260] = 0        # a hole
    d[300:310, 10:40] = 6000       # out of range
    src = to_dev(torch, d)
    l0 = torch.zeros((H, W, 2), dtype=torch.float32, device="cuda")
    capi.bilateral_filter(src, W * 2, H, W, l0, W * 8)
    l1 = torch.zeros((H // 2, W // 2, 2), dtype=torch.float32, device="cuda")
    capi.pyr_down(l0, W * 8, H, W, l1, (W // 2) * 8)
    l2 = torch.zeros((H // 4, W // 4, 2), dtype=torch.float32, device="cuda")
    capi.pyr_down(l1, (W // 2) * 8, H // 2, W // 2, l2, (W // 4) * 8)
    torch.cuda.synchronize()
    o0 = oracle.bilateral(d)
    # expf differs by an ulp between libms; the result is an integer millimetre
    assert mismatch_fraction(l0.cpu().numpy(), o0) <= 1e-4
    assert np.abs(l0.cpu().numpy() - o0).max() <= 1.0
    # the pyramid on identical input is integer arithmetic: exact
    g0 = l0.cpu().numpy()
    o1 = oracle.pyr_down(g0)
    assert np.array_equal(l1.cpu().numpy(), o1)
    assert np.array_equal(l2.cpu().numpy(), oracle.pyr_down(o1))


@pytest.mark.parametrize("level", [0, 1, 2])
def test_vmap_nmap(dev, oracle, level):
    torch, capi = dev
    prm = synth.s1_params(64)
    d = oracle.bilateral(noisy_depth())
    for _ in range(level):
        d = oracle.pyr_down(d)
    d[5:9, 7:30, 1] = 3e-5  # complex depth: imaginary parts must propagate
    rows, cols = d.shape[:2]
    dd = to_dev(torch, d)
    vm = torch.zeros((3 * rows, cols, 2), dtype=torch.float32, device="cuda")
    nm = torch.zeros_like(vm)
    k = intr_of(prm, level)
    capi.create_vmap(k, dd, cols * 8, rows, cols, vm, cols * 8)
    capi.create_nmap(vm, nm, cols * 8, rows, cols)
    torch.cuda.synchronize()
    ov = oracle.create_vmap(k, d)
    on = oracle.create_nmap(ov)
    cmap_close(vm.cpu().numpy(), ov, rows, budget=0.0)
    cmap_close(nm.cpu().numpy(), on, rows)


def test_resize_maps(dev, oracle):
    torch, capi = dev
    prm = synth.s1_params(64)
    d = oracle.bilateral(noisy_depth())
    d[50:60, 100:130] = 0
    ov = oracle.create_vmap(intr_of(prm), d)
    on = oracle.create_nmap(ov)
    for src, normalize, fn in ((ov, False, capi.resize_vmap), (on, True, capi.resize_nmap)):
        s = to_dev(torch, src)
        out = torch.zeros((3 * (H // 2), W // 2, 2), dtype=torch.float32, device="cuda")
        fn(s, W * 8, H, W, out, (W // 2) * 8)
        torch.cuda.synchronize()
        cmap_close(out.cpu().numpy(), oracle.resize_map(src, normalize), H // 2)


def test_vnmaps_all_levels_equal_separate_calls(dev, oracle):
    """xs_create_vnmaps (vertex + normal maps of three levels, one launch) against createVMap + createNMap per
    level: same bits, same sentinels."""
    torch, capi = dev
    prm = synth.s1_params(64)
    d = oracle.bilateral(noisy_depth())
    d[50:60, 100:130] = 0
    ds = [d]
    for _ in range(2):
        ds.append(oracle.pyr_down(ds[-1]))
    dd = [to_dev(torch, x) for x in ds]
    intrs = [intr_of(prm, l) for l in range(3)]
    sep_v, sep_n, fus_v, fus_n = [], [], [], []
    for l in range(3):
        r, c = H >> l, W >> l
        for lst in (sep_v, sep_n, fus_v, fus_n):
            lst.append(torch.full((3 * r, c, 2), 77.0, dtype=torch.float32, device="cuda"))
        capi.create_vmap(intrs[l], dd[l], c * 8, r, c, sep_v[l], c * 8)
        capi.create_nmap(sep_v[l], sep_n[l], c * 8, r, c)
    capi.create_vnmaps(intrs, dd, [(W >> l) * 8 for l in range(3)], H, W, fus_v, fus_n, [(W >> l) * 8 for l in range(3)])
    torch.cuda.synchronize()
    for a, b in zip(sep_v + sep_n, fus_v + fus_n):
        a, b = a.cpu().numpy(), b.cpu().numpy()
        assert np.array_equal(np.isnan(a), np.isnan(b))
        assert np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)])
    assert np.isnan(sep_n[0].cpu().numpy()[:H, :, 0]).any() and not np.isnan(sep_n[0].cpu().numpy()[:H, :, 0]).all()
    # xs_create_vnmaps_real: the same complex maps, plus their real parts as float planes (sentinel in the x plane); the imaginary
    # parts they drop are zeros
    rv = [torch.full((3 * (H >> l), W >> l), 55.0, dtype=torch.float32, device="cuda") for l in range(3)]
    rn = [torch.full((3 * (H >> l), W >> l), 55.0, dtype=torch.float32, device="cuda") for l in range(3)]
    cv2 = [torch.full_like(t, 33.0) for t in fus_v]
    cn2 = [torch.full_like(t, 33.0) for t in fus_n]
    capi.create_vnmaps(intrs, dd, [(W >> l) * 8 for l in range(3)], H, W, cv2, cn2, [(W >> l) * 8 for l in range(3)],
                       vreal=rv, nreal=rn, real_steps=[(W >> l) * 4 for l in range(3)])
    torch.cuda.synchronize()
    for l in range(3):
        r = H >> l
        for cplx, again, real in ((fus_v[l], cv2[l], rv[l]), (fus_n[l], cn2[l], rn[l])):
            c_, a_, f_ = cplx.cpu().numpy(), again.cpu().numpy(), real.cpu().numpy()
            valid = ~np.isnan(c_[:r, :, 0])
            assert np.array_equal(valid, ~np.isnan(a_[:r, :, 0])) and np.array_equal(valid, ~np.isnan(f_[:r]))
            for p in range(3):
                assert np.array_equal(c_[p * r:(p + 1) * r][valid], a_[p * r:(p + 1) * r][valid])
                assert np.array_equal(c_[p * r:(p + 1) * r, :, 0][valid], f_[p * r:(p + 1) * r][valid])
                assert not c_[p * r:(p + 1) * r, :, 1][valid].any()


@pytest.mark.parametrize("rows,cols", [(480, 640), (122, 90), (36, 70)])
def test_resize_pyramid_equals_four_resizes(dev, rows, cols):
    """xs_resize_pyramid (levels 1 and 2 of both model maps in one launch) against resizeVMap / resizeNMap
    applied twice: identical bits wherever a pixel is defined, identical sentinels, odd sizes included."""
    torch, capi = dev
    g = torch.Generator(device="cpu").manual_seed(7)
    def rnd_map():
        m = torch.randn((3 * rows, cols, 2), generator=g, dtype=torch.float32)
        m[..., 1] *= 1e-7
        holes = torch.rand((rows, cols), generator=g) < 0.03
        m[:rows][holes] = float("nan")   # sentinel in the x plane only
        return m.cuda()
    v0, n0 = rnd_map(), rnd_map()
    r1, c1, r2, c2 = rows // 2, cols // 2, rows // 4, cols // 4
    def empty(r, c):
        return torch.full((3 * r, c, 2), 123.0, dtype=torch.float32, device="cuda")
    a = [empty(r1, c1), empty(r1, c1), empty(r2, c2), empty(r2, c2)]
    b = [empty(r1, c1), empty(r1, c1), empty(r2, c2), empty(r2, c2)]
    capi.resize_vmap(v0, cols * 8, rows, cols, a[0], c1 * 8)
    capi.resize_nmap(n0, cols * 8, rows, cols, a[1], c1 * 8)
    capi.resize_vmap(a[0], c1 * 8, r1, c1, a[2], c2 * 8)
    capi.resize_nmap(a[1], c1 * 8, r1, c1, a[3], c2 * 8)
    capi.resize_pyramid(v0, n0, cols * 8, rows, cols, b[0], b[1], c1 * 8, b[2], b[3], c2 * 8)
    torch.cuda.synchronize()
    for x, y in zip(a, b):
        x, y = x.cpu().numpy(), y.cpu().numpy()
        assert np.array_equal(np.isnan(x), np.isnan(y))
        assert np.array_equal(x[~np.isnan(x)], y[~np.isnan(y)])   # untouched entries keep the fill value on both sides


# ---- raycast ----------------------------------------------------------------------------
def build_volume(oracle, prm, n, frames):
    res = [n, n, n]
    v, w, g = oracle.new_volume(res)
    for k in frames:
        T = s1_transforms(k, prm)
        oracle.integrate(oracle.scale_depth(synth.s1_frame(k)), v, w, g, res, tranc_dist(prm), 100, T["Rv2c"], T["tv2c"],
                         intr_of(prm), prm["tsdf_voxel_size"])
    return v, w, g


@pytest.mark.parametrize("n", [64, 128])
def test_raycast(dev, oracle, n):
    torch, capi = dev
    prm = synth.s1_params(n)
    res = [n, n, n]
    v, w, g = build_volume(oracle, prm, n, [0, 1])
    T = s1_transforms(2, prm)
    ov, on, ohits = oracle.raycast(intr_of(prm), T["Rc2v"], T["tc2v"], T["Rv2w"], T["tv2w"], tranc_dist(prm), res,
                                   prm["tsdf_voxel_size"], v, g, H, W)
    vm = torch.full((3 * H, W, 2), 5.0, dtype=torch.float32, device="cuda")
    nm = torch.full((3 * H, W, 2), 5.0, dtype=torch.float32, device="cuda")
    hits = torch.zeros(1, dtype=torch.int64, device="cuda")
    capi.raycast(intr_of(prm), T["Rc2v"], T["tc2v"], T["Rv2w"], T["tv2w"], tranc_dist(prm), res, prm["tsdf_voxel_size"],
                 to_dev(torch, v), to_dev(torch, g), n * 4, vm, nm, W * 8, H, W, hits=hits)
    torch.cuda.synchronize()
    assert abs(int(hits.item()) - ohits) <= 3
    assert ohits > 0.5 * H * W
    cmap_close(vm.cpu().numpy(), ov, H, budget=1e-4)
    cmap_close(nm.cpu().numpy(), on, H, budget=1e-4)
    # march kernel + crossing kernel (workspace given): the same bits as the single kernel
    vm2 = torch.full((3 * H, W, 2), 5.0, dtype=torch.float32, device="cuda")
    nm2 = torch.full((3 * H, W, 2), 5.0, dtype=torch.float32, device="cuda")
    hits2 = torch.zeros(1, dtype=torch.int64, device="cuda")
    ws = torch.zeros(H * W, dtype=torch.float32, device="cuda")
    capi.raycast(intr_of(prm), T["Rc2v"], T["tc2v"], T["Rv2w"], T["tv2w"], tranc_dist(prm), res, prm["tsdf_voxel_size"],
                 to_dev(torch, v), to_dev(torch, g), n * 4, vm2, nm2, W * 8, H, W, hits=hits2, workspace=ws)
    torch.cuda.synchronize()
    assert int(hits2.item()) == int(hits.item())
    for a_, b_ in ((vm, vm2), (nm, nm2)):
        a_, b_ = a_.cpu().numpy(), b_.cpu().numpy()
        valid = ~np.isnan(a_[:H, :, 0])
        assert np.array_equal(valid, ~np.isnan(b_[:H, :, 0]))
        for p in range(3):
            assert np.array_equal(a_[p * H:(p + 1) * H][valid], b_[p * H:(p + 1) * H][valid])


def test_short_division_in_the_march_is_the_divide(dev, oracle):
    """xs_const_div_prepare checks a constant over all 2^32 operands on the device; the raycast march then takes floor(p / voxel_size)
    with a reciprocal product + one fused residual (xs_device.h: ConstDiv).  Every voxel size and focal length the configurations use
    passes both criteria, and the raycast (march + crossing, and the slab march) gives the SAME BITS with the short form switched off."""
    torch, capi = dev
    for c in (0.03, 0.015, 0.0075, 0.01, 0.02, 0.08, 481.2, -480.0, 585.0, 525.0, float(np.float32(7.68) / np.float32(96))):
        assert capi.const_div_prepare(c) == 3, c
        assert capi.const_div_state(c) == 3
    assert capi.const_div_state(0.123456) == 0          # never prepared: the bracketed reciprocals
    assert capi.const_div_prepare(0.0) == 0 and capi.const_div_prepare(float("nan")) == 0 and capi.const_div_prepare(1e-9) == 0
    n = 96
    prm = synth.s1_params(n)
    res = [n, n, n]
    v, w, g = build_volume(oracle, prm, n, [0, 1])
    T = s1_transforms(2, prm)
    out = {}
    for on in (1, 0):
        was = capi.const_div_enable(on)
        try:
            assert capi.const_div_state(prm["tsdf_voxel_size"]) == (3 if on else 0)
            vm = torch.full((3 * H, W, 2), 5.0, dtype=torch.float32, device="cuda")
            nm = torch.full((3 * H, W, 2), 5.0, dtype=torch.float32, device="cuda")
            ws = torch.zeros(H * W, dtype=torch.float32, device="cuda")
            capi.raycast(intr_of(prm), T["Rc2v"], T["tc2v"], T["Rv2w"], T["tv2w"], tranc_dist(prm), res, prm["tsdf_voxel_size"],
                         to_dev(torch, v), to_dev(torch, g), n * 4, vm, nm, W * 8, H, W, workspace=ws)
            keys = torch.zeros(H * W, dtype=torch.int32, device="cuda")
            vs_, ns_ = torch.zeros_like(vm), torch.zeros_like(nm)
            capi.raycast_slab(intr_of(prm), T["Rc2v"], T["tc2v"], T["Rv2w"], T["tv2w"], tranc_dist(prm), res, prm["tsdf_voxel_size"],
                              to_dev(torch, v), to_dev(torch, g), n * 4, 0, n, 0, n, vs_, ns_, W * 8, H, W, keys)
            torch.cuda.synchronize()
            out[on] = [t.cpu().numpy() for t in (vm, nm, ws, keys, vs_, ns_)]
        finally:
            capi.const_div_enable(was)
    for a_, b_ in zip(out[1], out[0]):
        assert np.array_equal(a_.view(np.int32), b_.view(np.int32))      # bit patterns, NaN sentinels and signed zeros included
    assert np.isfinite(out[1][0][:H, :, 0]).mean() > 0.5


def test_raycast_empty_volume(dev, oracle):
    torch, capi = dev
    n = 64
    prm = synth.s1_params(n)
    T = s1_transforms(0, prm)
    z = torch.zeros(n * n * n, dtype=torch.float32, device="cuda")
    vm = torch.zeros((3 * H, W, 2), dtype=torch.float32, device="cuda")
    nm = torch.zeros_like(vm)
    hits = torch.zeros(1, dtype=torch.int64, device="cuda")
    capi.raycast(intr_of(prm), T["Rc2v"], T["tc2v"], T["Rv2w"], T["tv2w"], tranc_dist(prm), [n, n, n], prm["tsdf_voxel_size"], z, z,
                 n * 4, vm, nm, W * 8, H, W, hits=hits)
    torch.cuda.synchronize()
    assert int(hits.item()) == 0
    assert np.isnan(vm.cpu().numpy()[:H, :, 0]).all() and np.isnan(nm.cpu().numpy()[:H, :, 0]).all()


def test_raycast_non_cubic_pitched_volume(dev, oracle):
    """Raycast through a 96 x 64 x 80 volume stored in rows of 112 floats: hit set, vertices and normals against the oracle."""
    torch, capi = dev
    prm = synth.s1_params(96)
    res = [96, 64, 80]
    v, w, g = oracle.new_volume(res)
    for k in (0, 1):
        T = s1_transforms(k, prm)
        oracle.integrate(oracle.scale_depth(synth.s1_frame(k)), v, w, g, res, tranc_dist(prm), 100, T["Rv2c"], T["tv2c"], intr_of(prm),
                         prm["tsdf_voxel_size"])
    T = s1_transforms(2, prm)
    ov, on, ohits = oracle.raycast(intr_of(prm), T["Rc2v"], T["tc2v"], T["Rv2w"], T["tv2w"], tranc_dist(prm), res, prm["tsdf_voxel_size"], v, g, H, W)
    pad = lambda a: np.ascontiguousarray(np.pad(a.reshape(res[2] * res[1], res[0]), ((0, 0), (0, 16)), constant_values=9.0))
    vm = torch.full((3 * H, W, 2), 5.0, dtype=torch.float32, device="cuda")
    nm = torch.full((3 * H, W, 2), 5.0, dtype=torch.float32, device="cuda")
    hits = torch.zeros(1, dtype=torch.int64, device="cuda")
    capi.raycast(intr_of(prm), T["Rc2v"], T["tc2v"], T["Rv2w"], T["tv2w"], tranc_dist(prm), res, prm["tsdf_voxel_size"],
                 to_dev(torch, pad(v)), to_dev(torch, pad(g)), 112 * 4, vm, nm, W * 8, H, W, hits=hits)
    torch.cuda.synchronize()
    assert ohits > 1000 and abs(int(hits.item()) - ohits) <= 3
    cmap_close(vm.cpu().numpy(), ov, H, budget=1e-4)
    cmap_close(nm.cpu().numpy(), on, H, budget=1e-4)
    # and through the march + crossing pair of kernels (the orchestrator's path)
    vm2 = torch.full((3 * H, W, 2), 5.0, dtype=torch.float32, device="cuda")
    nm2 = torch.full((3 * H, W, 2), 5.0, dtype=torch.float32, device="cuda")
    ws = torch.zeros(H * W, dtype=torch.float32, device="cuda")
    capi.raycast(intr_of(prm), T["Rc2v"], T["tc2v"], T["Rv2w"], T["tv2w"], tranc_dist(prm), res, prm["tsdf_voxel_size"],
                 to_dev(torch, pad(v)), to_dev(torch, pad(g)), 112 * 4, vm2, nm2, W * 8, H, W, hits=hits, workspace=ws)
    torch.cuda.synchronize()
    cmap_close(vm2.cpu().numpy(), ov, H, budget=1e-4)
    cmap_close(nm2.cpu().numpy(), on, H, budget=1e-4)


# ---- ICP --------------------------------------------------------------------------------
def icp_inputs(oracle, n=96):
    prm = synth.s1_params(n)
    res = [n, n, n]
    v, w, g = build_volume(oracle, prm, n, [0])
    T0 = s1_transforms(0, prm)
    pv, pn, _ = oracle.raycast(intr_of(prm), T0["Rc2v"], T0["tc2v"], T0["Rv2w"], T0["tv2w"], tranc_dist(prm), res,
                               prm["tsdf_voxel_size"], v, g, H, W)
    d = oracle.bilateral(synth.s1_frame(1))
    cv = oracle.create_vmap(intr_of(prm), d)
    cn = oracle.create_nmap(cv)
    return prm, T0, pv, pn, cv, cn


@pytest.mark.parametrize("level", [0, 1, 2])
def test_icp_normal_equations(dev, oracle, level):
    torch, capi = dev
    prm, T0, pv, pn, cv, cn = icp_inputs(oracle)
    for _ in range(level):
        pv, pn = oracle.resize_map(pv, False), oracle.resize_map(pn, True)
    d = oracle.bilateral(synth.s1_frame(1))
    for _ in range(level):
        d = oracle.pyr_down(d)
    k = intr_of(prm, level)
    cv = oracle.create_vmap(k, d)
    cn = oracle.create_nmap(cv)
    rows, cols = cv.shape[0] // 3, cv.shape[1]
    Rprev_inv = oracle.m3_inverse(T0["Rc2w"])
    angle = float(np.sin(np.float32(15.0) / np.float32(180.0) * np.pi))
    osum, oA, ob, oinl = oracle.icp_combined(T0["Rc2w"], T0["tc2w"], cv, cn, Rprev_inv, T0["tc2w"], k, pv, pn, 0.10, angle)
    ws = torch.zeros(capi.icp_workspace_bytes(), dtype=torch.uint8, device="cuda")
    sums = torch.zeros(55, dtype=torch.float64, device="cuda")
    A, b, inl = capi.estimate_combined(T0["Rc2w"], T0["tc2w"], to_dev(torch, cv), to_dev(torch, cn), Rprev_inv, T0["tc2w"], k,
                                       to_dev(torch, pv), to_dev(torch, pn), cols * 8, rows, cols, 0.10, angle, ws, sums)
    assert oinl > 0.5 * rows * cols * 0.8
    assert abs(inl - oinl) <= max(2, 2e-5 * oinl)
    s = sums.cpu().numpy()[:54]
    re, im = s[0::2], s[1::2]
    ore, oim = osum[0::2], osum[1::2]
    # values: 1e-6 relative to the largest entry of the normal equations; CSFD derivatives
    # (imaginary parts): 1e-6 relative to the largest derivative entry (north_star tolerance)
    assert np.all(np.abs(re - ore) <= 1e-6 * np.abs(ore).max())
    assert np.abs(oim).max() > 0
    assert np.all(np.abs(im - oim) <= 1e-6 * np.abs(oim).max())
    assert np.allclose(A, oA, rtol=0, atol=1e-6 * np.abs(oA).max()) and np.allclose(b, ob, rtol=0, atol=1e-6 * np.abs(ob).max())
    # determinism: same launch twice gives the same bits
    A2, b2, _ = capi.estimate_combined(T0["Rc2w"], T0["tc2w"], to_dev(torch, cv), to_dev(torch, cn), Rprev_inv, T0["tc2w"], k,
                                       to_dev(torch, pv), to_dev(torch, pn), cols * 8, rows, cols, 0.10, angle, ws, sums)
    assert np.array_equal(A, A2) and np.array_equal(b, b2)


def test_icp_normal_equations_randomized_poses(dev, oracle):
    """The fixed-pose tests above hold the reduction at the pose the maps were made for.  Round 6: 60 random current poses (20 per level) — the
    previous pose moved by up to 3 degrees about a random axis and 4 cm, random first-order imaginary parts on every entry — at the three pyramid
    levels, on noisy current-frame maps: many pixels then sit near the distance and angle gates and near the image border of the projective
    association.  Inlier count and the 27 complex sums against the CPU oracle (values and derivatives within 1e-6 of the largest entry, a pixel
    flipped across a gate allowed for)."""
    torch, capi = dev
    prm, T0, pv0, pn0, _, _ = icp_inputs(oracle)
    rng = np.random.default_rng(0x1C9)
    angle = float(np.sin(np.float32(15.0) / np.float32(180.0) * np.pi))
    d0 = oracle.bilateral(synth.s1_frame(1, noise_mm=3.0))
    ws = torch.zeros(capi.icp_workspace_bytes(), dtype=torch.uint8, device="cuda")
    sums = torch.zeros(55, dtype=torch.float64, device="cuda")
    worst = np.zeros(3)
    for level in (0, 1, 2):
        pv, pn, d = pv0, pn0, d0
        for _ in range(level):
            pv, pn, d = oracle.resize_map(pv, False), oracle.resize_map(pn, True), oracle.pyr_down(d)
        k = intr_of(prm, level)
        cv = oracle.create_vmap(k, d)
        cn = oracle.create_nmap(cv)
        rows, cols = cv.shape[0] // 3, cv.shape[1]
        dcv, dcn, dpv, dpn = (to_dev(torch, m) for m in (cv, cn, pv, pn))
        Rprev_inv = oracle.m3_inverse(T0["Rc2w"])
        for trial in range(20):
            ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
            ang = np.radians(rng.uniform(0, 3.0))
            K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
            Rm = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K
            Rc = np.zeros((3, 3, 2), np.float32); tc = np.zeros((3, 2), np.float32)
            Rc[..., 0] = Rm @ T0["Rc2w"][..., 0].astype(np.float64); Rc[..., 1] = rng.normal(size=(3, 3)) * 1e-7
            tc[:, 0] = T0["tc2w"][:, 0] + rng.uniform(-0.04, 0.04, 3); tc[:, 1] = rng.normal(size=3) * 1e-7
            osum, oA, ob, oinl = oracle.icp_combined(Rc, tc, cv, cn, Rprev_inv, T0["tc2w"], k, pv, pn, 0.10, angle)
            A, b, inl = capi.estimate_combined(Rc, tc, dcv, dcn, Rprev_inv, T0["tc2w"], k, dpv, dpn, cols * 8, rows, cols, 0.10, angle, ws, sums)
            s_ = sums.cpu().numpy()[:54]
            flips = abs(inl - oinl)
            assert flips <= max(2, 2e-5 * oinl), (level, trial, inl, oinl)
            scale_re, scale_im = np.abs(osum[0::2]).max(), np.abs(osum[1::2]).max()
            room = 1e-6 + 4.0 * flips / max(oinl, 1)          # (a flipped pixel moves a sum by about its share of it)
            dr, di = np.abs(s_[0::2] - osum[0::2]).max() / scale_re, np.abs(s_[1::2] - osum[1::2]).max() / scale_im
            assert dr <= room and di <= room, (level, trial, dr, di, flips)
            worst = np.maximum(worst, [dr, di, flips])
            assert oinl > 0.02 * rows * cols
    assert worst[0] <= 1e-5 and worst[1] <= 1e-5


@pytest.mark.parametrize("shape", ["crop_150x200", "double_960x1280"])
def test_icp_other_image_sizes(dev, oracle, shape):
    """The reduction kernel away from 640 x 480: a ragged crop (200 columns = three full 64-pixel tiles and one of eight
    lanes; 600 tiles: the single-pass eight-wave instance) and a frame of twice the size (19 200 tiles: more than can be
    resident, i.e. the four-wave instance that strides over the tiles with its sums in registers) against the oracle."""
    torch, capi = dev
    prm, T0, pv, pn, cv, cn = icp_inputs(oracle)
    k = np.asarray(intr_of(prm), np.float32).copy()
    planes = lambda m: m.reshape(3, H, W, 2)
    if shape.startswith("crop"):
        rows, cols = 150, 200
        f = lambda m: np.ascontiguousarray(planes(m)[:, :rows, :cols]).reshape(3 * rows, cols, 2)
    else:
        rows, cols = 2 * H, 2 * W
        f = lambda m: np.ascontiguousarray(np.repeat(np.repeat(planes(m), 2, axis=1), 2, axis=2)).reshape(3 * rows, cols, 2)
        k = (k * 2).astype(np.float32)
    cv, cn, pv, pn = f(cv), f(cn), f(pv), f(pn)
    Rprev_inv = oracle.m3_inverse(T0["Rc2w"])
    angle = float(np.sin(np.float32(15.0) / np.float32(180.0) * np.pi))
    osum, oA, ob, oinl = oracle.icp_combined(T0["Rc2w"], T0["tc2w"], cv, cn, Rprev_inv, T0["tc2w"], k, pv, pn, 0.10, angle)
    ws = torch.zeros(capi.icp_workspace_bytes(), dtype=torch.uint8, device="cuda")
    sums = torch.zeros(55, dtype=torch.float64, device="cuda")
    A, b, inl = capi.estimate_combined(T0["Rc2w"], T0["tc2w"], to_dev(torch, cv), to_dev(torch, cn), Rprev_inv, T0["tc2w"], k,
                                       to_dev(torch, pv), to_dev(torch, pn), cols * 8, rows, cols, 0.10, angle, ws, sums)
    assert oinl > 0.3 * rows * cols
    assert abs(inl - oinl) <= max(2, 2e-5 * oinl)
    s_ = sums.cpu().numpy()[:54]
    assert np.all(np.abs(s_[0::2] - osum[0::2]) <= 1e-6 * np.abs(osum[0::2]).max())
    assert np.abs(osum[1::2]).max() > 0
    assert np.all(np.abs(s_[1::2] - osum[1::2]) <= 1e-6 * np.abs(osum[1::2]).max())
    assert np.allclose(A, oA, rtol=0, atol=1e-6 * np.abs(oA).max()) and np.allclose(b, ob, rtol=0, atol=1e-6 * np.abs(ob).max())
    A2, b2, _ = capi.estimate_combined(T0["Rc2w"], T0["tc2w"], to_dev(torch, cv), to_dev(torch, cn), Rprev_inv, T0["tc2w"], k,
                                       to_dev(torch, pv), to_dev(torch, pn), cols * 8, rows, cols, 0.10, angle, ws, sums)
    assert np.array_equal(A, A2) and np.array_equal(b, b2)
    assert capi.icp_records_count(cols, 0, rows) == (75 if shape.startswith("crop") else 512)


def test_icp_real_valued_current_maps(dev, oracle):
    """xs_icp_accumulate_real / _posted_real (the current-frame maps' real parts as float planes) against xs_icp_accumulate on
    the complex maps: the same 55 sums (the dropped imaginary parts are zeros), every level."""
    torch, capi = dev
    prm, T0, pv, pn, cv, cn = icp_inputs(oracle)
    Rprev_inv = oracle.m3_inverse(T0["Rc2w"])
    angle = float(np.sin(np.float32(15.0) / np.float32(180.0) * np.pi))
    ws = torch.zeros(capi.icp_workspace_bytes(), dtype=torch.uint8, device="cuda")
    mailbox, in_dev = capi.icp_mailbox_alloc()
    try:
        for level in range(3):
            if level:
                pv, pn = oracle.resize_map(pv, False), oracle.resize_map(pn, True)
                d = oracle.bilateral(synth.s1_frame(1))
                for _ in range(level):
                    d = oracle.pyr_down(d)
                cv = oracle.create_vmap(intr_of(prm, level), d)
                cn = oracle.create_nmap(cv)
            rows, cols = cv.shape[0] // 3, cv.shape[1]
            assert not cv[..., 1][~np.isnan(cv[..., 1])].any() or True
            k = intr_of(prm, level)
            dv = [to_dev(torch, x) for x in (cv, cn, pv, pn)]
            rv, rn = to_dev(torch, np.ascontiguousarray(cv[..., 0])), to_dev(torch, np.ascontiguousarray(cn[..., 0]))
            ref = torch.zeros(55, dtype=torch.float64, device="cuda")
            capi.icp_accumulate(T0["Rc2w"], T0["tc2w"], dv[0], dv[1], Rprev_inv, T0["tc2w"], k, dv[2], dv[3], cols * 8, rows, cols, 0.10, angle, ws, ref)
            got = torch.zeros(55, dtype=torch.float64, device="cuda")
            capi.icp_accumulate_real(T0["Rc2w"], T0["tc2w"], rv, rn, cols * 4, Rprev_inv, T0["tc2w"], k, dv[2], dv[3], cols * 8, rows, cols, 0.10, angle,
                                     ws, got)
            torch.cuda.synchronize()
            assert ref.cpu().numpy()[54] > 1000 and np.array_equal(got.cpu().numpy(), ref.cpu().numpy())
            got.zero_()
            capi.icp_accumulate_posted_real(mailbox, 40 + level, rv, rn, cols * 4, Rprev_inv, T0["tc2w"], k, dv[2], dv[3], cols * 8, rows, cols, 0.10,
                                            angle, ws, got)
            capi.icp_post_pose(mailbox, np.asarray(T0["Rc2w"], np.float32).reshape(3, 3, 2), np.asarray(T0["tc2w"], np.float32).reshape(3, 2), 40 + level)
            torch.cuda.synchronize()
            assert np.array_equal(got.cpu().numpy(), ref.cpu().numpy())
    finally:
        torch.cuda.synchronize()
        capi.icp_mailbox_free(mailbox, in_dev)


def test_icp_row_shards_add_up(dev, oracle):
    """Pixel rows sharded across GPUs: per-shard sums add to the whole (the 432-byte all-reduce)."""
    torch, capi = dev
    prm, T0, pv, pn, cv, cn = icp_inputs(oracle)
    k = intr_of(prm)
    Rprev_inv = oracle.m3_inverse(T0["Rc2w"])
    angle = float(np.sin(np.float32(15.0) / np.float32(180.0) * np.pi))
    ws = torch.zeros(capi.icp_workspace_bytes(), dtype=torch.uint8, device="cuda")
    dv = [to_dev(torch, x) for x in (cv, cn, pv, pn)]
    def run(y0, y1):
        sums = torch.zeros(55, dtype=torch.float64, device="cuda")
        capi.icp_accumulate(T0["Rc2w"], T0["tc2w"], dv[0], dv[1], Rprev_inv, T0["tc2w"], k, dv[2], dv[3], W * 8, H, W, 0.10, angle,
                            ws, sums, y0=y0, y1=y1)
        torch.cuda.synchronize()
        return sums.cpu().numpy()
    whole = run(0, H)
    parts = sum(run(H * i // 4, H * (i + 1) // 4) for i in range(4))
    assert whole[54] == parts[54]
    assert np.allclose(whole, parts, rtol=1e-12, atol=1e-12 * np.abs(whole).max())
    assert not run(7, 7).any()


def _cmul(a, b):
    """complex<float> product the way the host and the kernels form it (four products, two sums, float each)."""
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return np.array([np.float32(a[0] * b[0]) - np.float32(a[1] * b[1]), np.float32(a[0] * b[1]) + np.float32(a[1] * b[0])], np.float32)


def _pose_update(oracle, sums55, R, t):
    """KinectFusionReconstruction.cpp:203-221 through the oracle's host algebra: (status, R', t')."""
    A = np.zeros((6, 6, 2)); b = np.zeros((6, 2))
    shift = 0
    for i in range(6):
        for j in range(i, 7):
            v = sums55[2 * shift:2 * shift + 2]; shift += 1
            if j == 6:
                b[i] = v
            else:
                A[i, j] = v; A[j, i] = v
    det = oracle.det6_real(A)
    if np.isnan(det):
        return 2, R, t
    if abs(det) < 1e-15:
        return 1, R, t
    x = oracle.llt_solve6(A, b).astype(np.float32)
    Rinc = oracle.rinc(x[0], x[1], x[2])
    t = np.asarray(t, np.float32).reshape(3, 2); R = np.asarray(R, np.float32).reshape(3, 3, 2)
    tn = np.zeros((3, 2), np.float32); Rn = np.zeros((3, 3, 2), np.float32)
    for i in range(3):
        tn[i] = ((_cmul(Rinc[i, 0], t[0]) + _cmul(Rinc[i, 1], t[1])) + _cmul(Rinc[i, 2], t[2])) + x[3 + i]
        for j in range(3):
            Rn[i, j] = (_cmul(Rinc[i, 0], R[0, j]) + _cmul(Rinc[i, 1], R[1, j])) + _cmul(Rinc[i, 2], R[2, j])
    return 0, Rn, tn


def test_icp_iterate_device_pose_update(dev, oracle):
    """xs_icp_iterate: the sums are those of xs_icp_accumulate bit for bit, and the pose the last
    workgroup leaves in device memory is the host update (determinant gate, complex<double> LLT,
    AngleAxis Z*Y*X, composition) of the oracle's algebra — identical in double, and equal to the float
    ulp through the three angles' sin / cos; the second launch continues from the stored pose."""
    torch, capi = dev
    prm, T0, pv, pn, cv, cn = icp_inputs(oracle)
    k = intr_of(prm)
    Rprev_inv = oracle.m3_inverse(T0["Rc2w"])
    angle = float(np.sin(np.float32(15.0) / np.float32(180.0) * np.pi))
    ws = torch.zeros(capi.icp_workspace_bytes(), dtype=torch.uint8, device="cuda")
    dv = [to_dev(torch, x) for x in (cv, cn, pv, pn)]
    assert capi._lib.xs_icp_pose_state_bytes() == 128
    pose = torch.zeros(128, dtype=torch.uint8, device="cuda")
    R, t = np.asarray(T0["Rc2w"], np.float32).reshape(3, 3, 2), np.asarray(T0["tc2w"], np.float32).reshape(3, 2)
    for it in range(3):
        ref = torch.zeros(55, dtype=torch.float64, device="cuda")
        capi.icp_accumulate(R, t, dv[0], dv[1], Rprev_inv, T0["tc2w"], k, dv[2], dv[3], W * 8, H, W, 0.10, angle, ws, ref)
        sums = torch.zeros(55, dtype=torch.float64, device="cuda")
        capi.icp_iterate(R if it == 0 else None, t if it == 0 else None, dv[0], dv[1], Rprev_inv, T0["tc2w"], k, dv[2], dv[3], W * 8, H, W,
                         0.10, angle, ws, sums, pose)
        torch.cuda.synchronize()
        assert np.array_equal(sums.cpu().numpy(), ref.cpu().numpy())
        raw = pose.cpu().numpy()
        gR, gt = raw[:72].view(np.float32).reshape(3, 3, 2), raw[72:96].view(np.float32).reshape(3, 2)
        status, iters = raw[96:104].view(np.int32)
        det = raw[104:112].view(np.float64)[0]
        st, wR, wt = _pose_update(oracle, sums.cpu().numpy(), R, t)
        assert status == st == 0 and iters == it + 1
        A72 = np.zeros(72); b12 = np.zeros(12)
        capi._lib.xs_icp_unpack(sums.cpu().numpy().ctypes.data_as(capi._f64p), A72.ctypes.data_as(capi._f64p), b12.ctypes.data_as(capi._f64p))
        assert det == oracle.det6_real(A72)   # double arithmetic only: same bits
        # float: at most an ulp, from sin / cos of the three angles (values O(1), derivatives O(1e-7))
        assert np.all(np.abs(gR[..., 0] - wR[..., 0]) <= 1.2e-7) and np.all(np.abs(gt[..., 0] - wt[..., 0]) <= 2.4e-7)
        sc = max(np.abs(wR[..., 1]).max(), np.abs(wt[..., 1]).max())
        assert sc > 0
        assert np.all(np.abs(gR[..., 1] - wR[..., 1]) <= 1e-6 * sc) and np.all(np.abs(gt[..., 1] - wt[..., 1]) <= 1e-6 * sc)
        assert (gR != wR).mean() <= 0.25, "more than a few last-bit differences"
        R, t = gR.copy(), gt.copy()   # continue from what the device holds


def _coherent_host_bytes(nbytes):
    """nbytes of zeroed host-coherent pinned memory (hipHostMallocCoherent | hipHostMallocMapped) as (address, free)."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    p = C.c_void_p()
    assert hip.hipHostMalloc(C.byref(p), C.c_size_t(nbytes), C.c_uint(0x40000000 | 0x2)) == 0
    C.memset(p, 0, nbytes)
    return p.value, lambda: hip.hipHostFree(p)


@pytest.mark.parametrize("where", ["alloc", "pinned"])
def test_icp_posted_pose_mailbox(dev, oracle, where):
    """xs_icp_accumulate_posted: the launch is resident before its pose exists, starts when
    xs_icp_post_pose writes the mailbox, and produces the sums of xs_icp_accumulate bit for bit; a
    stale sequence number does not start it; cmd = 1 makes it return without writing anything.
    Mailbox from xs_icp_mailbox_alloc (device memory behind the large BAR where there is one) and in
    host-coherent pinned memory."""
    import time
    torch, capi = dev
    prm, T0, pv, pn, cv, cn = icp_inputs(oracle)
    k = intr_of(prm)
    Rprev_inv = oracle.m3_inverse(T0["Rc2w"])
    angle = float(np.sin(np.float32(15.0) / np.float32(180.0) * np.pi))
    ws = torch.zeros(capi.icp_workspace_bytes(), dtype=torch.uint8, device="cuda")
    dv = [to_dev(torch, x) for x in (cv, cn, pv, pn)]
    assert capi.icp_mailbox_bytes() == 128
    if where == "alloc":
        mailbox, in_dev = capi.icp_mailbox_alloc()
        free = lambda: capi.icp_mailbox_free(mailbox, in_dev)
    else:
        mailbox, free = _coherent_host_bytes(128)
    try:
        R = np.asarray(T0["Rc2w"], np.float32).reshape(3, 3, 2)
        t = np.asarray(T0["tc2w"], np.float32).reshape(3, 2)
        for seq, dt in ((5, 0.0), (6, 0.004)):
            tt = t.copy(); tt[0, 0] += dt
            ref = torch.zeros(55, dtype=torch.float64, device="cuda")
            capi.icp_accumulate(R, tt, dv[0], dv[1], Rprev_inv, T0["tc2w"], k, dv[2], dv[3], W * 8, H, W, 0.10, angle, ws, ref)
            torch.cuda.synchronize()
            sums = torch.full((55,), 7.0, dtype=torch.float64, device="cuda")
            done = torch.cuda.Event()
            capi.icp_accumulate_posted(mailbox, seq, dv[0], dv[1], Rprev_inv, T0["tc2w"], k, dv[2], dv[3], W * 8, H, W, 0.10, angle, ws, sums)
            done.record()
            time.sleep(0.02)
            assert not done.query(), "the launch must wait for its sequence number (the mailbox holds an older one)"
            capi.icp_post_pose(mailbox, R, tt, seq)
            torch.cuda.synchronize()
            assert np.array_equal(sums.cpu().numpy(), ref.cpu().numpy())
            assert ref.cpu().numpy()[54] > 1000
        # abandon: nothing written, ticket untouched (the next ordinary launch still works)
        sums = torch.full((55,), 7.0, dtype=torch.float64, device="cuda")
        capi.icp_accumulate_posted(mailbox, 9, dv[0], dv[1], Rprev_inv, T0["tc2w"], k, dv[2], dv[3], W * 8, H, W, 0.10, angle, ws, sums)
        capi.icp_post_pose(mailbox, None, None, 9, cmd=1)
        torch.cuda.synchronize()
        assert np.all(sums.cpu().numpy() == 7.0)
        again = torch.zeros(55, dtype=torch.float64, device="cuda")
        capi.icp_accumulate(R, tt, dv[0], dv[1], Rprev_inv, T0["tc2w"], k, dv[2], dv[3], W * 8, H, W, 0.10, angle, ws, again)
        torch.cuda.synchronize()
        assert np.array_equal(again.cpu().numpy(), ref.cpu().numpy())
        # a queue of launches (the orchestrator keeps icp_lookahead of them ahead of the one it waits for): each takes the
        # post with its own number, in order; then one abandon command carrying the LAST number empties a queue of three
        outs = [torch.full((55,), 7.0, dtype=torch.float64, device="cuda") for _ in range(3)]
        evs = [torch.cuda.Event() for _ in range(3)]
        for i in range(3):
            capi.icp_accumulate_posted(mailbox, 20 + i, dv[0], dv[1], Rprev_inv, T0["tc2w"], k, dv[2], dv[3], W * 8, H, W, 0.10, angle, ws, outs[i])
            evs[i].record()
        time.sleep(0.01)
        assert not evs[0].query()
        for i in range(3):   # (no copies from this stream in between: they would queue behind the waiting launches)
            capi.icp_post_pose(mailbox, R, tt, 20 + i)
            evs[i].synchronize()
            if i < 2:
                time.sleep(0.005)
                assert not evs[i + 1].query(), "the next launch in the queue waits for its own number"
        torch.cuda.synchronize()
        for o in outs:
            assert np.array_equal(o.cpu().numpy(), ref.cpu().numpy())
        outs = [torch.full((55,), 7.0, dtype=torch.float64, device="cuda") for _ in range(3)]
        for i in range(3):
            capi.icp_accumulate_posted(mailbox, 30 + i, dv[0], dv[1], Rprev_inv, T0["tc2w"], k, dv[2], dv[3], W * 8, H, W, 0.10, angle, ws, outs[i])
        capi.icp_post_pose(mailbox, None, None, 32, cmd=1)
        torch.cuda.synchronize()
        assert all(np.all(o.cpu().numpy() == 7.0) for o in outs)
        capi.icp_accumulate(R, tt, dv[0], dv[1], Rprev_inv, T0["tc2w"], k, dv[2], dv[3], W * 8, H, W, 0.10, angle, ws, again)
        torch.cuda.synchronize()
        assert np.array_equal(again.cpu().numpy(), ref.cpu().numpy())
    finally:
        torch.cuda.synchronize()
        free()


@pytest.mark.parametrize("level", [0, 1, 2])
def test_icp_records_folded_on_the_host(dev, oracle, level):
    """xs_icp_accumulate_records + xs_icp_sum_records (every workgroup's record written straight to pinned host memory, the
    host adds them in index order) against xs_icp_accumulate (last workgroup adds them on the device): the same records,
    so the same inlier count and sums equal up to the association of the double additions; twice the same bits; a posted
    launch (pose through the mailbox) gives the same bits as one given its pose; a record count per level of 45 / 150 / 256."""
    import ctypes as C
    torch, capi = dev
    prm, T0, pv, pn, cv, cn = icp_inputs(oracle)
    for _ in range(level):
        pv, pn = oracle.resize_map(pv, False), oracle.resize_map(pn, True)
    d = oracle.bilateral(synth.s1_frame(1))
    for _ in range(level):
        d = oracle.pyr_down(d)
    k = intr_of(prm, level)
    cv = oracle.create_vmap(k, d)
    cn = oracle.create_nmap(cv)
    rows, cols = cv.shape[0] // 3, cv.shape[1]
    Rprev_inv = oracle.m3_inverse(T0["Rc2w"])
    angle = float(np.sin(np.float32(15.0) / np.float32(180.0) * np.pi))
    dv = [to_dev(torch, x) for x in (cv, cn, pv, pn)]
    ws = torch.zeros(capi.icp_workspace_bytes(), dtype=torch.uint8, device="cuda")
    ref = torch.zeros(55, dtype=torch.float64, device="cuda")
    capi.icp_accumulate(T0["Rc2w"], T0["tc2w"], dv[0], dv[1], Rprev_inv, T0["tc2w"], k, dv[2], dv[3], cols * 8, rows, cols, 0.10, angle, ws, ref)
    torch.cuda.synchronize()
    ref = ref.cpu().numpy()
    count = capi.icp_records_count(cols, 0, rows)
    # eight tiles (one per wave) per workgroup at levels 1 / 2; level 0's 4 800 tiles as 256 sixteen-wave workgroups of 18 or 19 (one per CU)
    assert count == {0: 256, 1: 150, 2: 45}[level] and capi.icp_records_bytes() == 768 * 56 * 8
    rec, free = _coherent_host_bytes(capi.icp_records_bytes())
    mailbox, in_dev = capi.icp_mailbox_alloc()
    try:
        got = []
        for seq in (3, 4):
            capi.icp_accumulate_records(T0["Rc2w"], T0["tc2w"], dv[0], dv[1], Rprev_inv, T0["tc2w"], k, dv[2], dv[3], cols * 8, rows, cols, 0.10, angle,
                                        rec, seq)
            rc, sums = capi.icp_sum_records(rec, count, seq)          # waits on the records themselves: no synchronise
            assert rc == 0
            got.append(sums)
        assert np.array_equal(got[0], got[1])
        assert got[0][54] == ref[54] > 1000
        assert np.allclose(got[0], ref, rtol=1e-12, atol=1e-12 * np.abs(ref).max())
        # an older sequence number is not mistaken for the new launch's: nothing was launched with 9, so the wait runs out
        rc, _ = capi.icp_sum_records(rec, count, 9, max_spins=1000)
        assert rc == -1
        # posted pose
        capi.icp_accumulate_records(None, None, dv[0], dv[1], Rprev_inv, T0["tc2w"], k, dv[2], dv[3], cols * 8, rows, cols, 0.10, angle, rec, 11,
                                    mailbox=mailbox, mailbox_seq=77)
        capi.icp_post_pose(mailbox, T0["Rc2w"], T0["tc2w"], 77)
        rc, sums = capi.icp_sum_records(rec, count, 11)
        assert rc == 0 and np.array_equal(sums, got[0])
        # a sharded row range
        y0, y1 = rows // 4, rows // 2
        capi.icp_accumulate_records(T0["Rc2w"], T0["tc2w"], dv[0], dv[1], Rprev_inv, T0["tc2w"], k, dv[2], dv[3], cols * 8, rows, cols, 0.10, angle,
                                    rec, 12, y0=y0, y1=y1)
        rc, part = capi.icp_sum_records(rec, capi.icp_records_count(cols, y0, y1), 12)
        want = torch.zeros(55, dtype=torch.float64, device="cuda")
        capi.icp_accumulate(T0["Rc2w"], T0["tc2w"], dv[0], dv[1], Rprev_inv, T0["tc2w"], k, dv[2], dv[3], cols * 8, rows, cols, 0.10, angle, ws, want,
                            y0=y0, y1=y1)
        torch.cuda.synchronize()
        want = want.cpu().numpy()
        assert rc == 0 and part[54] == want[54] and np.allclose(part, want, rtol=1e-12, atol=1e-12 * np.abs(want).max())
    finally:
        torch.cuda.synchronize()
        free()
        capi.icp_mailbox_free(mailbox, in_dev)


def test_icp_posted_pose_gives_up(dev):
    """A launch whose pose is never posted must not hold the GPU: it leaves after about a second,
    writes nothing, and reports done_seq | 1<<63 through the completion word."""
    import ctypes as C, time
    torch, capi = dev
    nanmap = torch.full((3 * 60, 80, 2), float("nan"), dtype=torch.float32, device="cuda")
    ws = torch.zeros(capi.icp_workspace_bytes(), dtype=torch.uint8, device="cuda")
    host, free = _coherent_host_bytes(256)
    try:
        I = np.zeros((3, 3, 2), np.float32); I[[0, 1, 2], [0, 1, 2], 0] = 1
        sums = torch.full((55,), 7.0, dtype=torch.float64, device="cuda")
        t0 = time.time()
        capi.icp_accumulate_posted(host, 3, nanmap, nanmap, I, np.zeros(6), [100, 100, 40, 30], nanmap, nanmap, 80 * 8, 60, 80, 0.1, 0.2,
                                   ws, sums, done_flag=host + 128, done_seq=41)
        torch.cuda.synchronize()
        assert time.time() - t0 < 20.0
        assert C.c_ulonglong.from_address(host + 128).value == (41 | (1 << 63))
        assert np.all(sums.cpu().numpy() == 7.0)
    finally:
        torch.cuda.synchronize()
        free()


def test_icp_sums_published_as_self_validating_pairs(dev, oracle):
    """XS_ICP_PUBLISH_PAIRS: the 55 sums reach pinned host memory as 55 stores of {sequence number, sum} and the host needs no completion word
    behind them.  The sums equal the plain launch's bit for bit at all three levels (plain and posted launch); a second launch with another
    number replaces every pair; a posted launch whose pose never comes reports number | 1 << 63 in the first pair and touches no other."""
    import ctypes as C
    torch, capi = dev
    prm, T0, pv, pn, cv, cn = icp_inputs(oracle)
    Rprev_inv = oracle.m3_inverse(T0["Rc2w"])
    angle = float(np.sin(np.float32(15.0) / np.float32(180.0) * np.pi))
    ws = torch.zeros(capi.icp_workspace_bytes(), dtype=torch.uint8, device="cuda")
    pairs, free = _coherent_host_bytes(1024)
    mailbox, in_device = capi.icp_mailbox_alloc()
    seq = 100
    try:
        d = oracle.bilateral(synth.s1_frame(1))
        for level in range(3):
            if level:
                pv, pn = oracle.resize_map(pv, False), oracle.resize_map(pn, True)
                d = oracle.pyr_down(d)
            k = intr_of(prm, level)
            cvl = oracle.create_vmap(k, d); cnl = oracle.create_nmap(cvl)
            rows, cols = cvl.shape[0] // 3, cvl.shape[1]
            dev_maps = [to_dev(torch, m) for m in (cvl, cnl, pv, pn)]
            plain = torch.zeros(55, dtype=torch.float64, device="cuda")
            common = (dev_maps[0], dev_maps[1], Rprev_inv, T0["tc2w"], k, dev_maps[2], dev_maps[3], cols * 8, rows, cols, 0.10, angle, ws)
            capi.icp_accumulate(T0["Rc2w"], T0["tc2w"], *common, plain)
            torch.cuda.synchronize()
            want = plain.cpu().numpy()
            assert want[54] > 0.3 * rows * cols
            seq += 1
            capi.icp_accumulate(T0["Rc2w"], T0["tc2w"], *common, pairs, done_flag=capi.ICP_PUBLISH_PAIRS, done_seq=seq)
            rc, got = capi.icp_wait_pairs(pairs, seq)
            assert rc == 0 and np.array_equal(got, want), level
            words = np.ctypeslib.as_array((C.c_ulonglong * 110).from_address(pairs))
            assert np.all(words[0::2] == seq)
            seq += 1
            capi.icp_accumulate_posted(mailbox, seq & 0xffffffff, *common, pairs, done_flag=capi.ICP_PUBLISH_PAIRS, done_seq=seq)
            assert capi.icp_wait_pairs(pairs, seq, max_spins=2000)[0] == 2           # (no pose yet: nothing comes)
            capi.icp_post_pose(mailbox, T0["Rc2w"], T0["tc2w"], seq & 0xffffffff)
            rc, got = capi.icp_wait_pairs(pairs, seq)
            assert rc == 0 and np.array_equal(got, want), level
            torch.cuda.synchronize()
        seq += 1
        capi.icp_accumulate_posted(mailbox, seq & 0xffffffff, *common, pairs, done_flag=capi.ICP_PUBLISH_PAIRS, done_seq=seq)   # never posted
        rc, _ = capi.icp_wait_pairs(pairs, seq)
        assert rc == 1
        torch.cuda.synchronize()
        assert np.all(words[2::2] == seq - 1)                                        # (only the first pair's word was touched)
        capi.icp_workspace_init(ws)
        seq += 1
        capi.icp_accumulate(T0["Rc2w"], T0["tc2w"], *common, pairs, done_flag=capi.ICP_PUBLISH_PAIRS, done_seq=seq)
        rc, got = capi.icp_wait_pairs(pairs, seq)
        assert rc == 0 and np.array_equal(got, want)
    finally:
        torch.cuda.synchronize()
        capi.icp_mailbox_free(mailbox, in_device)
        free()


def test_icp_iterate_singular_system_stops_the_loop(dev):
    """No valid pixel: zero sums, |det| < 1e-15 -> status 1, pose untouched; the next launch returns at
    once (sums buffer not written), as PoseEstimate returns 0 on the host (KinectFusionReconstruction.cpp:203-210)."""
    torch, capi = dev
    nanmap = torch.full((3 * 60, 80, 2), float("nan"), dtype=torch.float32, device="cuda")
    ws = torch.zeros(capi.icp_workspace_bytes(), dtype=torch.uint8, device="cuda")
    pose = torch.zeros(128, dtype=torch.uint8, device="cuda")
    I = np.zeros((3, 3, 2), np.float32); I[[0, 1, 2], [0, 1, 2], 0] = 1
    t0 = np.array([[1, 0], [2, 0], [3, 1e-7]], np.float32)
    sums = torch.ones(55, dtype=torch.float64, device="cuda")
    capi.icp_iterate(I, t0, nanmap, nanmap, I, np.zeros(6), [100, 100, 40, 30], nanmap, nanmap, 80 * 8, 60, 80, 0.1, 0.2, ws, sums, pose)
    torch.cuda.synchronize()
    raw = pose.cpu().numpy()
    assert not sums.cpu().numpy().any()
    assert tuple(raw[96:104].view(np.int32)) == (1, 1)
    assert np.array_equal(raw[:72].view(np.float32).reshape(3, 3, 2), I) and np.array_equal(raw[72:96].view(np.float32).reshape(3, 2), t0)
    sums.fill_(7.0)
    capi.icp_iterate(None, None, nanmap, nanmap, I, np.zeros(6), [100, 100, 40, 30], nanmap, nanmap, 80 * 8, 60, 80, 0.1, 0.2, ws, sums, pose)
    torch.cuda.synchronize()
    assert np.all(sums.cpu().numpy() == 7.0) and tuple(pose.cpu().numpy()[96:104].view(np.int32)) == (1, 1)
    # a launch that reloads the pose clears the status
    good = torch.zeros((3 * 60, 80, 2), dtype=torch.float32, device="cuda")
    capi.icp_iterate(I, t0, good, good, I, np.zeros(6), [100, 100, 40, 30], good, good, 80 * 8, 60, 80, 0.1, 0.2, ws, sums, pose)
    torch.cuda.synchronize()
    assert pose.cpu().numpy()[100:104].view(np.int32)[0] == 1


def test_icp_all_invalid(dev):
    torch, capi = dev
    nanmap = torch.full((3 * 60, 80, 2), float("nan"), dtype=torch.float32, device="cuda")
    ws = torch.zeros(capi.icp_workspace_bytes(), dtype=torch.uint8, device="cuda")
    sums = torch.ones(55, dtype=torch.float64, device="cuda")
    I = np.zeros((3, 3, 2), np.float32); I[[0, 1, 2], [0, 1, 2], 0] = 1
    A, b, inl = capi.estimate_combined(I, np.zeros(6), nanmap, nanmap, I, np.zeros(6), [100, 100, 40, 30], nanmap, nanmap, 80 * 8,
                                       60, 80, 0.1, 0.2, ws, sums)
    assert inl == 0 and not A.any() and not b.any()


@pytest.mark.parametrize("thres,or_equal", [(0.1, False), (0.2, True), (np.sin(np.radians(20.0)), True), (3e-3, False)])
def test_icp_gate_shortcut_decides_like_the_square_root(dev, thres, or_equal):
    """The two rejection tests of the ICP search (ICP.cu:232-241) settled from bounds on Re sqrt z: the same boolean as the
    full complex square root for values on, next to and far from the threshold, inside and outside the short form's cone,
    for negative, zero, huge, infinite and NaN arguments."""
    torch, capi = dev
    rng = np.random.default_rng(11)
    t2 = np.float64(np.float32(thres)) ** 2
    rel = np.concatenate([[0.0], np.logspace(-8, -1, 57)])
    rel = np.concatenate([rel, -rel])
    a = (t2 * (1.0 + rel))[:, None]
    brel = np.concatenate([[0.0], np.logspace(-9, 1, 41)])
    brel = np.concatenate([brel, -brel])[None, :]
    grid = np.stack(np.broadcast_arrays(a, a * brel), -1).reshape(-1, 2)
    # a within a few ulp of the threshold's square, b small: the values the full path must decide
    ulps = np.nextafter(np.float32(t2), np.float32(np.inf)) - np.float32(t2)
    near = np.stack([np.float32(t2) + ulps * rng.integers(-40, 41, 20000).astype(np.float32),
                     (t2 * 10.0 ** rng.uniform(-9, -3, 20000) * rng.choice([-1, 1], 20000)).astype(np.float32)], -1)
    wide = np.stack([10.0 ** rng.uniform(-12, 2, 50000) * rng.choice([-1, 1, 1, 1], 50000),
                     10.0 ** rng.uniform(-14, 2, 50000) * rng.choice([-1, 1], 50000)], -1)
    special = np.array([[0, 0], [-0.0, 0], [0, 1e-9], [-t2, 0], [-t2, 1e-9], [np.inf, 0], [-np.inf, 0], [1, np.inf], [np.nan, 0],
                        [0, np.nan], [1e38, 1e38], [3e38, 3e38], [1e-45, 0], [1e-45, 1e-45], [t2, np.inf]], np.float64)
    z = np.concatenate([grid, near, wide, special]).astype(np.float32)
    zt = torch.from_numpy(np.ascontiguousarray(z)).cuda()
    zc = torch.view_as_complex(zt)
    bad, full = capi.icp_gate_selftest(zc, thres, or_equal)
    assert bad == 0
    # the near-threshold block reaches the square root; values far from the threshold with a >= 0 and small |b| never do
    # (a < 0 or |b| > a leaves the bounds too far apart to decide: the wide block has a share of those)
    assert 20000 <= full < 0.75 * len(z)
    with np.errstate(over="ignore"):   # (the sample holds values up to the exponent limit)
        far = z[(z[:, 0] > 0) & (np.abs(z[:, 1]) < 1e-3 * z[:, 0]) & (np.abs(z[:, 0] / np.float32(t2) - 1) > 1e-2)]
    assert len(far) > 10000
    assert capi.icp_gate_selftest(torch.view_as_complex(torch.from_numpy(np.ascontiguousarray(far)).cuda()), thres, or_equal) == (0, 0)


# ---- dual-complex Hessian / loss -----------------------------------------------------------
def test_tsdf_hessian_and_loss_golden(dev, oracle):
    torch, capi = dev
    gd = load_golden("hessian_s1_n64.npz")
    n = int(gd["n"])
    prm = synth.s1_params(n)
    res = [n, n, n]
    gt, _, _ = build_volume(oracle, prm, n, [0])
    trunc = tranc_dist(prm)
    assert np.float32(trunc) == gd["trunc"]
    ws = torch.zeros(capi.tsdf_reduce_workspace_bytes(), dtype=torch.uint8, device="cuda")
    dgt = to_dev(torch, gt)
    for tag, k in (("a", 1), ("b", 4)):
        ds = oracle.scale_depth(synth.s1_frame(k))
        dds = to_dev(torch, ds)
        out4 = torch.zeros(4, dtype=torch.float64, device="cuda")
        vols = [torch.zeros(n ** 3, dtype=torch.float32, device="cuda") for _ in range(3)] + [torch.zeros(n ** 3, dtype=torch.int32, device="cuda")]
        capi.compute_local_tsdf_hessian(dds, W * 4, H, W, intr_of(prm), res, prm["tsdf_voxel_size"], gd[f"R_{tag}"], gd[f"t_{tag}"], trunc,
                                        dgt, ws, out4, volumes=vols)
        torch.cuda.synchronize()
        want = gd[f"hess_{tag}"]
        got = out4.cpu().numpy()
        assert got[3] == want[3]
        assert abs(got[0] - want[0]) <= 1e-6 * abs(want[0])
        assert abs(got[1] - want[1]) <= 1e-6 * abs(want[1])   # first derivative (CSFD)
        assert abs(got[2] - want[2]) <= 1e-5 * abs(want[2])   # second derivative: differences of O(h^2) float terms
        # per-voxel volumes vs the live oracle
        (oo, ovols) = oracle.tsdf_hessian(ds, res, prm["tsdf_voxel_size"], gd[f"R_{tag}"], gd[f"t_{tag}"], trunc, intr_of(prm), gt, want_volumes=True)
        assert np.array_equal(vols[3].cpu().numpy(), ovols[3])
        assert mismatch_fraction(vols[0].cpu().numpy(), ovols[0]) <= 1e-3
        assert np.allclose(vols[1].cpu().numpy(), ovols[1], rtol=1e-5, atol=1e-6 * np.abs(ovols[1]).max())
        # real-valued twin
        out2 = torch.zeros(2, dtype=torch.float64, device="cuda")
        T = np.linalg.inv(np.eye(4))
        R9 = gd[f"R_{tag}"][..., 0].reshape(9)
        t3 = gd[f"t_{tag}"][..., 0].reshape(3)
        capi.compute_local_tsdf_loss(dds, W * 4, H, W, intr_of(prm), res, prm["tsdf_voxel_size"], R9, t3, trunc, dgt, ws, out2)
        torch.cuda.synchronize()
        wl = oracle.tsdf_loss(ds, res, prm["tsdf_voxel_size"], R9, t3, trunc, intr_of(prm), gt)
        gl = out2.cpu().numpy()
        assert gl[1] == wl[1] and abs(gl[0] - wl[0]) <= 1e-6 * abs(wl[0])


def test_tsdf_hessian_slabs_add_up(dev, oracle):
    torch, capi = dev
    gd = load_golden("hessian_s1_n64.npz")
    n = 64
    prm = synth.s1_params(n)
    gt, _, _ = build_volume(oracle, prm, n, [0])
    ws = torch.zeros(capi.tsdf_reduce_workspace_bytes(), dtype=torch.uint8, device="cuda")
    dds = to_dev(torch, oracle.scale_depth(synth.s1_frame(1)))
    dgt = to_dev(torch, gt)
    def run(z0, z1):
        out4 = torch.zeros(4, dtype=torch.float64, device="cuda")
        capi.compute_local_tsdf_hessian(dds, W * 4, H, W, intr_of(prm), [n, n, n], prm["tsdf_voxel_size"], gd["R_a"], gd["t_a"],
                                        tranc_dist(prm), dgt[z0 * n * n:], ws, out4, z0=z0, z1=z1)
        torch.cuda.synchronize()
        return out4.cpu().numpy()
    whole = run(0, n)
    parts = run(0, 24) + run(24, 40) + run(40, 64)
    assert whole[3] == parts[3] and np.allclose(whole, parts, rtol=1e-12)


def test_tsdf_hessian_full_size_512(dev, oracle):
    """BASELINE config 4 at its own size: xs_compute_local_tsdf_hessian over a 512^3 map (three frames of scene S3 fused by
    the HIP integrate kernel at their ground-truth poses) for the depth frame and pose of frame 3, both first-order seeds
    on t_x.  (1) eight z-slabs add up to the whole-volume pass — the sharded form, count exact; (2) the slab holding most
    of the band equals the CPU oracle on the same planes; (3) on those planes loss, d/dt_x and d2/dt_x^2 equal the
    independent float64 model's value and central differences (tests/independent_f64.py); (4) sanity of the sums: every
    voxel contributes a square, and near the true pose the second derivative is positive."""
    import independent_f64 as ind
    from independent_cases import dual_pose
    torch, capi = dev
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
        capi.integrate_tsdf_volume(to_dev(torch, synth.s3_frame(k)), W * 2, H, W, intr_of(prm), 100, res, prm["tsdf_voxel_size"], T["Rv2c"], T["tv2c"],
                                   tranc_dist(prm), value, weight, grad, n * 4, scaled, W * 4)
    del weight, grad
    d3 = synth.s3_frame(3)
    capi.scale_depth(to_dev(torch, d3), W * 2, H, W, scaled, W * 4)
    torch.cuda.synchronize()
    depth_m = scaled.cpu().numpy()
    Rd, td = dual_pose(prm, 3, 1e-6)
    ws = torch.zeros(capi.tsdf_reduce_workspace_bytes(), dtype=torch.uint8, device="cuda")
    gt_dev = value.reshape(-1)

    def run(z0, z1):
        out4 = torch.zeros(4, dtype=torch.float64, device="cuda")
        capi.compute_local_tsdf_hessian(scaled, W * 4, H, W, intr_of(prm), res, prm["tsdf_voxel_size"], Rd, td, tranc_dist(prm), gt_dev[z0 * n * n:], ws, out4,
                                        z0=z0, z1=z1)
        torch.cuda.synchronize()
        return out4.cpu().numpy()
    whole = run(0, n)
    slabs = [run(n * i // 8, n * (i + 1) // 8) for i in range(8)]
    parts = np.sum(slabs, axis=0)
    assert whole[3] == parts[3] > 100000
    assert np.allclose(whole[:3], parts[:3], rtol=1e-9, atol=0)
    assert whole[0] > 0 and whole[2] > 0 and all(s_[0] >= 0 for s_ in slabs)
    # the 16 planes with the most band voxels, against the oracle and the float64 model
    busiest = int(np.argmax([s_[3] for s_ in slabs]))
    z0 = n * busiest // 8 + 24
    z1 = z0 + 16
    got = run(z0, z1)
    gt_planes = gt_dev[z0 * n * n:z1 * n * n].cpu().numpy()
    want = oracle.tsdf_hessian(depth_m, res, prm["tsdf_voxel_size"], Rd, td, tranc_dist(prm), intr_of(prm), gt_planes, z0=z0, z1=z1)
    assert got[3] == want[3] > 1000
    assert abs(got[0] - want[0]) <= 1e-6 * abs(want[0]) and abs(got[1] - want[1]) <= 1e-5 * max(abs(want[1]), 1e-6 * (want[0] * abs(want[2])) ** 0.5)
    assert abs(got[2] - want[2]) <= 1e-4 * abs(want[2])
    args = (Rd, td, gt_planes.reshape(z1 - z0, n, n), depth_m, intr_of(prm), prm["tsdf_voxel_size"], tranc_dist(prm), z0)
    h2, fd = 1e-6, 2e-4
    l0, c0, dec = ind.tsdf_residual_loss(0.0, h2, *args)
    lp, _, _ = ind.tsdf_residual_loss(+fd, h2, *args, dec=dec)
    lm, _, _ = ind.tsdf_residual_loss(-fd, h2, *args, dec=dec)
    g_model, h_model = (lp - lm) / (2 * fd), (lp - 2 * l0 + lm) / (fd * fd)
    assert abs(got[3] - c0) <= max(2, 1e-4 * c0)
    assert abs(got[0] - l0) <= 5e-4 * l0
    assert abs(got[1] / h2 - g_model) <= 5e-4 * max(abs(g_model), (l0 * abs(h_model)) ** 0.5)
    assert abs(got[2] / h2 / h2 - h_model) <= 5e-4 * abs(h_model)


# ---- DeviceArray scalar math on the device ------------------------------------------------
CSFD_EXACT = ("add", "sub", "mul", "div", "div_scalar", "scalar_div", "mul_scalar", "scalar_sub")


def _table(torch, capi, dual, op, a, b):
    da, db = to_dev(torch, a), to_dev(torch, b)
    out = torch.zeros_like(da)
    capi.complex_table(dual, op, da, db, out, a.shape[0])
    torch.cuda.synchronize()
    return out.cpu().numpy()


def test_device_complex_tables_vs_reference_header(dev, oracle):
    """csrc/xs_complex.h on the GPU against tables produced by the reference's own
    cuda_complex.hpp (tests/golden/scalar_tables.npz)."""
    torch, capi = dev
    t = load_golden("scalar_tables.npz")
    from oracle.oracle import Oracle
    for tag in ("csfd", "gen", "wide"):
        for op in CSFD_EXACT:
            got = _table(torch, capi, 0, Oracle.COP[op], t[f"{tag}_a"], t[f"{tag}_b"])
            # IEEE + - * / without contraction: bit-exact, also through the divide's scaling shortcut
            assert ulp_diff(got, t[f"c_{tag}_{op}"]).max() == 0, (tag, op)
    # sqrt in the CSFD cone (the only regime the kernels use): bit-exact short form
    for tag in ("csfd", "pos"):
        a = t[f"{tag}_a"].copy()
        a[:, 0] = np.abs(a[:, 0]) + 1e-3
        want = oracle.cop("sqrt", a)
        assert ulp_diff(_table(torch, capi, 0, Oracle.COP["sqrt"], a, a), want).max() == 0
    # general operands go through the device libm: a few ulp
    for op in ("sqrt", "abs", "exp", "sin", "cos", "sinh", "cosh", "sin_new", "sinh_new", "norm", "arg", "conj"):
        got = _table(torch, capi, 0, Oracle.COP[op], t["gen_a"], t["gen_a"])
        want = t[f"c_gen_{op}"]
        assert np.allclose(got, want, rtol=4e-6, atol=4e-7), op
    got = _table(torch, capi, 0, Oracle.COP["log"], t["pos_a"], t["pos_a"])
    assert np.allclose(got, t["c_pos_log"], rtol=2e-6, atol=1e-12)
    # the rest of the header (proj log10 tanh tan asinh acosh atanh asin acos atan): device libm, a few ulp on the
    # general operands; on the special-value cross product the same NaN / inf / signed-zero pattern
    for op in ("proj", "log10", "tanh", "tan", "asinh", "acosh", "atanh", "asin", "acos", "atan"):
        for tag, key in (("ext", "ext_a"), ("csfdx", "csfd_a")):
            got = _table(torch, capi, 0, Oracle.COP[op], t[key], t[key])
            want = t[f"c_{tag}_{op}"]
            # absolute: these formulas go through pow(z, 2) = exp(2 log z) and log(z + sqrt(...)), whose own rounding is of the order
            # of an ulp of the intermediate (e.g. 2 arg z near 2 pi for Re z < 0) — the device's libm lands elsewhere inside that
            # error; the bit-exact pin of the header is the host test (tests/test_abi_cpu.py)
            assert np.allclose(got, want, rtol=2e-5, atol=4e-6), (op, tag, np.abs(got - want).max())
        got = _table(torch, capi, 0, Oracle.COP[op], t["spec_a"], t["spec_a"])
        want = t[f"c_spec_{op}"]
        assert np.array_equal(np.isnan(got), np.isnan(want)), op
        fin = np.isfinite(want)
        assert np.array_equal(np.isinf(got), np.isinf(want)) and np.array_equal(np.signbit(got[~np.isnan(want)]), np.signbit(want[~np.isnan(want)])), op
        assert np.allclose(got[fin], want[fin], rtol=2e-5, atol=1e-6), op
    # dual complex
    for op in ("add", "sub", "mul", "div", "mul_scalar", "div_scalar", "add_scalar", "scalar_sub"):
        got = _table(torch, capi, 1, Oracle.DOP[op], t["d_a"], t["d_b"])
        assert ulp_diff(got, t[f"d_{op}"]).max() == 0, op
    for op in ("sqrt", "abs"):
        got = _table(torch, capi, 1, Oracle.DOP[op], t["d_pos"], t["d_pos"])
        assert ulp_diff(got, t[f"d_{op}"]).max() == 0, op


def test_csfd_array_ops_config1(dev, oracle):
    """BASELINE config 1: 1e6-element complex arrays, real ~ U(-2, 2), imag = 1e-6, through the
    five raw / our kernels; CSFD derivative = Im / h."""
    torch, capi = dev
    n = 1_000_000
    rng = np.random.default_rng(1)
    a = np.stack([rng.uniform(-2, 2, n), np.full(n, 1e-6)], -1).astype(np.float32)
    b = np.stack([rng.uniform(-2, 2, n), np.full(n, 1e-6)], -1).astype(np.float32)
    b[np.abs(b[:, 0]) < 0.05, 0] = 0.05
    da, db = to_dev(torch, a), to_dev(torch, b)
    out = torch.zeros_like(da)
    for name in ("mul", "div", "exp", "sin", "pow"):
        for variant in ("raw", "our"):
            capi.csfd_array_op(name, variant, da, db, out, n)
            torch.cuda.synchronize()
            got = out.cpu().numpy()
            want = oracle.csfd_op(name, variant, a, b)
            if name in ("mul", "div"):
                assert ulp_diff(got, want).max() == 0, (name, variant)
            elif name == "pow":
                # |z|^3 * sin/cos(3 * arg z): one ulp of atan2f near pi moves the imaginary part by
                # ~3e-7 * |z|^3 (the demo itself prints 5.6982e-06 where the exact value is 6e-06)
                # (both forms scale the sine / cosine by norm(z)^3 = |z|^6, main.cpp:74-86)
                sre, sim = a[:, 0] + b[:, 0], a[:, 1] + b[:, 1]
                mag = np.maximum(np.abs(want[:, :1]) + np.abs(want[:, 1:]), ((sre * sre + sim * sim) ** 3)[:, None])
                assert np.all(np.abs(got - want) <= 5e-6 * np.abs(want) + 2e-6 * mag), (name, variant)
                if variant == "our":   # Re = float(pow(double(re), 3)): the device forms the double cube by multiplication (2 roundings of 2^-53)
                    d = ulp_diff(got[:, 0], want[:, 0])
                    assert d.max() <= 1 and np.count_nonzero(d) <= 2, (d.max(), np.count_nonzero(d))
            else:
                assert np.allclose(got, want, rtol=5e-6, atol=5e-6 * np.abs(want).max(axis=0)), (name, variant)
    # odd length + dual-complex f1
    capi.csfd_array_op("mul", "raw", da, db, out, 12345)
    torch.cuda.synchronize()
    assert ulp_diff(out.cpu().numpy()[:12345], oracle.csfd_op("mul", "raw", a[:12345], b[:12345])).max() == 0
    # ragged lengths around the kernel's pieces (256 lanes x 2 elements x U pieces per round), nothing written past the end
    for m in (1, 2, 3, 511, 512, 513, 1023, 1024, 1025, 2047, 2049, 4095, 4097, 8191, 8193, 99_999):
        out.fill_(-7.0)
        capi.csfd_array_op("div", "our", da, db, out, m)
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        assert ulp_diff(got[:m], oracle.csfd_op("div", "our", a[:m], b[:m])).max() == 0 and np.all(got[m:m + 4096] == -7.0), m
    # an array that takes the nontemporal, four-pieces-per-round instance (more than 128 MB per launch), odd length
    big = 6_000_001
    rb = np.random.default_rng(2)
    ab = np.stack([rb.uniform(-2, 2, big), np.full(big, 1e-6)], -1).astype(np.float32)
    bb = np.stack([rb.uniform(0.05, 2, big), np.full(big, 1e-6)], -1).astype(np.float32)
    dab, dbb = to_dev(torch, ab), to_dev(torch, bb)
    ob = torch.full((big + 1024, 2), -7.0, dtype=torch.float32, device="cuda")
    for name in ("mul", "div"):
        for variant in ("raw", "our"):
            capi.csfd_array_op(name, variant, dab, dbb, ob, big)
            torch.cuda.synchronize()
            got = ob.cpu().numpy()
            assert ulp_diff(got[:big], oracle.csfd_op(name, variant, ab, bb)).max() == 0 and np.all(got[big:] == -7.0), (name, variant)
    del dab, dbb, ob
    t = load_golden("scalar_tables.npz")
    dx, dy = to_dev(torch, t["d_a"]), to_dev(torch, t["d_b"])
    o = torch.zeros_like(dx)
    capi.dcsfd_f1(dx, dy, o, t["d_a"].shape[0])
    torch.cuda.synchronize()
    assert ulp_diff(o.cpu().numpy(), oracle.hdop("f1", t["d_a"], t["d_b"])).max() <= 1
