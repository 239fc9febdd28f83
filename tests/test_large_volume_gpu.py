"""Volumes at and past the sizes BASELINE names: 1024^3 (north_star's 8-GPU volume; 4 GiB per array, exactly what 32-bit byte offsets
reach) and 1024 x 1024 x 1040 (past 4 GiB per array: the raycast's 64-bit-offset kernel instances).  No oracle finishes at these
sizes in seconds, so each path is held bit for bit against another path of the library that the oracle tests pin at small sizes:
the box classes against the per-voxel walk, the 64-bit raycast instances against a composite of 32-bit slab marches."""
import numpy as np
import pytest

from helpers import synth

pytestmark = pytest.mark.gpu
H, W = synth.HEIGHT, synth.WIDTH
NO_TILES, COUNT_CLASSES = 32, 64


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available()
    return torch, __import__("importlib").import_module("x-slam_amd.capi")


def looking_up_z(x, y, z, h=1e-7):
    """volume-to-camera pose of a camera at (x, y, z) of the volume frame looking along +z; CSFD seed on t_x."""
    R = np.zeros((3, 3, 2), np.float32); R[[0, 1, 2], [0, 1, 2], 0] = 1
    t = np.zeros((3, 2), np.float32); t[:, 0] = [-x, -y, -z]; t[0, 1] = h
    return R, t


def new_volume(torch, capi, res):
    X, Y, Z = res
    v = torch.empty((Y * Z, X), dtype=torch.float32, device="cuda"); w = torch.empty((Y * Z, X), dtype=torch.int32, device="cuda")
    g = torch.empty((Y * Z, X), dtype=torch.float32, device="cuda")
    capi.init_volume(v, w, g, X * 4, res)
    return v, w, g


def test_classes_against_the_walk_at_1024_cubed(dev):
    """Scene S2 at 1024^3 (the frustum fills the volume: 180 K listed bricks, 0.7 M boxes), a clean frame and a sensor-like one (2 mm
    noise, 40 holes / out-of-range patches, 0.2 % speckle: EDGE and SPECKLE planes by the million), classes decided with the launch's
    own pose and decided ahead: volume and count against the per-voxel walk of every listed brick (XS_INTEGRATE_NO_TILES), bit for bit."""
    torch, capi = dev
    n = 1024
    prm = synth.s2_params(n)
    res = [n, n, n]
    vs, trunc = float(np.float32(prm["tsdf_voxel_size"])), synth.tranc_dist(prm)
    k4 = np.array([synth.FX, synth.FY, synth.CX, synth.CY], np.float32)
    R, t = looking_up_z(prm["init_x"], prm["init_y"], prm["init_z"])
    rng = np.random.default_rng(1024)
    frames = [synth.render_s2(), synth.holed(synth.render_s2(noise_mm=2.0, frame=1), rng)]
    scaled = torch.empty((H, W), dtype=torch.float32, device="cuda")
    dmax = torch.zeros(1, dtype=torch.float32, device="cuda")
    tiles = torch.zeros(capi.depth_tiles_bytes(H, W), dtype=torch.uint8, device="cuda")
    ws = torch.zeros(capi.integrate_workspace_bytes(res), dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    ref, vol = new_volume(torch, capi, res), new_volume(torch, capi, res)
    streamed = np.zeros(8, np.int64)
    for i, d in enumerate(frames):
        capi.scale_depth_tiles(torch.from_numpy(d.view(np.int16)).cuda(), W * 2, H, W, scaled, W * 4, dmax, tiles)
        args = lambda v: (scaled, W * 4, H, W, k4, 100, res, vs, R, t, trunc, v[0], v[1], v[2], n * 4)
        cnt.zero_()
        capi.integrate_scaled_ex2(*args(ref), capi.integrate_opts(flags=NO_TILES), updated=cnt, depth_max=dmax, workspace=ws)
        U = int(cnt.item())
        assert U > 200_000_000, U
        cnt.zero_()
        flags = COUNT_CLASSES
        if i == 1:   # decided ahead, as the orchestrator does behind the last ICP launch
            capi.integrate_classify_ex(H, W, k4, res, vs, R, t, trunc, ws, capi.integrate_opts(flags=COUNT_CLASSES, depth_tiles=tiles), slack_scale=2.0, depth_max=dmax)
            flags |= 4 | 1
        capi.integrate_scaled_ex2(*args(vol), capi.integrate_opts(flags=flags, depth_tiles=tiles), updated=cnt, depth_max=dmax, workspace=ws)
        assert int(cnt.item()) == U
        streamed += ws[192:224].view(torch.int32).cpu().numpy()
        for a, b, what in zip(ref, vol, ("value", "weight", "grad")):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (i, what)
    assert streamed[0] > 100_000 and streamed[6] > 10_000 and streamed[7] > 10_000, streamed


def test_volume_past_4_gib_per_array_takes_the_64_bit_raycast_instances(dev):
    """A 1024 x 1024 x 1040 volume (4.06 GiB per array: byte offsets leave 32 bits) with the surface in its LAST planes (a wall at plane
    1030, i.e. at byte offsets past 4 GiB): two frames integrated (classes against the walk bit for bit), then every raycast form that takes
    a 64-bit-offset instance on it — plain (k_raycast<0, false, false>), march + crossing with a workspace (<2> + <3>), the sign-map march
    (<0, false, false, true>), a slab march over all planes (<4> + <5>) — against the composite of two slab marches of 526 planes each,
    which fit 32 bits: the same vertex and normal maps, bit for bit."""
    torch, capi = dev
    res = [1024, 1024, 1040]
    X, Y, Z = res
    vs = 0.0045
    trunc = 3.0 * vs
    k4 = np.array([synth.FX, synth.FY, synth.CX, synth.CY], np.float32)
    eye = (X * vs / 2, Y * vs / 2, 0.05)
    R, t = looking_up_z(*eye)
    wall_mm = 1030.5 * vs * 1000.0 - 50.0
    rng = np.random.default_rng(1040)
    scaled = torch.empty((H, W), dtype=torch.float32, device="cuda")
    dmax = torch.zeros(1, dtype=torch.float32, device="cuda")
    tiles = torch.zeros(capi.depth_tiles_bytes(H, W), dtype=torch.uint8, device="cuda")
    ws = torch.zeros(capi.integrate_workspace_bytes(res), dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    ref, vol = new_volume(torch, capi, res), new_volume(torch, capi, res)
    for i in range(2):
        d = np.clip(np.rint(wall_mm + 2.0 * (rng.random((H, W)) * 2 - 1)), 0, 65535).astype(np.uint16)
        if i:
            d = synth.holed(d, rng, n_holes=20)
        capi.scale_depth_tiles(torch.from_numpy(d.view(np.int16)).cuda(), W * 2, H, W, scaled, W * 4, dmax, tiles)
        args = lambda v: (scaled, W * 4, H, W, k4, 100, res, vs, R, t, trunc, v[0], v[1], v[2], X * 4)
        cnt.zero_()
        capi.integrate_scaled_ex2(*args(ref), capi.integrate_opts(flags=NO_TILES), updated=cnt, depth_max=dmax, workspace=ws)
        U = int(cnt.item())
        cnt.zero_()
        capi.integrate_scaled_ex2(*args(vol), capi.integrate_opts(flags=0, depth_tiles=tiles), updated=cnt, depth_max=dmax, workspace=ws)
        assert int(cnt.item()) == U and U > 100_000_000
    for a, b in zip(ref, vol):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    value, _, grad = vol
    del ref
    assert (value[1028 * Y:1033 * Y] < 0).any(), "the wall's negative side must lie past plane 1024"

    # camera-to-volume = the inverse of (R, t): identity rotation, translation = eye; volume-to-world = identity
    I = np.zeros((3, 3, 2), np.float32); I[[0, 1, 2], [0, 1, 2], 0] = 1
    tc2v = np.zeros((3, 2), np.float32); tc2v[:, 0] = eye; tc2v[0, 1] = 1e-7
    zero = np.zeros((3, 2), np.float32)
    maps = lambda: (torch.full((3 * H, W, 2), 5.0, dtype=torch.float32, device="cuda"), torch.full((3 * H, W, 2), 5.0, dtype=torch.float32, device="cuda"))
    common = (k4, I, tc2v, I, zero, trunc, res, vs, value, grad, X * 4)

    def slab(zs0, zs1, z0, z1):
        vm, nm = maps()
        keys = torch.zeros(H * W, dtype=torch.int32, device="cuda")
        off = zs0 * Y
        capi.raycast_slab(k4, I, tc2v, I, zero, trunc, res, vs, value[off:], grad[off:], X * 4, zs0, zs1, z0, z1, vm, nm, W * 8, H, W, keys)
        return vm, nm, keys

    def compose(parts):
        min_keys = parts[0][2].clone()
        for p in parts[1:]:
            min_keys = torch.minimum(min_keys, p[2])
        vsum = torch.zeros((3 * H, W, 2), dtype=torch.int32, device="cuda"); nsum = torch.zeros_like(vsum)
        for vm, nm, keys in parts:
            capi.raycast_compose_mask(keys, min_keys, vm, nm, W * 8, H, W)
            vsum += vm.view(torch.int32); nsum += nm.view(torch.int32)
        vm, nm = vsum.view(torch.float32), nsum.view(torch.float32)
        hits = torch.zeros(1, dtype=torch.int64, device="cuda")
        capi.raycast_compose_finish(min_keys, vm, nm, W * 8, H, W, hits=hits)
        return vm, nm, int(hits.item())

    want_v, want_n, want_hits = compose([slab(0, 526, 0, 520), slab(514, Z, 520, Z)])    # 526 planes x 4 MiB: 32-bit offsets
    assert want_hits > 0.3 * H * W    # (the march ends 5 m along the ray, RayCaster.cu:222: a disc of ~210 px radius reaches the wall)
    valid = ~torch.isnan(want_v[:H, :, 0])
    ok3 = valid.repeat(3, 1)

    def same(vm, nm, what):
        assert torch.equal(torch.isnan(vm[:H, :, 0]), ~valid), what
        assert torch.equal(vm.view(torch.int32)[ok3], want_v.view(torch.int32)[ok3]), what
        assert torch.equal(nm.view(torch.int32)[ok3], want_n.view(torch.int32)[ok3]), what

    hits = torch.zeros(1, dtype=torch.int64, device="cuda")
    vm, nm = maps()
    capi.raycast(*common, vm, nm, W * 8, H, W, hits=hits)
    same(vm, nm, "plain")
    assert int(hits.item()) == want_hits
    vm, nm = maps()
    hits.zero_()
    rws = torch.zeros(H * W, dtype=torch.float32, device="cuda")
    capi.raycast(*common, vm, nm, W * 8, H, W, hits=hits, workspace=rws)
    same(vm, nm, "march + crossing")
    assert int(hits.item()) == want_hits
    shift = capi.raycast_signmap_shift(k4, vs, trunc)
    if shift:
        sm = torch.zeros(capi.signmap_bytes(res, shift), dtype=torch.uint8, device="cuda")
        capi.signmap_rebuild(sm, res, shift, trunc, value, X * 4)
        vm, nm = maps()
        hits.zero_()
        capi.raycast_ex(*common, vm, nm, W * 8, H, W, capi.raycast_opts(signmap=sm, shift=shift, tranc_dist=trunc), hits=hits, workspace=rws)
        same(vm, nm, "sign-map march")
        assert int(hits.item()) == want_hits
    whole = compose([slab(0, Z, 0, Z)])
    same(whole[0], whole[1], "slab march over all planes")
    assert whole[2] == want_hits
