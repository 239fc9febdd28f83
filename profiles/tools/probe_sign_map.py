"""GPU box: the march kernel alone on a tracked 512^3 volume (scene S1, 40 frames), every step against the sign-map start (bricks of 8^3 and
16^3 voxels): event time over 50 launches of the march + crossing pair, and of the march alone (rocprofv3 --kernel-trace gives the split)."""
import importlib, sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
pl = importlib.import_module("x-slam_amd.pipeline")
capi = importlib.import_module("x-slam_amd.capi")
synth = importlib.import_module("x-slam_amd.synth")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
H, W = synth.HEIGHT, synth.WIDTH
prm = synth.s1_params(N)
kf = pl.KinectFusion(prm)
for k in range(40):
    assert kf.process_frame(torch.from_numpy(synth.s1_frame(k).view(np.int16)).cuda()) == 1
kf.synchronize()
if os.environ.get("PROBE_STAGE"):
    os.environ["XS_RAY_STEPS_EVALUATED"] = os.environ["PROBE_STAGE"]
w2c = kf.world2camera().astype(np.float64)
w2c = w2c[..., 0] + 1j * w2c[..., 1]
c2w = np.linalg.inv(w2c)
w2v = np.eye(4, dtype=np.complex128)
w2v[:3, 3] = [prm["init_x"], prm["init_y"], prm["init_z"]]
c2v, v2w = w2v @ c2w, np.linalg.inv(w2v)
f = lambda m: synth.cmat(m.real, m.imag)
pv, step = kf.volume_ptr("value")
pg, _ = kf.volume_ptr("grad")
vm = torch.empty((3 * H, W, 2), dtype=torch.float32, device="cuda")
nm = torch.empty_like(vm)
ws = torch.zeros(H * W, dtype=torch.float32, device="cuda")
steps = torch.zeros(H * W, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream()
td = kf.tranc_dist()
args = (synth.intr_of(prm), f(c2v[:3, :3]), f(c2v[:3, 3]), f(v2w[:3, :3]), f(v2w[:3, 3]), td, [N, N, N], prm["tsdf_voxel_size"], pv, pg, step,
        vm, nm, W * 8, H, W)
ref = None
for shift in (0, 3, 4, 5):
    sm = None
    if shift:
        sm = torch.zeros(capi.signmap_bytes([N, N, N], shift), dtype=torch.uint8, device="cuda")
        capi.signmap_rebuild(sm, [N, N, N], shift, td, pv, step, stream=s)
    capi.raycast(*args, workspace=ws, stream=s, signmap=sm, signmap_shift=shift or 3, signmap_tranc_dist=td, steps=steps)
    torch.cuda.synchronize()
    out = (ws.cpu().numpy().copy(), steps.cpu().numpy().copy())
    if ref is None:
        ref = out
    same = all(np.array_equal(a.view(np.int32), b.view(np.int32)) for a, b in zip(out, ref))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    o = capi.raycast_opts(signmap=sm, shift=shift or 3, tranc_dist=td)
    for _ in range(50):
        capi.raycast_ex(*args, o, workspace=ws, stream=s)
    e1.record(s)
    torch.cuda.synchronize()
    if os.environ.get("PROBE_STAGE") == "1":
        st = out[1]
        print("   evaluated steps per ray: percentiles 10/25/50/75/90/99/max", [int(v) for v in np.percentile(st, [10, 25, 50, 75, 90, 99, 100])],
              " rays needing > 16:", round(float((st > 16).mean()), 3), " > 24:", round(float((st > 24).mean()), 3), flush=True)
        tiles = st.reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64)
        print("   per 8x8 tile: max evaluated: mean", round(float(tiles.max(1).mean()), 1), " tiles needing > 16:", round(float((tiles.max(1) > 16).mean()), 3),
              " > 24:", round(float((tiles.max(1) > 24).mean()), 3), " > 32:", round(float((tiles.max(1) > 32).mean()), 3), flush=True)
    print(f"shift {shift}: march + crossing {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us   mean steps/ray {out[1].mean():.1f}   same bits {same}", flush=True)
