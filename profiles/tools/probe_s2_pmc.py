"""GPU box, repository root: the S2 integrate launches in a fixed order for the counter passes (collect_pmc_r4.sh) — launch 0 into a freshly
initialised volume (the first touch), then 5 launches of the same frame (the steady state of a static camera).  Prints U."""
import importlib, json, os, sys
sys.path.insert(0, '.')
import numpy as np, torch
capi = importlib.import_module('x-slam_amd.capi'); synth = importlib.import_module('x-slam_amd.synth')
H, W, n = synth.HEIGHT, synth.WIDTH, 512
prm = synth.s2_params(n); res = [n, n, n]; vs = float(np.float32(prm["tsdf_voxel_size"])); trunc = synth.tranc_dist(prm)
value = torch.empty((n * n, n), dtype=torch.float32, device="cuda"); weight = torch.empty((n * n, n), dtype=torch.int32, device="cuda")
grad = torch.empty((n * n, n), dtype=torch.float32, device="cuda")
capi.init_volume(value, weight, grad, n * 4, res)
frame = synth.render_s2()
if os.environ.get("S2_NOISY"):      # SURVEY 8(d)'s +-2 mm noise + holes / out-of-range patches + 0.2 % speckle (bench.py roofline_s2.noisy)
    frame = synth.holed(synth.render_s2(noise_mm=2.0), np.random.default_rng(0xC5FD), n_holes=40, speckle=0.002)
depth = torch.from_numpy(frame.view(np.int16)).cuda()
scaled = torch.empty((H, W), dtype=torch.float32, device="cuda"); dmax = torch.zeros(1, dtype=torch.float32, device="cuda")
capi.scale_depth_max(depth, W * 2, H, W, scaled, W * 4, dmax)
ws = torch.zeros(capi.integrate_workspace_bytes(res), dtype=torch.uint8, device="cuda")
R = np.zeros((3, 3, 2), np.float32); R[[0, 1, 2], [0, 1, 2], 0] = 1
t = np.zeros((3, 2), np.float32); t[:, 0] = [-prm["init_x"], -prm["init_y"], -prm["init_z"]]; t[0, 1] = 1e-7
intr = np.array([synth.FX, synth.FY, synth.CX, synth.CY], np.float32)
counter = torch.zeros(1, dtype=torch.int64, device="cuda")
for k in range(6):
    counter.zero_()
    capi.integrate_scaled(scaled, W * 4, H, W, intr, 100, res, vs, R, t, trunc, value, weight, grad, n * 4, updated=counter, depth_max=dmax, workspace=ws)
    torch.cuda.synchronize()
print(json.dumps({"U": int(counter.item())}))
