import importlib, sys, json
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
capi = importlib.import_module('x-slam_amd.capi'); synth = importlib.import_module('x-slam_amd.synth')
from helpers import intr_of, tranc_dist
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
prm = synth.s1_params(n); res = [n, n, n]; W, H = 640, 480
gt = torch.zeros(n ** 3, dtype=torch.float32, device="cuda")
scaled = torch.full((H, W), 2.0, dtype=torch.float32, device="cuda")
Rs = np.zeros((6, 3, 3, 2), np.float32); ts = np.zeros((6, 3, 2), np.float32)
for k in range(6):
    Rs[k, [0, 1, 2], [0, 1, 2], 0] = 1; ts[k, :, 0] = [-3.2, -3.2, -3.2]
ws = torch.zeros(capi.tsdf_reduce_workspace_bytes(), dtype=torch.uint8, device="cuda")
out = torch.zeros(32, dtype=torch.float64, device="cuda")
s = torch.cuda.current_stream()
def run(): capi.tsdf_gauss_newton_terms(scaled, W * 4, H, W, intr_of(prm), res, prm["tsdf_voxel_size"], Rs, ts, tranc_dist(prm), gt, ws, out, stream=s)
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(s)
for _ in range(10): run()
e1.record(s); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(json.dumps({"n": n, "gn_all_zero_gt_ms": round(ms, 4), "GBs": round(4 * n ** 3 / ms / 1e6, 1)}))
o4 = torch.zeros(4, dtype=torch.float64, device="cuda"); o2 = torch.zeros(2, dtype=torch.float64, device="cuda")
Rd = np.zeros((3, 3, 4), np.float32); Rd[[0, 1, 2], [0, 1, 2], 0] = 1; td = np.zeros((3, 4), np.float32); td[:, 0] = -3.2
for name, fn in (("hessian", lambda: capi.compute_local_tsdf_hessian(scaled, W * 4, H, W, intr_of(prm), res, prm["tsdf_voxel_size"], Rd, td, tranc_dist(prm), gt, ws, o4, stream=s)),
                 ("loss", lambda: capi.compute_local_tsdf_loss(scaled, W * 4, H, W, intr_of(prm), res, prm["tsdf_voxel_size"], Rd[..., 0], td[..., 0], tranc_dist(prm), gt, ws, o2, stream=s))):
    fn(); torch.cuda.synchronize()
    e0.record(s)
    for _ in range(10): fn()
    e1.record(s); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(json.dumps({"n": n, name + "_all_zero_gt_ms": round(ms, 4), "GBs": round(4 * n ** 3 / ms / 1e6, 1)}))
