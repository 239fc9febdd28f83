import importlib, sys, json
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')  # run from the repository root
import numpy as np, torch
capi = importlib.import_module('x-slam_amd.capi'); synth = importlib.import_module('x-slam_amd.synth'); pl = importlib.import_module('x-slam_amd.pipeline')
from helpers import intr_of, s1_transforms, tranc_dist
n = 512; prm = synth.s1_params(n); res = [n, n, n]; W, H = 640, 480
kf = pl.KinectFusion(prm)
frames = [torch.from_numpy(synth.s1_frame(k).view(np.int16)).cuda() for k in range(4)]
for f in frames: assert kf.process_frame(f) == 1
v, w, g = kf.volume()
gt = torch.from_numpy(v).cuda()
T = s1_transforms(3, prm)
scaled = torch.empty((H, W), dtype=torch.float32, device="cuda")
capi.scale_depth(frames[3], W * 2, H, W, scaled, W * 4)
Rd = np.zeros((3, 3, 4), np.float32); td = np.zeros((3, 4), np.float32)
Rd[..., 0] = np.asarray(T["Rv2c"])[..., 0]; td[..., 0] = np.asarray(T["tv2c"])[..., 0]; td[0, 1] = 1e-6; td[0, 2] = 1e-6
ws = torch.zeros(capi.tsdf_reduce_workspace_bytes(), dtype=torch.uint8, device="cuda")
out4 = torch.zeros(4, dtype=torch.float64, device="cuda"); out2 = torch.zeros(2, dtype=torch.float64, device="cuda")
s = torch.cuda.current_stream()
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(reps): fn()
    e1.record(s); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
th = timeit(lambda: capi.compute_local_tsdf_hessian(scaled, W * 4, H, W, intr_of(prm), res, prm["tsdf_voxel_size"], Rd, td, tranc_dist(prm), gt, ws, out4, stream=s))
tl = timeit(lambda: capi.compute_local_tsdf_loss(scaled, W * 4, H, W, intr_of(prm), res, prm["tsdf_voxel_size"], Rd[..., 0], td[..., 0], tranc_dist(prm), gt, ws, out2, stream=s))
print(json.dumps({"hessian_ms": round(th, 4), "loss_ms": round(tl, 4), "hess_GBs": round(4 * n**3 / th / 1e6, 1), "loss_GBs": round(4 * n**3 / tl / 1e6, 1), "count": out4.cpu().numpy()[3]}))
gz = torch.zeros_like(gt)
tz = timeit(lambda: capi.compute_local_tsdf_loss(scaled, W * 4, H, W, intr_of(prm), res, prm["tsdf_voxel_size"], Rd[..., 0], td[..., 0], tranc_dist(prm), gz, ws, out2, stream=s))
print(json.dumps({"loss_ms_all_zero_gt": round(tz, 4), "GBs": round(4 * n**3 / tz / 1e6, 1)}))
