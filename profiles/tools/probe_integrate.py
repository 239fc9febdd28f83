"""The integrate kernel of scene S1 at 512^3 on its own (GPU box, repository root): bricks listed, voxels written, kernel time
(HIP events on the launch packet), for the brick depths XS_BRICK_Z selects.  python profiles/tools/probe_integrate.py [frames]"""
import ctypes as C, importlib, os, subprocess, sys
sys.path.insert(0, '.')
if len(sys.argv) > 1 and sys.argv[1] == "sweep":
    for bz in ("0", "2", "4", "8", "16", "32"):
        env = dict(os.environ, XS_BRICK_Z=bz)
        print("XS_BRICK_Z=" + bz, subprocess.run([sys.executable, __file__], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1])
    sys.exit(0)
import numpy as np, torch
capi = importlib.import_module('x-slam_amd.capi'); synth = importlib.import_module('x-slam_amd.synth')
H, W, n = synth.HEIGHT, synth.WIDTH, int(os.environ.get('XS_PROBE_N', '512'))
prm = synth.s1_params(n); res = [n, n, n]; vs = float(np.float32(prm["tsdf_voxel_size"])); trunc = synth.tranc_dist(prm)
value = torch.empty((n * n, n), dtype=torch.float32, device="cuda"); weight = torch.empty((n * n, n), dtype=torch.int32, device="cuda")
grad = torch.empty((n * n, n), dtype=torch.float32, device="cuda")
capi.init_volume(value, weight, grad, n * 4, res)
scaled = torch.empty((H, W), dtype=torch.float32, device="cuda"); dmax = torch.zeros(1, dtype=torch.float32, device="cuda")
ws = torch.zeros(capi.integrate_workspace_bytes(res), dtype=torch.uint8, device="cuda")
intr = np.array([synth.FX, synth.FY, synth.CX, synth.CY], np.float32)
counter = torch.zeros(1, dtype=torch.int64, device="cuda")
hip = C.CDLL("libamdhip64.so"); ev = [C.c_void_p(), C.c_void_p()]
for e in ev:
    assert hip.hipEventCreate(C.byref(e)) == 0
s = torch.cuda.current_stream()
times, bricks, Us = [], [], []
for k in range(24):
    depth = torch.from_numpy(synth.s1_frame(k).view(np.int16)).cuda()
    capi.scale_depth_max(depth, W * 2, H, W, scaled, W * 4, dmax)
    T = synth.s1_transforms(k, prm)
    counter.zero_()
    capi.integrate_scaled_ex(scaled, W * 4, H, W, intr, 100, res, vs, T["Rv2c"], T["tv2c"], trunc, value, weight, grad, n * 4, 0 if os.environ.get("XS_PROBE_NO_COUNT") else 64, start_event=ev[0], stop_event=ev[1], updated=counter, depth_max=dmax, workspace=ws, stream=s)
    torch.cuda.synchronize()
    classes = [int(x) for x in ws[192:204].view(torch.int32).cpu().numpy()]
    dt = C.c_float(0); assert hip.hipEventElapsedTime(C.byref(dt), ev[0], ev[1]) == 0
    if k >= 4:
        times.append(dt.value * 1e3); bricks.append(sum(capi.integrate_listed(ws))); Us.append(int(counter.item()))
b, U, t = np.median(bricks), np.median(Us), np.median(times)
print(f"bricks listed {b:.0f}  voxels written {U:.0f}  kernel {t:.1f} us  ->  {24 * U / t / 1e6:.2f} TB/s algorithmic; boxes free / nothing / walk (last frame) {classes}")
