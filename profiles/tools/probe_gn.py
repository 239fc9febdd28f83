import importlib, sys, json
sys.path.insert(0, '.')
import numpy as np, torch
synth = importlib.import_module('x-slam_amd.synth'); pl = importlib.import_module('x-slam_amd.pipeline')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
prm = synth.s1_params(n)
kf = pl.KinectFusion(prm)
frames = [torch.from_numpy(synth.s1_frame(k).view(np.int16)).cuda() for k in range(6)]
for f in frames: assert kf.process_frame(f) == 1
c2v = kf.camera2volume()
import time
kf.gauss_newton_terms(frames[5], c2v); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): s = kf.gauss_newton_terms(frames[5], c2v)
dt = (time.perf_counter() - t0) / 20
print(json.dumps({"n": n, "gn_terms_ms_incl_sync": round(dt * 1e3, 4), "count": s[28], "gt_read_GBs": round(4 * n ** 3 / dt / 1e9, 1)}))
ok, ref, hist = kf.relocalize(frames[5], c2v, iterations=5)
print(ok, hist.tolist())
