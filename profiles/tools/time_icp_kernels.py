"""55 launches of the ICP reduction per pyramid level on real maps, back to back (profiles/tools/time_icp_kernels.sh reads the kernel trace)."""

import ctypes as C, importlib, sys
sys.path.insert(0, '.')
import numpy as np, torch
capi = importlib.import_module('x-slam_amd.capi'); synth = importlib.import_module('x-slam_amd.synth')
exec(open('profiles/tools/probe_icp.py').read().split("ws = torch.zeros")[0])
ws = torch.zeros(capi.icp_workspace_bytes(), dtype=torch.uint8, device="cuda")
sums = torch.zeros(64, dtype=torch.float64, device="cuda")
I = np.zeros((3, 3, 2), np.float32); I[[0, 1, 2], [0, 1, 2], 0] = 1
t0 = np.array([[1e-3, 1e-7], [0, 0], [0, 0]], np.float32)
for l in (2, 1, 0):
    k, v, nm, h, w = maps[l]; _, pv, pn, _, _ = prev[l]
    for rep in range(55):
        capi.icp_accumulate(I, t0, v, nm, I, np.zeros(6), k, pv, pn, w * 8, h, w, 0.1, 0.26, ws, sums)
    torch.cuda.synchronize()
