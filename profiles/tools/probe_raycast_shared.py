"""GPU box, library built with -DXS_EXPERIMENTS -DXS_RAY_SHARED_LOADS (profiles/tools/ab_raycast_shared.sh): raycast alone (march + crossing
on a tracked 512^3 volume, bench.py's raycast figures) with the crossing's loads as they are, and with the second sample of every pair
reusing the first one's corners (the 24 of the normal's 48 8-byte loads and their address arithmetic gone — more than real sharing can remove (x pair 16 -> 8, y and z pairs 16 -> 12: 16 of 48); wrong normals, timing only)."""
import ctypes as C, importlib, sys
sys.path.insert(0, '.')
import numpy as np, torch
import bench
capi = importlib.import_module('x-slam_amd.capi'); synth = importlib.import_module('x-slam_amd.synth'); pl = importlib.import_module('x-slam_amd.pipeline')
pl.set_stream(torch.cuda.current_stream())
N = 512
prm = synth.s1_params(N)
r = pl.KinectFusion(prm)
for k in range(40):
    assert r.process_frame(torch.from_numpy(synth.s1_frame(k).view(np.int16)).cuda()) == 1
f = capi._lib.xs_raycast_exp_shared_loads
f.restype = C.c_int; f.argtypes = [C.c_int]
for rep in range(3):
    for on in (0, 1):
        assert f(on) == 0
        fig = bench.raycast_figures(torch, capi, synth, r, prm, N)
        print(f"second sample reuses the first one's corners: {bool(on)}   raycast alone {fig['ms_per_frame_alone']:.5f} ms   every step {fig['every_step']['ms_per_frame_alone']:.5f} ms   hit fraction {fig['hit_fraction']}", flush=True)
f(0)
