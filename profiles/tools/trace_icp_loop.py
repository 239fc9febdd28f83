"""How long an ICP launch of the running pipeline waits for its pose (experiment build -DXS_ICP_TRACE, profiles/tools/trace_icp.sh
builds it): tracks 30 frames of scene S1 and, after each, reads the stamps of the frame's last launch (level 0, fifth iteration).
entry -> pose is then the time from that launch becoming resident — right behind the previous iteration's last workgroup — to the
host's post arriving: completion word over PCIe, host reads the 55 sums, solves, posts through the BAR, the poll sees it."""
import ctypes as C, importlib, sys
sys.path.insert(0, '.')
import numpy as np, torch
capi = importlib.import_module('x-slam_amd.capi'); synth = importlib.import_module('x-slam_amd.synth'); pl = importlib.import_module('x-slam_amd.pipeline')
capi._lib.xs_debug_icp_trace.argtypes = [C.c_void_p]
runner = pl.KinectFusion(synth.s1_params(512))
frames = [torch.from_numpy(synth.s1_frame(k).view(np.int16)).cuda() for k in range(40)]
rows = []
for i, f in enumerate(frames):
    assert runner.process_frame(f) == 1
    runner.synchronize()
    if i < 10:
        continue
    tr = np.zeros(768 * 16, np.uint64)
    assert capi._lib.xs_debug_icp_trace(tr.ctypes.data) == 0
    tr = tr.reshape(768, 16)[:600, :10].astype(np.int64)
    last = int(np.argmax(tr[:, 9]))
    rel = (tr - tr[:, 0].min()) * 0.01
    rows.append([np.median(rel[:, 1] - rel[:, 0]), (rel[:, 1]).min(), rel[last, 9]])
rows = np.array(rows)
print("level-0 launch inside the tracking loop, 30 frames (us): entry -> pose per workgroup, median %.2f; first workgroup to have the pose "
      "%.2f after the first entry; whole launch %.2f (with its pose already there: see the standalone timeline)" % tuple(np.median(rows, axis=0)))
