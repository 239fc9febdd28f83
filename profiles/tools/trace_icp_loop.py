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
    where = tr.reshape(768, 16)[:600, 10]
    tr = tr.reshape(768, 16)[:600, :10].astype(np.int64)
    last = int(np.argmax(tr[:, 9]))
    rel = (tr - tr[:, 0].min()) * 0.01
    rows.append([np.median(rel[:, 1] - rel[:, 0]), (rel[:, 1]).min(), rel[last, 9]])
names = ["entry", "pose", "pixels", "fold", "record stored", "released", "ticket back", "acquired", "gathered", "out"]
print("phase timeline of the last frame's level-0 launch inside the loop (us after the first entry; last workgroup | median over workgroups):")
for i, nme in enumerate(names):
    col = rel[:, i] if i <= 6 else rel[last:last + 1, i]
    print(f"   {nme:14s} {rel[last, i]:7.2f} | {np.median(col):7.2f}")
arr = rel[:, 1]
xcc = ((where >> 32) & 0xF).astype(int)
print("pose arrival per workgroup (us after the first entry): percentiles 0/10/50/90/99/100 =", np.round(np.percentile(arr, [0, 10, 50, 90, 99, 100]), 2))
print("   by XCC:", {int(x): (round(float(np.median(arr[xcc == x])), 2), round(float(arr[xcc == x].max()), 2)) for x in sorted(set(xcc))})
print("   by block index quartile (median, max):", [(round(float(np.median(a)), 2), round(float(a.max()), 2)) for a in np.array_split(arr, 4)])
print("   entry by block index quartile (median, max):", [(round(float(np.median(a)), 2), round(float(a.max()), 2)) for a in np.array_split(rel[:, 0], 4)])
cu = ((where >> 32) & 0xF) * 4096 + ((where >> 8) & 0xFF)
_, inv, cnt = np.unique(cu, return_inverse=True, return_counts=True)
order_on_cu = np.zeros(len(cu), int)
seen = {}
for b in range(len(cu)):
    seen[cu[b]] = seen.get(cu[b], 0) + 1
    order_on_cu[b] = seen[cu[b]]
print("   pose arrival by the workgroup's order of arrival on its CU (median, max, n):", {int(o): (round(float(np.median(arr[order_on_cu == o])), 2), round(float(arr[order_on_cu == o].max()), 2), int((order_on_cu == o).sum())) for o in sorted(set(order_on_cu))})
rows = np.array(rows)
print("level-0 launch inside the tracking loop, 30 frames (us): entry -> pose per workgroup, median %.2f; first workgroup to have the pose "
      "%.2f after the first entry; whole launch %.2f (with its pose already there: see the standalone timeline)" % tuple(np.median(rows, axis=0)))
