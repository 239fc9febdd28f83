"""Phase timeline of the ICP reduction kernel (experiment build -DXS_ICP_TRACE; profiles/tools/trace_icp.sh): one launch per pyramid level
on real maps, every workgroup's 100 MHz stamps; prints, relative to the earliest entry, when the LAST workgroup (the one that gathers)
passed each phase and the spread over all workgroups."""
import ctypes as C, importlib, sys
sys.path.insert(0, '.')
import numpy as np, torch
capi = importlib.import_module('x-slam_amd.capi'); synth = importlib.import_module('x-slam_amd.synth')
exec(open('profiles/tools/probe_icp.py').read().split("ws = torch.zeros")[0])
ws = torch.zeros(capi.icp_workspace_bytes(), dtype=torch.uint8, device="cuda")
sums = torch.zeros(64, dtype=torch.float64, device="cuda")
I = np.zeros((3, 3, 2), np.float32); I[[0, 1, 2], [0, 1, 2], 0] = 1
t0 = np.array([[1e-3, 1e-7], [0, 0], [0, 0]], np.float32)
names = ["entry", "pose", "pixels", "fold", "record stored", "released", "ticket back", "acquired", "gathered", "out"]
capi._lib.xs_debug_icp_trace.argtypes = [C.c_void_p]
for l in (2, 1, 0):
    k, v, nm, h, w = maps[l]; _, pv, pn, _, _ = prev[l]
    blocks = capi.icp_records_count(w, 0, h)
    for rep in range(3):
        capi.icp_accumulate(I, t0, v, nm, I, np.zeros(6), k, pv, pn, w * 8, h, w, 0.1, 0.26, ws, sums)
        torch.cuda.synchronize()
    tr = np.zeros(768 * 16, np.uint64)
    assert capi._lib.xs_debug_icp_trace(tr.ctypes.data) == 0
    where = tr.reshape(768, 16)[:blocks, 10]                      # XCC_ID << 32 | HW_ID of the workgroup's first wave
    sub = tr.reshape(768, 16)[:blocks, 11:15].astype(np.int64)    # inside the fold (two-pass instances): pass 0 written / added, pass 1 written / added
    tr = tr.reshape(768, 16)[:blocks, :10].astype(np.int64)
    base = tr[:, 0].min()
    last = int(np.argmax(tr[:, 9]))            # only the last workgroup stamps 7..9 in this launch (older stamps are smaller)
    rel = (tr - base) * 10.0 / 1000.0          # us
    print(f"level {l}: {blocks} workgroups; last = {last}")
    for i, nme in enumerate(names):
        col = rel[:, i] if i <= 6 else rel[last:last + 1, i]
        print(f"   {nme:14s} last wg {rel[last, i]:7.2f} us   all wgs min {col.min():7.2f}  median {np.median(col):7.2f}  max {col.max():7.2f}")
    if sub.min() > 0:
        st = np.concatenate([tr[:, 2:3], sub, tr[:, 3:4]], axis=1)
        d = np.diff(st, axis=1) * 0.01
        print("   inside the fold, first wave of each workgroup (median / p90 us): " + "  ".join(f"{n} {np.median(d[:, i]):.2f} / {np.percentile(d[:, i], 90):.2f}" for i, n in enumerate(
              ["pixels -> pass 0 written", "-> pass 0 added", "-> pass 1 written", "-> pass 1 added", "-> record summed (barrier)"])))
    # the same per workgroup: how long each phase took, and how that goes with the number of workgroups sharing its CU
    cu = ((where >> 32) & 0xF) * 4096 + ((where >> 8) & 0xFF)       # (XCC, SE / SH / CU fields of HW_ID)
    _, inv, cnt = np.unique(cu, return_inverse=True, return_counts=True)
    share = cnt[inv]
    dur = np.diff(rel[:, :7], axis=1)
    print(f"   workgroups per CU: " + ", ".join(f"{c} on {int((cnt == c).sum())} CUs" for c in sorted(set(cnt))))
    for i in range(6):
        d = dur[:, i]
        by = "  ".join(f"[{c}/CU] {np.median(d[share == c]):5.2f}" for c in sorted(set(cnt)))
        print(f"   {names[i]:>13s} -> {names[i + 1]:14s} median {np.median(d):5.2f}  p90 {np.percentile(d, 90):5.2f}  max {d.max():5.2f} us   medians {by}")
