"""ICP reduction kernel alone: average duration of back-to-back launches per pyramid level (HIP events on the launch stream),
on real maps (frame 3 of scene S1 against the maps of frame 2) and on all-NaN maps (no pixel passes the first gate: what
a launch costs before any pixel work — dispatch, LDS fold of zeros, record, ticket, last workgroup's sum).
usage (repository root): python profiles/tools/probe_icp.py"""
import importlib, sys, json
sys.path.insert(0, '.')
import numpy as np, torch
capi = importlib.import_module('x-slam_amd.capi'); synth = importlib.import_module('x-slam_amd.synth')
W, H = 640, 480
intr = [481.2, -480.0, 319.5, 239.5]
def frame_maps(frame):
    depth = torch.from_numpy(synth.s1_frame(frame).view(np.int16)).cuda()
    d0 = torch.zeros((H, W, 2), dtype=torch.float32, device="cuda")
    capi.bilateral_filter(depth, W * 2, H, W, d0, W * 8)
    ds = [d0]
    for l in (1, 2):
        dn = torch.zeros((H >> l, W >> l, 2), dtype=torch.float32, device="cuda")
        capi.pyr_down(ds[-1], (W >> (l - 1)) * 8, H >> (l - 1), W >> (l - 1), dn, (W >> l) * 8)
        ds.append(dn)
    out = []
    for l in range(3):
        h, w = H >> l, W >> l
        k = [intr[0] / 2 ** l, intr[1] / 2 ** l, intr[2] / 2 ** l, intr[3] / 2 ** l]
        v = torch.zeros((3 * h, w, 2), dtype=torch.float32, device="cuda"); n = torch.zeros_like(v)
        capi.create_vmap(k, ds[l], w * 8, h, w, v, w * 8)
        capi.create_nmap(v, n, w * 8, h, w)
        out.append((k, v, n, h, w))
    return out
maps, prev = frame_maps(3), frame_maps(2)     # current frame against the previous one's maps (camera moved ~7 mm)
ws = torch.zeros(capi.icp_workspace_bytes(), dtype=torch.uint8, device="cuda")
sums = torch.zeros(64, dtype=torch.float64, device="cuda")
I = np.zeros((3, 3, 2), np.float32); I[[0, 1, 2], [0, 1, 2], 0] = 1
t0 = np.array([[1e-3, 1e-7], [0, 0], [0, 0]], np.float32)
s = torch.cuda.current_stream()
out = {}
for tag in ("real", "nan"):
    for l in (2, 1, 0):
        k, v, nm, h, w = maps[l]
        _, pv, pn, _, _ = prev[l]
        if tag == "nan":
            v = torch.full_like(v, float("nan")); nm = torch.full_like(nm, float("nan"))
        run = lambda: capi.icp_accumulate(I, t0, v, nm, I, np.zeros(6), k, pv, pn, w * 8, h, w, 0.1, 0.26, ws, sums, stream=s)
        for _ in range(10): run()
        torch.cuda.synchronize()
        batches = []
        for _ in range(9):      # median of nine batches: a host-side hiccup while enqueueing (tens of ms, once in a while) lands in one of them
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(40): run()
            e1.record(s); torch.cuda.synchronize()
            batches.append(e0.elapsed_time(e1) / 40 * 1000)
        out[f"{tag}_level{l}_us"] = round(sorted(batches)[4], 2)
        if tag == "real": out[f"inliers_level{l}"] = float(sums[54].item())
print(json.dumps(out))
