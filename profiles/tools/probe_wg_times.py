"""GPU box, repository root, library built with -DXS_EXPERIMENTS -DXS_WG_TIMES (profiles/tools/probe_wg_times.sh; XS_PROBE_N = volume edge, default 512): when every workgroup of the S1 integrate kernel begins and ends
(100 MHz wall clock), by the class of its box.  python profiles/tools/probe_wg_times.py"""
import importlib, os, sys
sys.path.insert(0, '.')
import numpy as np, torch
capi = importlib.import_module('x-slam_amd.capi'); synth = importlib.import_module('x-slam_amd.synth')
H, W, n = synth.HEIGHT, synth.WIDTH, int(os.environ.get("XS_PROBE_N", "512"))
prm = synth.s1_params(n); res = [n, n, n]; vs = float(np.float32(prm["tsdf_voxel_size"])); trunc = synth.tranc_dist(prm)
value = torch.empty((n * n, n), dtype=torch.float32, device="cuda"); weight = torch.empty((n * n, n), dtype=torch.int32, device="cuda")
grad = torch.empty((n * n, n), dtype=torch.float32, device="cuda")
capi.init_volume(value, weight, grad, n * 4, res)
scaled = torch.empty((H, W), dtype=torch.float32, device="cuda"); dmax = torch.zeros(1, dtype=torch.float32, device="cuda")
nws = capi.integrate_workspace_bytes(res)
ws = torch.zeros(nws, dtype=torch.uint8, device="cuda")
intr = np.array([synth.FX, synth.FY, synth.CX, synth.CY], np.float32)
s = torch.cuda.current_stream()
nb = (n // 32) * (n // 8) * (n // 8)
LIST0 = 256 + 8192 * 4                                       # header + the update counts' room (WS_LIST_OFFSET)
list_bytes = (LIST0 + nb * 4 * 4 + 255) // 256 * 256          # workspace_list_bytes: bricks of 2 planes -> 4x the 8-plane count
class_bytes = (nb * 4 * 4 * 4 + 255) // 256 * 256      # a 32-bit word per box, four boxes per (2-plane) brick
off = list_bytes + class_bytes + (1 << 18)
import ctypes as C
hip = C.CDLL("libamdhip64.so")
ev = [C.c_void_p(), C.c_void_p()]
for e_ in ev:
    assert hip.hipEventCreate(C.byref(e_)) == 0
for k in range(12):
    depth = torch.from_numpy(synth.s1_frame(k).view(np.int16)).cuda()
    capi.scale_depth_max(depth, W * 2, H, W, scaled, W * 4, dmax)
    T = synth.s1_transforms(k, prm)
    o = capi.integrate_opts(flags=0, start_event=ev[0], stop_event=ev[1])
    capi.integrate_scaled_ex2(scaled, W * 4, H, W, intr, 100, res, vs, T["Rv2c"], T["tv2c"], trunc, value, weight, grad, n * 4, o, depth_max=dmax, workspace=ws, stream=s)
    torch.cuda.synchronize()
dt = C.c_float(0); assert hip.hipEventElapsedTime(C.byref(dt), ev[0], ev[1]) == 0
print(f"the same launch by its event pair (begin / end timestamps of the dispatch packet): {dt.value * 1e3:.2f} us")
if os.environ.get("XS_PROBE_EVENT_ONLY"):      # (a library without -DXS_WG_TIMES leaves no records to read)
    sys.exit(0)
nwalk, nother = capi.integrate_listed(ws); listed = nwalk + nother
pair2 = ws[208:216].view(torch.int32).cpu().numpy()          # the ordered list's two runs (a brick with a long walk has two entries in the first)
entries = int(pair2[0]) + int(pair2[1])
print(f"entries of the ordered list {entries} ({int(pair2[0])} in the front run)")
G = int(os.environ.get("XS_BRICK_GRID", "0")) or 8192      # (an -DXS_EXPERIMENTS library sizes the grid from the same variable)
count = min(entries, G)     # (a launch has 8 192 workgroups: with more bricks listed a workgroup takes a second one; its record then covers both)
print(f"bricks listed {listed} ({nwalk} with planes to walk)")
rec = ws[off:off + G * 4 * 16].view(torch.int32).cpu().numpy().astype(np.int64).reshape(G, 4, 4) & 0xffffffff
t0 = rec[..., 0].min()
b = (rec[..., 0] - t0) * 0.01; e = (rec[..., 1] - t0) * 0.01
print(f"bricks {count}; workgroups {G}; kernel span {e.max():.2f} us; last workgroup begins at {b.max():.2f} us")
work = rec[:count]
nfree, nempty = work[..., 2] & 0xff, (work[..., 2] >> 8) & 0xff
walked = 8 - nfree - nempty
print("planes walked per box:", {int(k): int((walked == k).sum()) for k in range(9)})
for cls, name in ((1, "free"), (2, "nothing to write"), (0, "per-voxel walk")):
    m = (nfree == 8) if cls == 1 else (nempty == 8) if cls == 2 else (walked > 0)
    if m.any():
        d = (work[..., 1] - work[..., 0])[m] * 0.01
        print(f"{name:18s} waves {m.sum():5d}  begin {b[:count][m].mean():6.2f} us (max {b[:count][m].max():6.2f})  duration mean {d.mean():6.2f}  p50 {np.median(d):6.2f}  p90 {np.percentile(d, 90):6.2f}  max {d.max():6.2f}  end max {e[:count][m].max():6.2f}")
idle = rec[count:]
if len(idle):
    print(f"workgroups without a brick: {len(idle)}  begin mean {((idle[..., 0] - t0) * 0.01).mean():.2f}  last end {((idle[..., 1] - t0) * 0.01).max():.2f} us")
else:
    print(f"every workgroup has work: {listed} bricks for {G} workgroups (some take a second brick)")
hist, edges = np.histogram(e[:count].max(axis=1), bins=12)
print("end-time histogram of working workgroups (us):", [(round(float(a), 1), int(c)) for a, c in zip(edges[:-1], hist)])
_, cap, _, second_off = capi.integrate_list_layout(res)
region = ws[second_off:second_off + cap * 4].view(torch.int32).cpu().numpy()      # the ordered list: walked bricks from the front of its region, the others from its back
n2 = int(pair2[0])
lst = np.array([region[e] if e < n2 else region[cap - 1 - (e - n2)] for e in range(count)])
dur = (work[..., 1] - work[..., 0]) * 0.01
order = np.argsort(-dur.max(axis=1))[:16]
print("slowest workgroups: duration per wave (us), classes, lane-0 voxels written, brick (bx, by, bz)")
for i in order:
    b_ = int(lst[i])
    print(f"  wg {i:5d}  {np.round(dur[i], 1).tolist()}  free/empty planes {[(int(a_), int(b_)) for a_, b_ in zip(nfree[i], nempty[i])]}  n0 {(work[i, :, 3] & 0xff).tolist()}  brick {(b_ & 1023, (b_ >> 10) & 1023, b_ >> 20)}  begin {b[i].min():.2f}")
for k in range(1, 9):
    sel = walked == k
    if sel.any():
        print(f"waves that walk {k} planes: {sel.sum():5d}  mean {dur[sel].mean():6.2f} us  max {dur[sel].max():6.2f}")

# where the work ran: a CU is (XCC, SE, SH, CU) of the hardware id each wave recorded
hw = work[..., 3]
cu_key = ((hw >> 8) & 0xf) * 4096 + ((hw >> 29) & 7) * 512 + ((hw >> 28) & 1) * 256 + ((hw >> 24) & 0xf)
cus = {}
for i in range(count):
    for w in range(4):
        c = cus.setdefault(int(cu_key[i, w]), [0, 0, 0.0, 0.0])
        c[0] += 1; c[1] += int(walked[i, w]); c[2] = max(c[2], float(e[i, w])); c[3] += float(dur[i, w]) if walked[i, w] > 0 else 0.0
rows = sorted(cus.values(), key=lambda c: -c[2])
print(f"CUs seen {len(cus)}; planes walked per CU: mean {np.mean([c[1] for c in rows]):.1f}  min {min(c[1] for c in rows)}  max {max(c[1] for c in rows)}")
print("last CUs to finish: (waves, planes walked, end us)", [(c[0], c[1], round(c[2], 1)) for c in rows[:12]])
print("first CUs to finish:", [(c[0], c[1], round(c[2], 1)) for c in rows[-12:]])
pw = np.array([c[1] for c in rows], float); en = np.array([c[2] for c in rows])
print(f"correlation(planes walked on the CU, the CU's end time) = {np.corrcoef(pw, en)[0, 1]:.2f};  end = {np.polyfit(pw, en, 1)[0]:.3f} us per plane + {np.polyfit(pw, en, 1)[1]:.2f}")
wg_cu = cu_key[:, 0]
print("workgroup -> CU: wg 0..15:", [hex(int(v)) for v in wg_cu[:16]], " same CU for wg i and i + 256:", float((wg_cu[:count - 256] == wg_cu[256:count]).mean()))
