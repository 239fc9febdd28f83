"""The integrate kernel of scene S1 on its own (GPU box, repository root), as the pipeline calls it and as a plain call does:
    own      xs_integrate_scaled_ex classifies with the launch's own pose (no pads: what the round-4 standalone figures were)
    ahead    xs_integrate_classify(slack) for the same pose, then the launch with LIST_IS_READY — the classes carry the pose slack's pads,
             as they do inside the pipeline (ClassifyAhead): wider pixel ranges, more boxes on the frustum's edge
    walk     XS_INTEGRATE_NO_TILES: the per-voxel walk everywhere
Kernel time from the event pair on the launch packet, class counts from the workspace header (free / nothing / walk boxes, planes walked,
planes streamed with the in-image test).  python profiles/tools/probe_edge.py [n ...]"""
import ctypes as C, importlib, sys
sys.path.insert(0, '.')
import numpy as np, torch
capi = importlib.import_module('x-slam_amd.capi'); synth = importlib.import_module('x-slam_amd.synth')
H, W = synth.HEIGHT, synth.WIDTH
hip = C.CDLL("libamdhip64.so")
ev = [C.c_void_p(), C.c_void_p()]
for e in ev:
    assert hip.hipEventCreate(C.byref(e)) == 0


def run(n, mode, slack=2.0, threshold=0.0, frames=20, hint=False, sign=False):
    prm = synth.s1_params(n, threshold=threshold); res = [n, n, n]; vs = float(np.float32(prm["tsdf_voxel_size"])); trunc = synth.tranc_dist(prm)
    value = torch.empty((n * n, n), dtype=torch.float32, device="cuda"); weight = torch.empty((n * n, n), dtype=torch.int32, device="cuda")
    grad = torch.empty((n * n, n), dtype=torch.float32, device="cuda")
    capi.init_volume(value, weight, grad, n * 4, res)
    scaled = torch.empty((H, W), dtype=torch.float32, device="cuda"); dmax = torch.zeros(1, dtype=torch.float32, device="cuda")
    tiles = torch.zeros(capi.depth_tiles_bytes(H, W), dtype=torch.uint8, device="cuda")
    ws = torch.zeros(capi.integrate_workspace_bytes(res), dtype=torch.uint8, device="cuda")
    intr = np.array([synth.FX, synth.FY, synth.CX, synth.CY], np.float32)
    counter = torch.zeros(1, dtype=torch.int64, device="cuda")
    s = torch.cuda.current_stream()
    times, Us, classes, vcount = [], [], None, 0
    sm = None
    if sign:   # the pipeline's instance: the kernel also marks the raycast's sign map
        shift = capi.raycast_signmap_shift(intr, vs, trunc)
        sm = torch.zeros(capi.signmap_bytes(res, shift), dtype=torch.uint8, device="cuda")
        capi.signmap_reset(sm, res, shift, trunc)
    for k in range(frames):
        depth = torch.from_numpy(synth.s1_frame(k).view(np.int16)).cuda()
        dmax.zero_()
        capi.scale_depth_tiles(depth, W * 2, H, W, scaled, W * 4, dmax, tiles)
        T = synth.s1_transforms(k, prm)
        counter.zero_()
        flags = 64
        if mode == "ahead":
            capi.integrate_classify_ex(H, W, intr, res, vs, T["Rv2c"], T["tv2c"], trunc, ws, capi.integrate_opts(flags=64, depth_tiles=tiles), slack_scale=slack, depth_max=dmax, stream=s)
            flags |= 4 | 1
        elif mode == "walk":
            flags |= 32
        o = capi.integrate_opts(flags=flags, depth_tiles=tiles, start_event=ev[0], stop_event=ev[1], signmap=sm)
        capi.integrate_scaled_ex2(scaled, W * 4, H, W, intr, 100, res, vs, T["Rv2c"], T["tv2c"], trunc, value, weight, grad, n * 4, o, threshold=threshold,
                                  updated=counter, depth_max=dmax, workspace=ws, stream=s)
        torch.cuda.synchronize()
        head = ws[192:232].view(torch.int32).cpu().numpy()
        classes = [int(x) for x in head[:7]]; vcount = 0
        dt = C.c_float(0); assert hip.hipEventElapsedTime(C.byref(dt), ev[0], ev[1]) == 0
        if k >= 4:
            times.append(dt.value * 1e3); Us.append(int(counter.item()))
    U, t = np.median(Us), np.median(times)
    print(f"n {n} {mode:6s}{' +sign' if sign else '      '} thr {threshold}: U {U:.0f}  kernel {t:.1f} us (min {min(times):.1f})  {24 * U / t / 1e6:.2f} TB/s algorithmic;  boxes free / nothing / walk "
          f"{classes[0:3]}  planes walked {classes[3]}  edge planes {classes[6]}  workgroups with work {vcount}", flush=True)
    return value, weight, grad


if __name__ == "__main__":
    import os
    sizes = [int(v) for v in sys.argv[1:]] or [512, 1024]
    for c in (synth.FX, synth.FY):   # what the orchestrator does in AllocateBuffers: the band's / fx, / fy take the verified short division
        capi.const_div_prepare(c)
    quick = os.environ.get("PROBE_QUICK")          # A/B sweeps: the two pipeline-like modes only, no comparison against the walk
    for n in sizes:
        if quick:
            run(n, "own"); run(n, "own", sign=True); run(n, "ahead"); run(n, "ahead", sign=True)
            torch.cuda.empty_cache()
            continue
        ref = None
        for mode, hint in (("walk", False), ("own", False), ("own", True), ("ahead", False), ("ahead", True)):
            out = run(n, mode, hint=hint)
            if ref is None: ref = [t.clone() for t in out]
            else:
                same = all(bool(torch.equal(a.view(torch.int32), b.view(torch.int32))) for a, b in zip(ref, out))
                print(f"   volumes identical to the walk's, bit for bit: {same}", flush=True)
                assert same
            del out
        del ref
        torch.cuda.empty_cache()
    if not quick:
        run(512, "walk", threshold=0.02); run(512, "own", threshold=0.02, hint=True); run(512, "ahead", threshold=0.02, hint=True)
