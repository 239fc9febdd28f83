#!/usr/bin/env python3
"""bench.py — frames/s of first-order-CSFD XKinectFusion (BASELINE.json metric) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size 512]

A step = ProcessFrame of one synthetic 640x480 depth frame (scene S1)
into the TSDF volume: surface measure -> 12 ICP iterations -> integrate -> raycast -> map
pyramid, all in complex float with the CSFD seed i*1e-7 on world2camera(0,3).  Depth frames
are resident in HBM before the timed region.  One JSON line on rank 0 with the contract's
fields plus
  roofline     : the TSDF-integrate kernel — algorithmic bytes 24*U + 2*W*H per launch (U = voxels
                 written, counted by the kernel) / its mean duration from a HIP event pair attached to
                 the kernel's dispatch on the launch stream, inside the timed region; peak = 8 TB/s HBM3E;
                 traffic = HBM bytes per launch from the committed rocprofv3 counter passes
                 (profiles/r01_integrate_pmc_v3.json)
  roofline_s2  : the same object for scene S2 (SURVEY 8d's frustum-filling placement, the one the
                 HBM claim is made on: ~0.95 GB per launch instead of S1's ~45 MB, which is over in
                 the time a launch takes to ramp up)
  cpu_baseline : the CPU oracle (oracle/, a port — not the product path) timed on this host's
                 cores on a bounded sample of the same workload
  stages_ms    : per-stage mean ms, from a separate untimed pass with an event pair around every stage
                 (those pairs are packets the kernels would queue behind: not in the timed region)
N > 1 (launched by torch.distributed.run, one rank per GPU): the volume is sharded by z-slab
(integrate, raycast + a two-collective composite over RCCL); the ICP runs replicated on every rank
(default) or row-sharded with the 6x6 / 6x1 normal equations all-reduced per iteration
(--icp-shard-rows; the other mode is timed briefly as "alt_mode"); see x-slam_amd/sharded.py.
Max-over-ranks timing, "scaling": "strong" (one frame stream, fixed total work).
--scene s3: a box room on the same camera path (every degree of freedom constrained) for long runs.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6.3 TB/s is what a copy achieves
W, H = 640, 480


def cpu_baseline(synth, size, budget_s=20.0):
    """Oracle pipeline (CPU restatement, OpenMP over all host cores) on the first frames of the
    same stream.  Reported beside the GPU number; not a target."""
    from oracle import oracle as orc
    orc.build(ref=False)
    o = orc.Oracle()
    kf = orc.OracleKinFu(o, orc.params_from_dict(synth.s1_params(size)))
    frames, t_used, k = 0, 0.0, 0
    kf.process_frame(synth.s1_frame(0))  # frame 0 has no ICP: not timed
    while t_used < budget_s and k < 30:
        k += 1
        d = synth.s1_frame(k)
        t0 = time.perf_counter()
        ok = kf.process_frame(d)
        t_used += time.perf_counter() - t0
        frames += 1
        if not ok:
            break
    cores = o._num_threads()
    kf.close()
    return {"value": round(frames / t_used, 4), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"oracle pipeline, scene S1 {size}^3, frames 1..{frames} ({t_used:.1f} s), OpenMP x{cores}"}


def integrate_s2_probe(torch, capi, synth, size=512, reps=20):
    """Scene S2 (frustum-filling placement): the configuration on which integrate is
    HBM-bound.  Times the integrate kernel alone with HIP events."""
    prm = synth.s2_params(size)
    n = size
    res = [n, n, n]
    vs = np.float32(prm["tsdf_voxel_size"])
    trunc = float(max(np.float32(vs * np.float32(3.0)), np.float32(np.float32(2.1) * vs)))
    value = torch.empty((n * n, n), dtype=torch.float32, device="cuda")
    weight = torch.empty((n * n, n), dtype=torch.int32, device="cuda")
    grad = torch.empty((n * n, n), dtype=torch.float32, device="cuda")
    capi.init_volume(value, weight, grad, n * 4, res)
    depth = torch.from_numpy(synth.render_s2().view(np.int16)).cuda()
    scaled = torch.empty((H, W), dtype=torch.float32, device="cuda")
    dmax = torch.zeros(1, dtype=torch.float32, device="cuda")
    capi.scale_depth_max(depth, W * 2, H, W, scaled, W * 4, dmax)
    ws = torch.zeros(capi.integrate_workspace_bytes(res), dtype=torch.uint8, device="cuda")
    # v2c for c2v = translate(2.56, 2.56, 0.05): identity rotation, seed on t_x
    R = np.zeros((3, 3, 2), np.float32)
    R[[0, 1, 2], [0, 1, 2], 0] = 1
    t = np.zeros((3, 2), np.float32)
    t[:, 0] = [-prm["init_x"], -prm["init_y"], -prm["init_z"]]
    t[0, 1] = 1e-7
    intr = np.array([synth.FX, synth.FY, synth.CX, synth.CY], np.float32)
    counter = torch.zeros(1, dtype=torch.int64, device="cuda")
    s = torch.cuda.current_stream()
    args = (scaled, W * 4, H, W, intr, 100, res, float(vs), R, t, trunc, value, weight, grad, n * 4)
    capi.integrate_scaled(*args, updated=counter, depth_max=dmax, workspace=ws, stream=s)
    torch.cuda.synchronize()
    U = int(counter.item())
    # the integrate kernel alone, like the pipeline's roofline: HIP events recorded by the launcher immediately
    # around k_integrate_bricks (after the clear and the brick classification, before the count fold)
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    ev = [C.c_void_p(), C.c_void_p()]
    for e in ev:
        assert hip.hipEventCreate(C.byref(e)) == 0
    capi._lib.xs_integrate_set_timing_events(ev[0], ev[1])
    ms_k = 0.0
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(s)
    for _ in range(reps):
        capi.integrate_scaled(*args, depth_max=dmax, workspace=ws, stream=s)
        torch.cuda.synchronize()
        dt = C.c_float(0)
        assert hip.hipEventElapsedTime(C.byref(dt), ev[0], ev[1]) == 0
        ms_k += dt.value
    t1.record(s)
    torch.cuda.synchronize()
    capi._lib.xs_integrate_set_timing_events(None, None)
    for e in ev:
        hip.hipEventDestroy(e)
    ms = ms_k / reps
    # and the whole call (clear + classification + integrate + fold), back to back without host waits
    t0.record(s)
    for _ in range(reps):
        capi.integrate_scaled(*args, depth_max=dmax, workspace=ws, stream=s)
    t1.record(s)
    torch.cuda.synchronize()
    ms_call = t0.elapsed_time(t1) / reps
    nbytes = 24.0 * U + 2.0 * W * H
    return {"kernel": "k_integrate_bricks (TSDF integrate)", "whole_call_ms": round(ms_call, 4),
            "scene": f"S2 {size}^3 (frustum-filling placement, SURVEY 8d)", "bound": "hbm", "achieved": round(nbytes / ms / 1e6, 1),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(nbytes / ms / 1e6 / HBM_PEAK_GBS, 4),
            "traffic": pmc_traffic("s2") if size == 512 else None, "algorithmic_bytes_per_launch": round(nbytes), "U": U,
            "U_frac": round(U / n ** 3, 4), "kernel_ms": round(ms, 4)}


PMC_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_integrate_pmc_v3.json")


def pmc_traffic(scene):
    """HBM bytes per launch of the integrate kernel from the committed rocprofv3 counter passes
    (FETCH_SIZE and WRITE_SIZE in separate runs, each scaled by the factor a known-bytes kernel with the
    same access pattern gave: profiles/tools/).  Counters cannot be read from inside this process;
    None when the file is absent."""
    try:
        with open(PMC_FILE) as f:
            v = json.load(f)[scene].get("traffic_bytes_per_launch")
        return None if v is None else round(v)
    except (OSError, KeyError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--scene", choices=["s1", "s3"], default="s1",
                    help="s1: SURVEY's wall + sphere (the headline workload; tracks for ~530 frames at 512^3).  s3: the inside of a box room on the "
                         "same camera path — every degree of freedom constrained, tracks indefinitely (long runs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-s2", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse)")
    ap.add_argument("--icp-shard-rows", action="store_true",
                    help="N > 1: split the ICP pixel rows over the ranks and all-reduce the 6x6|6x1 sums every iteration "
                         "(default: every rank runs the whole ICP, no collective inside the loop)")
    ap.add_argument("--host-frames", action="store_true",
                    help="N = 1: hand every frame over from host memory (pinned staging + asynchronous upload) instead of HBM; "
                         "the PCIe-inclusive rate quoted in DESIGN.md, never the headline value")
    ap.add_argument("--force-composite", action="store_true",
                    help="rehearsal on one GPU: run the sharded raycast composite and its RCCL collectives with a single rank")
    ap.add_argument("--no-alt", action="store_true", help="N > 1: skip the short run of the other ICP sharding mode after the timed region")
    ap.add_argument("--no-post-pose", action="store_true",
                    help="launch every ICP iteration after its solve (the reference's order) instead of posting the pose to an already enqueued launch")
    ap.add_argument("--icp-solve", choices=["host", "device"], default=None,
                    help="where the pose update between ICP iterations runs (default: the library's default)")
    ap.add_argument("--same-gpu", action="store_true", help="rehearsal: put every rank on cuda:0 (needs --backend gloo)")
    a = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    torch.cuda.set_device(0 if a.same_gpu else local_rank)
    synth = importlib.import_module("x-slam_amd.synth")
    capi = importlib.import_module("x-slam_amd.capi")
    pl = importlib.import_module("x-slam_amd.pipeline")

    K, Wm, N = a.steps, a.warmup, a.size
    T = 300
    nframes = K + Wm
    # frame 0 initialises the map (no ICP); the stream then follows the S1 trajectory
    render = synth.s3_frame if a.scene == "s3" else synth.s1_frame
    frames_np = [render(k % T) for k in range(min(nframes + 1, T))]
    dev_frames = [torch.from_numpy(f.view(np.int16)).cuda() for f in frames_np]
    stream = torch.cuda.current_stream()
    pl.set_stream(stream)

    if world > 1 or a.force_composite:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(a.backend, rank=rank, world_size=world)
        sharded = importlib.import_module("x-slam_amd.sharded")
        prm = dict(synth.s1_params(N), icp_shard_rows=bool(a.icp_shard_rows), force_shard_composite=bool(a.force_composite))
        runner = sharded.ShardedKinectFusion(prm, rank, world, dist)
    else:
        dist = None
        prm = synth.s1_params(N)
        if a.icp_solve is not None:
            prm["icp_solve_on_device"] = (a.icp_solve == "device")
        if a.no_post_pose:
            prm["icp_post_pose"] = False
        runner = pl.KinectFusion(prm)

    def frame(i):
        return dev_frames[i % len(dev_frames)]

    if a.host_frames and world == 1 and not a.force_composite:
        _process = runner.process_frame_host

        def host_step(i):
            buf = runner.ingest_buffer()
            buf[...] = frames_np[i % len(frames_np)]   # stands for the decode of the next image
            return _process(buf)
        runner.process_frame = host_step

        def frame(i):  # noqa: F811 — the host path takes the frame number
            return i

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    ok = runner.process_frame(frame(0))
    assert ok == 1
    for i in range(1, Wm + 1):
        assert runner.process_frame(frame(i)) == 1, "tracking lost during warm-up"
    # level 1: only the integrate kernel's event pair (the roofline figure) and the counters are live in the
    # timed region; the other stages' event pairs would each put a packet between two kernels
    runner.set_profiling(1)
    runner.reset_stage_times()
    usum = 0
    barrier()
    t0 = time.perf_counter()
    for i in range(Wm + 1, Wm + 1 + K):
        assert runner.process_frame(frame(i)) == 1, (
            f"tracking lost at frame {i}: scene S1 (a wall and a sphere) constrains sliding along the wall only weakly; it tracks for "
            f"~530 frames at 512^3 and ~120 at 256^3 — use fewer --steps")
    barrier()
    dt = time.perf_counter() - t0
    usum = runner.cumulative_counters()[0]
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        ut = torch.tensor([usum], dtype=torch.int64, device="cuda")
        dist.all_reduce(ut)
        usum = int(ut.item())
    st = runner.stage_times()
    # per-stage times: a separate, untimed pass with an event pair around every stage
    ks = max(10, min(K, 60))
    runner.set_profiling(2)
    runner.reset_stage_times()
    for i in range(Wm + 1 + K, Wm + 1 + K + ks):
        assert runner.process_frame(frame(i)) == 1, "tracking lost (stage pass)"
    barrier()
    st_all = runner.stage_times()
    runner.set_profiling(0)
    fps = K / dt
    U = usum / K
    int_ms = st["integrate"][0] / max(st["integrate"][1], 1)
    nbytes = 24.0 * U + 2.0 * W * H
    if world > 1:
        nbytes = nbytes / world  # per launch: each rank's kernel covers its own slab
    out = {
        "metric": "fps XKinectFusion 512^3 TSDF 640x480 CSFD @1/2/4/8 GPU; HBM GB/s on integrate", "value": round(fps, 3), "unit": "frames/s",
        "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": round(1000.0 * dt / K, 4), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32 (complex<f32> CSFD)",
        "data": "synthetic" + (" (frames handed over from host memory: PCIe inclusive, not the headline configuration)" if a.host_frames else ""),
        "config": {"workload": f"XKinectFusion scene {'S3 (box room' if a.scene == 's3' else 'S1 (plane+sphere'}, ICL intrinsics), {N}^3 TSDF, 640x480, first-order CSFD seed "
                               f"i*1e-7 on world2camera(0,3), 3 pyramid levels x (5,4,3) ICP iterations",
                   "volume": f"{N}^3", "voxel_size_m": round(7.68 / N, 6), "frames_resident_in_hbm": True,
                   "parallelism": "single GPU" if world == 1 else (
                       f"z-slab x{world} (integrate, raycast + RCCL composite: min of first-event keys, sum of owner maps); ICP " +
                       ("row shards + RCCL all-reduce of the 6x6|6x1 normal equations per iteration" if a.icp_shard_rows
                        else "replicated on every rank (identical bits, no collective inside the loop)"))},
        "roofline": {"kernel": "k_integrate_bricks (TSDF integrate)", "bound": "hbm", "achieved": round(nbytes / int_ms / 1e6, 2),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(nbytes / int_ms / 1e6 / HBM_PEAK_GBS, 5),
                     "traffic": pmc_traffic("s1") if (world == 1 and N == 512) else None,
                     "algorithmic_bytes_per_launch": round(nbytes), "U_per_frame": round(U, 1), "kernel_ms": round(int_ms, 5)},
        "stages_ms": dict({k: round(v[0] / max(v[1], 1), 5) for k, v in st_all.items()},
                          note=f"separate pass of {ks} frames with an event pair around every stage (not the timed region)"),
    }
    if world > 1 and not a.no_alt:
        # the other ICP sharding mode on the same node, briefly, so both are on record
        runner.close()
        alt = sharded.ShardedKinectFusion(dict(prm, icp_shard_rows=not a.icp_shard_rows), rank, world, dist)
        ka = max(10, min(K, 60))
        for i in range(0, 6):
            assert alt.process_frame(frame(i)) == 1
        barrier()
        t0 = time.perf_counter()
        for i in range(6, 6 + ka):
            assert alt.process_frame(frame(i)) == 1, "tracking lost (alternative mode)"
        barrier()
        tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        out["alt_mode"] = {"icp_shard_rows": (not a.icp_shard_rows), "frames": ka, "value": round(ka / float(tt.item()), 3), "unit": "frames/s"}
        alt.close()
    if rank == 0:
        if world == 1 and not a.no_s2:
            runner.close()
            out["roofline_s2"] = integrate_s2_probe(torch, capi, synth, 512)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(synth, N)
        elif world == 1:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
