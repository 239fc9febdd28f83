"""GPU: the sharded C++ pipeline (z-slab volume + ICP row shards + raycast composite) against the
single-GPU pipeline, with all ranks inside one process (one thread per rank, shared GPU and
stream; collectives meet at a barrier — x-slam_amd/sharded.py LocalWorld).  Integrate, the
raycast composite and the maps must be bit-identical to the unsharded run; the ICP sums differ
only in the association of double additions."""
import importlib
import threading

import numpy as np
import pytest

from helpers import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available()
    return torch, importlib.import_module("x-slam_amd.pipeline"), importlib.import_module("x-slam_amd.sharded")


def run_world(torch, sh, prm, world, frames):
    lw = sh.LocalWorld(torch, world)
    shards = [sh.ShardedKinectFusion(prm, r, world, collective=lw.collective_for(r)) for r in range(world)]
    depth = [torch.from_numpy(synth.s1_frame(k).view(np.int16)).cuda() for k in frames]
    results = [None] * world
    errors = []

    def work(r):
        try:
            for d in depth:
                assert shards[r].process_frame(d) == 1
            results[r] = (shards[r].world2camera(), shards[r].last_U(), shards[r].last_hits(), shards[r].icp_log())
        except BaseException as e:  # noqa: BLE001
            errors.append(e)
            lw.barrier.abort()

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errors, errors
    return shards, results


# world 4 and 8 (north_star's rank count) at 128^3: 32- and 16-plane slabs whose 6-plane halos reach into both neighbours; eight orchestrators as
# threads of one process on one card (x-slam_amd/sharded.py LocalWorld) — every collective of the real run, met at a barrier
WORLDS = [(2, 96), (3, 96), (4, 128), (8, 128)]


@pytest.mark.parametrize("rows_sharded", [False, True], ids=["icp_replicated", "icp_row_shards"])
@pytest.mark.parametrize("world,n", WORLDS)
def test_sharded_equals_single(dev, world, n, rows_sharded):
    torch, pl, sh = dev
    prm = dict(synth.s1_params(n), icp_shard_rows=rows_sharded)
    frames = [0, 1, 2]
    single = pl.KinectFusion(prm)
    for k in frames:
        assert single.process_frame(torch.from_numpy(synth.s1_frame(k).view(np.int16)).cuda()) == 1
    sv, sw, sg = single.volume()
    shards, results = run_world(torch, sh, prm, world, frames)
    # every rank ends with the same pose; equal to the single-GPU pose up to the double-sum association
    for r in range(world):
        assert np.array_equal(results[r][0], results[0][0])
    assert np.allclose(results[0][0], single.world2camera(), rtol=0, atol=2e-7)
    if not rows_sharded:
        # every rank ran the single-GPU ICP on the same maps: same bits, no collective in the loop
        assert np.array_equal(results[0][0], single.world2camera())
        assert np.array_equal(results[0][3], single.icp_log())
    assert sum(res[1] for res in results) == single.last_U()
    assert results[0][2] == single.last_hits()
    # ICP: first iteration of the last frame sees identical inputs -> sums agree to double rounding
    a, b = results[0][3], single.icp_log()
    assert a.shape == b.shape and np.array_equal(a[:, 54], b[:, 54])
    assert np.allclose(a[0, :54], b[0, :54], rtol=1e-12, atol=1e-12 * np.abs(b[0, :54]).max())
    # volume: owned slabs tile the single-GPU volume
    if np.array_equal(results[0][0], single.world2camera()):
        pieces = [s.owned_volume() for s in shards]
        for i, full in enumerate((sv, sw, sg)):
            assert np.array_equal(np.concatenate([p[i] for p in pieces]), full)
        # halo planes carry the neighbour's exact bits
        v0, _, _ = shards[0].volume()
        o, st = shards[0].owned, shards[0].stored
        plane = n * n
        assert np.array_equal(v0[(o[1] - st[0]) * plane:], sv[o[1] * plane: st[1] * plane])
        # Gauss-Newton terms of the last frame against the map: every rank's slab, all-reduced, equals the single-GPU sums
        d_last = torch.from_numpy(synth.s1_frame(frames[-1]).view(np.int16)).cuda()
        want = single.gauss_newton_terms(d_last, single.camera2volume())
        gots = [None] * world
        def gn(r):
            gots[r] = shards[r].gauss_newton_terms(d_last, shards[r].camera2volume())
        th = [threading.Thread(target=gn, args=(r,)) for r in range(world)]
        [t.start() for t in th]; [t.join() for t in th]
        assert want is not None and want[28] > 100
        for r in range(world):
            assert gots[r][28] == want[28] and np.allclose(gots[r][:28], want[:28], rtol=1e-9, atol=1e-12 * np.abs(want[:28]).max())
        # surface export: every rank reports the crossings of the planes it owns; together they are the single-GPU cloud
        sp, sn = single.export_point_cloud(1000000)
        parts = [s.export_point_cloud(1000000) for s in shards]
        allp = np.concatenate([p for p, _ in parts]); alln = np.concatenate([q for _, q in parts])
        assert len(allp) == len(sp) > 0
        o1, o2 = np.lexsort((sp[:, 2], sp[:, 1], sp[:, 0])), np.lexsort((allp[:, 2], allp[:, 1], allp[:, 0]))
        assert np.array_equal(sp[o1], allp[o2])
        ok = ~np.isnan(sn[o1]) & ~np.isnan(alln[o2])
        assert np.array_equal(sn[o1][ok], alln[o2][ok])
    # composed previous-frame maps identical on every rank, and to the single-GPU maps when the poses are
    for which in ("vmaps_g_prev", "nmaps_g_prev"):
        m0 = shards[0].map(which, 0)
        for s in shards[1:]:
            assert np.array_equal(np.nan_to_num(m0, nan=-7.0), np.nan_to_num(s.map(which, 0), nan=-7.0))
    for s in shards:
        s.close()
    single.close()


@pytest.mark.parametrize("world,n", WORLDS)
def test_composite_by_gather_equals_the_sum_of_maps_and_halves_the_bytes(dev, world, n):
    """The raycast composite as an owner-compacted exchange (every rank packs the pixels it owns, the packs are gathered and scattered:
    shard_composite_gather, the default) against the int32 sum of the maps (round 3): the same poses, counts, ICP sums and composed maps on
    every rank, bit for bit, over six frames — and at most 0.58 of the bytes received, summed over the ranks and counting the min-key
    all-reduce both forms share (a ring all-reduce moves every pixel's 48 bytes twice, the gather each owned pixel's 52 once)."""
    torch, pl, sh = dev
    prm = synth.s1_params(n)
    frames = list(range(6))
    s_g, g = run_world(torch, sh, dict(prm, shard_composite_gather=True), world, frames)
    s_s, a = run_world(torch, sh, dict(prm, shard_composite_gather=False), world, frames)
    for r in range(world):
        assert np.array_equal(g[r][0], a[r][0]) and g[r][1] == a[r][1] and g[r][2] == a[r][2]
        assert np.array_equal(g[r][3], a[r][3])
        for which in ("vmaps_g_prev", "nmaps_g_prev"):
            for level in range(3):
                # every pixel with a value, all three planes (the y and z planes of a pixel without one are not written: they keep what the buffer held)
                x, y = s_g[r].map(which, level), s_s[r].map(which, level)
                rows_ = x.shape[0] // 3
                valid = np.isfinite(y[:rows_, :, 0])
                assert np.array_equal(valid, np.isfinite(x[:rows_, :, 0])), (r, which, level)
                for p_ in range(3):
                    assert np.array_equal(x[p_ * rows_:(p_ + 1) * rows_][valid].view(np.int32), y[p_ * rows_:(p_ + 1) * rows_][valid].view(np.int32)), (r, which, level, p_)
    assert g[0][2] > 0.5 * synth.HEIGHT * synth.WIDTH
    # bytes received, summed over the ranks (in this scene one rank owns nearly every hit: it receives next to nothing and the others all of
    # it, where the ring all-reduce loads every rank alike)
    bg, bs = sum(s.composite_bytes() for s in s_g), sum(s.composite_bytes() for s in s_s)
    # per pixel and summed over N ranks: the sum of the maps moves 2 (N - 1) x (4 + 48) bytes, the gather 2 (N - 1) x 4 (the keys both forms share)
    # + (N - 1) x 52 x the fraction of pixels with a hit: a ratio of 0.077 + 0.5 x hit fraction whatever N (0.55 at this scene's 0.95)
    assert 0 < bg <= 0.58 * bs, (bg, bs)
    for s in s_g + s_s:
        s.close()


@pytest.mark.parametrize("world", [3, 8])
def test_sharded_sign_map_changes_nothing(dev, world):
    """Every rank keeps a sign map of the planes it stores (owned slab + halo, marked by its own integrate calls) and its slab march
    evaluates only the iterations that map leaves: poses, counts and ICP sums of a three- and an eight-rank run with and without, bit for bit."""
    torch, pl, sh = dev
    prm = synth.s1_params(128)
    frames = list(range(6))
    s_on, on = run_world(torch, sh, dict(prm, raycast_sign_map=True), world, frames)
    s_off, off = run_world(torch, sh, dict(prm, raycast_sign_map=False), world, frames)
    for a_, b_ in zip(on, off):
        assert np.array_equal(a_[0], b_[0]) and a_[1] == b_[1] and a_[2] == b_[2]
        assert np.array_equal(a_[3], b_[3])
    assert on[0][2] > 0.5 * synth.HEIGHT * synth.WIDTH
    for x, y in zip(s_on, s_off):
        for u, v in zip(x.volume(), y.volume()):
            assert np.array_equal(u, v)
    for r in s_on + s_off:
        r.close()


def test_sharded_first_frame_bit_exact(dev):
    """Frame 0 has no ICP: the sharded integrate + raycast composite must reproduce the single-GPU
    volume and maps bit for bit."""
    torch, pl, sh = dev
    n = 128
    prm = synth.s1_params(n)
    single = pl.KinectFusion(prm)
    assert single.process_frame(torch.from_numpy(synth.s1_frame(0).view(np.int16)).cuda()) == 1
    shards, results = run_world(torch, sh, prm, 4, [0])
    sv, sw, sg = single.volume()
    pieces = [s.owned_volume() for s in shards]
    for i, full in enumerate((sv, sw, sg)):
        assert np.array_equal(np.concatenate([p[i] for p in pieces]), full)
    assert results[0][2] == single.last_hits()
    H = synth.HEIGHT
    for which in ("vmaps_g_prev", "nmaps_g_prev"):
        for level in range(3):
            a, b = shards[1].map(which, level), single.map(which, level)
            rows = H >> level
            nan_a, nan_b = np.isnan(a[:rows, :, 0]), np.isnan(b[:rows, :, 0])
            assert np.array_equal(nan_a, nan_b)
            ok = ~nan_b
            for p in range(3):
                assert np.array_equal(a[p * rows:(p + 1) * rows][ok], b[p * rows:(p + 1) * rows][ok]), (which, level, p)
    for s in shards:
        s.close()
    single.close()


def test_single_rank_composite_path(dev):
    """force_shard_composite: the slab raycast + key / map composite with one rank (collectives are
    identities) must reproduce the ordinary raycast bit for bit — the path bench.py --gpus N takes,
    exercised without a second GPU."""
    torch, pl, sh = dev
    n = 96
    prm = synth.s1_params(n)
    frames = [0, 1, 2]
    single = pl.KinectFusion(prm)
    calls = []

    def ident(_user, op, ptr, count):
        calls.append((op, count))
    forced = sh.ShardedKinectFusion(dict(prm, force_shard_composite=True), 0, 1, collective=ident)
    for k in frames:
        d = torch.from_numpy(synth.s1_frame(k).view(np.int16)).cuda()
        assert single.process_frame(d) == 1 and forced.process_frame(d) == 1
    assert np.array_equal(single.world2camera(), forced.world2camera())
    assert single.last_hits() == forced.last_hits() and single.last_U() == forced.last_U()
    for which in ("vmaps_g_prev", "nmaps_g_prev"):
        for level in range(3):
            a, b = single.map(which, level), forced.map(which, level)
            rows = 480 >> level   # the sentinel lives in the x plane; y / z planes are only defined where it is absent
            na, nb = np.isnan(a[:rows, :, 0]), np.isnan(b[:rows, :, 0])
            assert np.array_equal(na, nb)
            for p in range(3):
                assert np.array_equal(a[p * rows:(p + 1) * rows][~na], b[p * rows:(p + 1) * rows][~nb])
    # per frame: one min over W*H keys, the sum of the owned-pixel counts (one per rank), the gather of the packed owned pixels
    per_frame = calls[-3:]
    assert per_frame[0] == (1, 640 * 480) and per_frame[1] == (2, 1) and per_frame[2] == (3, 1)
    # ... and with shard_composite_gather: false, round 3's form: one sum over both level-0 model maps (contiguous allocation), same maps
    calls.clear()
    summed = sh.ShardedKinectFusion(dict(prm, force_shard_composite=True, shard_composite_gather=False), 0, 1, collective=ident)
    for k in frames:
        assert summed.process_frame(torch.from_numpy(synth.s1_frame(k).view(np.int16)).cuda()) == 1
    per_frame = calls[-2:]
    assert per_frame[0] == (1, 640 * 480) and per_frame[1][0] == 2 and per_frame[1][1] == 2 * 3 * 480 * (640 * 8 // 4)
    assert np.array_equal(summed.world2camera(), forced.world2camera()) and summed.last_hits() == forced.last_hits()
    single.close(); forced.close(); summed.close()


def test_sharded_checkpoint_roundtrip(dev, tmp_path):
    """Every rank of a sharded run saves its own planes (owned slab + halo) and a fresh set of ranks restores them: the
    restored run continues with the same bits as the one that never stopped; a rank refuses another rank's file and a
    single-GPU checkpoint."""
    torch, pl, sh = dev
    n, world = 96, 2
    prm = synth.s1_params(n)
    shards, _ = run_world(torch, sh, prm, world, [0, 1, 2])
    paths = [str(tmp_path / f"rank{r}.ckpt") for r in range(world)]
    for r in range(world):
        shards[r].save_checkpoint(paths[r])
    single = pl.KinectFusion(prm)
    assert single.process_frame(torch.from_numpy(synth.s1_frame(0).view(np.int16)).cuda()) == 1
    single.save_checkpoint(str(tmp_path / "single.ckpt"))
    lw = sh.LocalWorld(torch, world)
    fresh = [sh.ShardedKinectFusion(prm, r, world, collective=lw.collective_for(r)) for r in range(world)]
    assert not fresh[0].load_checkpoint(paths[1]) and not fresh[1].load_checkpoint(paths[0])
    assert not fresh[0].load_checkpoint(str(tmp_path / "single.ckpt"))
    d3 = torch.from_numpy(synth.s1_frame(3).view(np.int16)).cuda()
    out = {}
    errors = []

    def resume(r):
        try:
            assert fresh[r].load_checkpoint(paths[r])      # the restore re-raycasts the model maps: a collective, every rank takes part
            assert fresh[r].frame_id == 3
            assert fresh[r].process_frame(d3) == 1
            out[("fresh", r)] = (fresh[r].world2camera(), fresh[r].volume())
        except BaseException as e:  # noqa: BLE001
            errors.append(e); lw.barrier.abort()
    th = [threading.Thread(target=resume, args=(r,)) for r in range(world)]
    [t.start() for t in th]; [t.join() for t in th]
    assert not errors, errors
    # the original ranks take the same frame
    errors2 = []

    def go_on(r):
        try:
            assert shards[r].process_frame(d3) == 1
            out[("orig", r)] = (shards[r].world2camera(), shards[r].volume())
        except BaseException as e:  # noqa: BLE001
            errors2.append(e)
    th = [threading.Thread(target=go_on, args=(r,)) for r in range(world)]
    [t.start() for t in th]; [t.join() for t in th]
    assert not errors2, errors2
    for r in range(world):
        assert np.array_equal(out[("fresh", r)][0], out[("orig", r)][0])
        for x, y in zip(out[("fresh", r)][1], out[("orig", r)][1]):
            assert np.array_equal(x, y)
    for s_ in shards + fresh:
        s_.close()
    single.close()


def test_cpp_host_runs_shard_mode_over_rccl_without_python():
    """x-slam_amd/smoke_rccl (host/smoke_rccl.cpp): a C++ program that creates an RCCL communicator through
    libxslam_rccl.so (include/xslam_amd_rccl.h), hands xs_rccl_collective to xs_kf_create_sharded and tracks four frames
    of a synthetic room corner in shard mode — world = 1 here (one GPU), so the raycast composite's collectives (three per
    frame: min of the keys, sum of the owned-pixel counts, gather of the packed owned pixels) are single-rank RCCL calls on the orchestrator's stream; with N ranks (`smoke_rccl <rank> <N> <id-file>`,
    one process per GPU) the 12 per-frame ICP all-reduces join them."""
    import json
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "x-slam_amd", "smoke_rccl")
    if not os.path.exists(exe) and not os.path.exists("/opt/rocm/include/rccl/rccl.h"):
        pytest.skip("no RCCL in this ROCm install: the RCCL targets are optional (host/Makefile)")
    assert os.path.exists(exe), "smoke_rccl not built: run __graft_entry__.build()"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["tracked"] == 1 and out["collective_calls"] == 12 and out["rccl_version"] > 0 and len(out["world2camera"]) == 32


def _gpu_count():
    import torch
    return torch.cuda.device_count()      # (does not initialise the GPU on this image)


@pytest.mark.skipif(_gpu_count() < 2, reason="needs two GPUs: RCCL with more than one rank (switches itself on on a multi-GPU node)")
def test_two_ranks_over_rccl_on_two_gpus():
    """The first thing to run on a node with more than one GPU: (1) the C++ smoke binary as two processes, one per GPU, sharing the
    RCCL unique id through a file — ncclCommInitRank with count 2, the raycast composite's two collectives per frame and the twelve
    440-byte ICP all-reduces per tracked frame; (2) bench.py --gpus 2 --native-rccl (it starts its own two ranks): both ranks seen,
    tracking holds on every rank, the N > 1 `scaling` block is there."""
    import json
    import os
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "x-slam_amd", "smoke_rccl")
    assert os.path.exists(exe)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    with tempfile.TemporaryDirectory() as tmp:
        idf = os.path.join(tmp, "rccl_id")
        procs = [subprocess.Popen([exe, str(r), "2", idf], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in range(2)]
        outs = [p.communicate(timeout=600) for p in procs]
    poses = []
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, (p.returncode, so[-1500:], se[-1500:])
        o = json.loads([l for l in so.splitlines() if l.startswith("{")][-1])
        assert o["count"] == 2 and o["tracked"] == 1 and o["collective_calls"] == 4 * 3 + 3 * 12
        poses.append(np.array(o["world2camera"], np.float32))
    # both ranks hold the same pose, bit for bit, and it is the one-GPU run's up to the association of the all-reduced double sums
    one = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert one.returncode == 0, (one.stdout[-1500:], one.stderr[-1500:])
    ref = np.array(json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])["world2camera"], np.float32)
    assert np.array_equal(poses[0].view(np.int32), poses[1].view(np.int32))
    assert np.allclose(poses[0], ref, rtol=0, atol=2e-7)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "3", "--native-rccl", "--size", "256",
                        "--reloc-size", "256", "--no-alt"], capture_output=True, text=True, timeout=1200, env=env, cwd=root)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_ranks_seen"] == 2 and out["n_gpus"] == 2 and out["value"] > 0
    assert "C++ RCCL" in out.get("collectives", "")
    assert set(out["scaling_vs_one_gpu"]) >= {"reloc", "hessian"} and all(v["speedup"] > 0 for v in out["scaling_vs_one_gpu"].values())
