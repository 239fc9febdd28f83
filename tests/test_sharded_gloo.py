"""CPU, world_size 2 over gloo: the multi-GPU protocol of x-slam_amd/sharded.py — z-slab bounds and
halo, pixel-row shards, the 55-double all-reduce, and the raycast composite (min of first-event
keys, then the gather of every rank's packed owned pixels) — driven with the CPU oracle's kernels in place of the HIP
ones (there is no GPU here).  Sharded and unsharded runs of the same protocol must agree: bit
for bit on the volume and the composed maps, to double rounding on the ICP sums."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ITERS = [5, 4, 3]  # icp_iterations indexed by level (KinectFusionReconstruction.cpp:54-55)


def cmul3(a, b):
    """3x3 complex product, inner index ascending: (a0*b0 + a1*b1) + a2*b2."""
    out = np.zeros((3,) + b.shape[1:], np.complex64)
    for i in range(3):
        out[i] = (a[i, 0] * b[0] + a[i, 1] * b[1]) + a[i, 2] * b[2]
    return out


def to_c(x):
    return (x[..., 0] + 1j * x[..., 1]).astype(np.complex64)


def to_f(x):
    return np.stack([x.real, x.imag], -1).astype(np.float32)


class ProtocolKinFu:
    """The sharded per-frame flow of KinectFusionReconstruction.cpp (shard mode), in Python over
    oracle kernels; rank/world/dist select the shard and the collectives."""

    def __init__(self, oracle, sh, capi, synth, prm, rank, world, dist):
        self.o, self.sh, self.capi, self.synth, self.prm = oracle, sh, capi, synth, prm
        self.rank, self.world, self.dist = rank, world, dist
        n = prm["tsdf_size_x"]
        self.res = [n, n, n]
        self.owned, self.stored = sh.slab_bounds(rank, world, n)
        self.value, self.weight, self.grad = oracle.new_volume(self.res)
        self.intr = np.array([prm["fx"], prm["fy"], prm["cx"], prm["cy"]], np.float32)
        vs = np.float32(prm["tsdf_voxel_size"])
        self.trunc = float(max(np.float32(vs * np.float32(prm["thres_range"])), np.float32(np.float32(2.1) * vs)))
        w2c = np.zeros((4, 4, 2), np.float32)
        w2c[[0, 1, 2, 3], [0, 1, 2, 3], 0] = 1
        w2c[0, 3, 1] = 1e-7
        self.record = [w2c]
        self.w2v = np.zeros((4, 4, 2), np.float32)
        self.w2v[[0, 1, 2, 3], [0, 1, 2, 3], 0] = 1
        self.w2v[:3, 3, 0] = [prm["init_x"], prm["init_y"], prm["init_z"]]
        self.angle = float(np.sin(np.float32(prm["angleThres"]) / np.float32(180.0) * np.pi))
        self.prev_v, self.prev_n = [None] * 3, [None] * 3
        self.frame_id = 0
        self.violations = 0
        self.U = 0
        self.bytes_sum_form = self.bytes_gather_form = 0.0

    def allreduce(self, op, arr):
        if self.world == 1:
            return arr
        import torch
        t = torch.from_numpy(arr)
        self.sh.reduce_tensor(self.dist, op, t)
        return arr

    def level_intr(self, l):
        return (self.intr / np.float32(1 << l)).astype(np.float32)

    def process(self, depth):
        o = self.o
        d = [o.bilateral(depth)]
        for l in (1, 2):
            d.append(o.pyr_down(d[l - 1]))
        cv = [o.create_vmap(self.level_intr(l), d[l]) for l in range(3)]
        cn = [o.create_nmap(v) for v in cv]
        if self.frame_id > 0:
            c2w_prev = o.m4_inverse(self.record[-1])
            Rprev, tprev = c2w_prev[:3, :3], c2w_prev[:3, 3]
            Rprev_inv = o.m3_inverse(Rprev)
            Rcurr, tcurr = to_c(Rprev), to_c(tprev)
            c2w = c2w_prev.copy()
            for level in (2, 1, 0):
                rows = self.synth.HEIGHT >> level
                y0, y1 = self.sh.row_bounds(self.rank, self.world, rows)
                for _ in range(ITERS[level]):
                    sums, _, _, inl = o.icp_combined(to_f(Rcurr), to_f(tcurr), cv[level], cn[level], Rprev_inv, tprev, self.level_intr(level),
                                                     self.prev_v[level], self.prev_n[level], self.prm["distThres"], self.angle, y0=y0, y1=y1)
                    buf = np.concatenate([sums, [float(inl)]])
                    buf = self.allreduce(self.sh.OP_SUM_F64, buf)
                    A, b = self.capi.icp_unpack(buf[:54])
                    assert abs(o.det6_real(A)) > 1e-15
                    x = o.llt_solve6(A, b).astype(np.float32)
                    Rinc = to_c(o.rinc(x[0], x[1], x[2]))
                    tcurr = cmul3(Rinc, tcurr) + to_c(x[3:6])
                    Rcurr = cmul3(Rinc, Rcurr)
                    c2w[:3, :3] = to_f(Rcurr)
                    c2w[:3, 3] = to_f(tcurr)
                    c2w[3, 3] = [1, 0]
            self.record.append(o.m4_inverse(c2w))
        w2c = self.record[-1]
        c2w = o.m4_inverse(w2c)
        c2v = o.m4_mul(self.w2v, c2w)
        v2c = o.m4_inverse(c2v)
        ds = o.scale_depth(depth)
        # owned planes (counted) and the halo bands: disjoint ranges of the stored planes
        self.U = 0
        for i, (za, zb) in enumerate(((self.owned[0], self.owned[1]), (self.stored[0], self.owned[0]), (self.owned[1], self.stored[1]))):
            if zb > za:
                u = o.integrate(ds, self.value, self.weight, self.grad, self.res, self.trunc, 100, v2c[:3, :3], v2c[:3, 3], self.intr,
                                self.prm["tsdf_voxel_size"], 0.0, z0=za, z1=zb)
                self.U += u if i == 0 else 0
        v2w = o.m4_inverse(self.w2v)
        H, W = self.synth.HEIGHT, self.synth.WIDTH
        vm, nm, keys, bad = o.raycast_slab(self.intr, c2v[:3, :3], c2v[:3, 3], v2w[:3, :3], v2w[:3, 3], self.trunc, self.res,
                                           self.prm["tsdf_voxel_size"], self.value, self.grad, H, W, self.stored, self.owned)
        self.violations += bad
        mk = self.allreduce(self.sh.OP_MIN_I32, keys.copy())
        mine = ((keys == mk) & ((keys & 1) == 0) & (keys != 0x7FFFFFFF)).reshape(H, W)
        for m in (vm, nm):
            for p in range(3):
                m[p * H:(p + 1) * H][~mine] = 0
        # the owner-compacted exchange (KinectFusionReconstruction::CalculatePointCloud, shard mode): every rank packs the pixels it owns —
        # {pixel index, vertex x / y / z (re, im), normal x / y / z (re, im)}: 13 words —, the counts travel as an int32 sum with one non-zero
        # entry per rank, the packs are gathered at the offsets every rank derives from the counts, and a scatter writes them into the maps
        ring = 2.0 * (self.world - 1) / self.world if self.world > 1 else 0.0
        self.bytes_sum_form += ring * (keys.nbytes + vm.nbytes + nm.nbytes)           # what round 3's form (an int32 sum of both maps) moved
        self.bytes_gather_form += ring * keys.nbytes
        if self.world > 1:
            import torch
            idx = np.flatnonzero(mine.reshape(-1)).astype(np.int32)
            ys, xs = idx // W, idx % W
            pack = np.empty((len(idx), 13), np.int32)
            pack[:, 0] = idx
            for q in range(3):
                pack[:, 1 + 2 * q:3 + 2 * q] = vm[q * H:(q + 1) * H][ys, xs].view(np.int32)
                pack[:, 7 + 2 * q:9 + 2 * q] = nm[q * H:(q + 1) * H][ys, xs].view(np.int32)
            counts = np.zeros(self.world, np.int32)
            counts[self.rank] = len(idx)
            counts = self.allreduce(self.sh.OP_SUM_I32, counts)
            off = [0] + [int(v) for v in np.cumsum(counts.astype(np.int64)) * 52]
            buf = torch.zeros(max(off[-1], 1), dtype=torch.uint8)
            buf[off[self.rank]:off[self.rank + 1]] = torch.from_numpy(pack.reshape(-1).view(np.uint8).copy())
            self.sh.gatherv_tensor(self.dist, torch, buf, off)
            self.bytes_gather_form += ring * counts.nbytes + (off[-1] - (off[self.rank + 1] - off[self.rank]))
            got = buf[:off[-1]].numpy().view(np.int32).reshape(-1, 13)
            assert len(np.unique(got[:, 0])) == len(got)          # every pixel has one owner at most
            gy, gx = got[:, 0] // W, got[:, 0] % W
            for q in range(3):
                vm[q * H:(q + 1) * H][gy, gx] = got[:, 1 + 2 * q:3 + 2 * q].view(np.float32)
                nm[q * H:(q + 1) * H][gy, gx] = got[:, 7 + 2 * q:9 + 2 * q].view(np.float32)
        nohit = (((mk & 1) == 1) | (mk == 0x7FFFFFFF)).reshape(H, W)
        qnan = np.array([0x7FFFFFFF], np.uint32).view(np.float32)[0]
        for m in (vm, nm):
            m[:H][nohit] = [qnan, 0]
        self.hits = int((~nohit).sum())
        self.prev_v[0], self.prev_n[0] = vm, nm
        for l in (1, 2):
            self.prev_v[l] = o.resize_map(self.prev_v[l - 1], False)
            self.prev_n[l] = o.resize_map(self.prev_n[l - 1], True)
        self.frame_id += 1


def _worker(rank, world, port, n, nframes, threads=2):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.oracle import Oracle, OracleKinFu, params_from_dict
    synth = importlib.import_module("x-slam_amd.synth")
    sh = importlib.import_module("x-slam_amd.sharded")
    capi = importlib.import_module("x-slam_amd.capi")
    o = Oracle()
    o._set_num_threads(threads)
    prm = synth.s1_params(n)
    frames = [synth.s1_frame(k) for k in range(nframes)]
    whole = ProtocolKinFu(o, sh, capi, synth, prm, 0, 1, None)
    part = ProtocolKinFu(o, sh, capi, synth, prm, rank, world, dist)
    for d in frames:
        whole.process(d)
        part.process(d)
    assert part.violations == 0, "a slab raycast read outside its stored planes: halo too thin"
    # poses: identical inputs, sums differ only by the association of double additions
    assert np.allclose(part.record[-1], whole.record[-1], rtol=0, atol=2e-7)
    same_pose = np.array_equal(part.record[-1], whole.record[-1])
    # hits and U add up
    import torch
    u = torch.tensor([part.U], dtype=torch.int64)
    dist.all_reduce(u)
    # the composite's bytes, summed over the ranks: the gathered packs against round 3's sum of the maps
    bb = torch.tensor([part.bytes_gather_form, part.bytes_sum_form], dtype=torch.float64)
    dist.all_reduce(bb)
    assert 0 < bb[0].item() <= 0.55 * bb[1].item(), bb
    if same_pose:
        assert int(u.item()) == whole.U and part.hits == whole.hits
        plane = n * n
        a, b = part.stored[0] * plane, part.stored[1] * plane
        for x, y in ((part.value, whole.value), (part.weight, whole.weight), (part.grad, whole.grad)):
            assert np.array_equal(x[a:b], y[a:b])  # owned + halo planes carry the unsharded bits
        for l in range(3):
            for x, y in ((part.prev_v[l], whole.prev_v[l]), (part.prev_n[l], whole.prev_n[l])):
                assert np.array_equal(x.view(np.int32), y.view(np.int32))
    # the unsharded protocol run itself agrees with the oracle's C++ pipeline
    kf = OracleKinFu(o, params_from_dict(prm))
    for d in frames:
        assert kf.process_frame(d) == 1
    assert np.allclose(whole.record[-1], kf.world2camera(), rtol=0, atol=5e-7)
    assert whole.U == kf.last_U() or not np.array_equal(whole.record[-1], kf.world2camera())
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(600)
def test_sharded_protocol_world2_gloo(oracle):
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(2, _free_port(), 48, 2), nprocs=2, join=True)


@pytest.mark.timeout(900)
def test_sharded_protocol_world4_gloo(oracle):
    """Four ranks (12-plane slabs of a 48^3 volume: every inner rank's 6-plane halos reach half way into both neighbours), each a process,
    gloo collectives: min-key all-reduce, counts, the four-way gather-v of the owned pixels, the 440-byte ICP all-reduce."""
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(4, _free_port(), 48, 2), nprocs=4, join=True)


@pytest.mark.timeout(1200)
def test_sharded_protocol_world8_gloo(oracle):
    """Eight ranks, each a process (the shape of the driver's 8-GPU launch; the GPU tests reach world 8 only as threads of one process): a
    128-plane volume in 16-plane slabs, so every inner rank's 6-plane halos lie wholly inside both neighbours; the eight-way rendezvous, the
    min-key all-reduce, the eight-part gather-v descriptor of the owned pixels, the ICP all-reduce over eight pixel-row shards (60 / 30 / 15 rows
    per rank and level).  One oracle thread per rank: eight processes on the container's eight cores."""
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(8, _free_port(), 128, 2, 1), nprocs=8, join=True)


def test_slab_and_row_bounds_tile():
    sh = importlib.import_module("x-slam_amd.sharded")
    for world in (1, 2, 3, 4, 8):
        for Z in (48, 90, 512, 1024):
            own = [sh.slab_bounds(r, world, Z)[0] for r in range(world)]
            assert own[0][0] == 0 and own[-1][1] == Z and all(own[i][1] == own[i + 1][0] for i in range(world - 1))
            for r in range(world):
                (z0, z1), (s0, s1) = sh.slab_bounds(r, world, Z)
                assert 0 <= s0 <= z0 < z1 <= s1 <= Z and (world == 1 or (z0 - s0 in (0, sh.HALO) and s1 - z1 in (0, sh.HALO)))
        for rows in (120, 240, 480):
            rb = [sh.row_bounds(r, world, rows) for r in range(world)]
            assert rb[0][0] == 0 and rb[-1][1] == rows and all(rb[i][1] == rb[i + 1][0] for i in range(world - 1))


def _hess_worker(rank, world, port):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.oracle import Oracle
    from conftest import load_golden
    from helpers import intr_of, tranc_dist
    synth = importlib.import_module("x-slam_amd.synth")
    sh = importlib.import_module("x-slam_amd.sharded")
    o = Oracle()
    o._set_num_threads(2)
    gd = load_golden("hessian_s1_n64.npz")
    n = 64
    prm = synth.s1_params(n)
    # gt = TSDF after frame 0 (as in the fixture)
    from oracle.oracle import OracleKinFu, params_from_dict
    kf = OracleKinFu(o, params_from_dict(prm))
    kf.process_frame(synth.s1_frame(0))
    gt, _, _ = kf.volume()
    ds = o.scale_depth(synth.s1_frame(1))
    (z0, z1), _ = sh.slab_bounds(rank, world, n)
    part = o.tsdf_hessian(ds, [n, n, n], prm["tsdf_voxel_size"], gd["R_a"], gd["t_a"], tranc_dist(prm), intr_of(prm), gt, z0=z0, z1=z1)
    t = torch.from_numpy(part.copy())
    sh.reduce_tensor(dist, sh.OP_SUM_F64, t)
    want = gd["hess_a"]
    assert t[3].item() == want[3] and np.allclose(t.numpy(), want, rtol=1e-10)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_sharded_hessian_world2_gloo(oracle):
    """BASELINE config 4 protocol: per-slab dual-complex Hessian sums + one all-reduce == whole volume."""
    import torch.multiprocessing as mp
    mp.spawn(_hess_worker, args=(2, _free_port()), nprocs=2, join=True)
