"""Design aid (CPU, numpy): what k_classify_boxes decides for every wave-sized box of a frame of scene S1, and why a box is walked.

Re-states classify_box (x-slam_amd/csrc/xs_tsdf.hip) in float64 — near enough for counts — for a given volume edge N, frame and box shape,
and reports, for the boxes that see anything: wholly free / wholly empty / walked, the walked planes (what the integrate kernel's waves
issue), and the reason a plane is walked: the box's pixel range leaves the image ("edge": free space seen through the image border), or
the plane's depth range meets the truncation band of what the box can see ("band").  Variants answer "what would a finer class buy":
    --shape WX,WY,WZ   voxels of a box (default 32,2,8 = what one wave handles)
    --edge             planes that are free wherever they are in the image count as streamed (the EDGE class)
    --split-x K        a box's classes decided per K sub-boxes along x; a plane is walked if any sub-box walks it (wave-level union) ...
    --per-lane         ... or report the lane-planes walked instead (what a per-lane plane range would execute)
Usage: python profiles/tools/emulate_box_classes.py --n 1024 --frame 20 [--slack 2]
"""
import argparse
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
synth = importlib.import_module("x-slam_amd.synth")

TILE = 8


def tile_table(depth_mm):
    d = depth_mm.astype(np.float64)
    d = np.where((d > 5000) | (d < 200), 0.0, d / 1000.0)
    H, W = d.shape
    ty, tx = (H + TILE - 1) // TILE, (W + TILE - 1) // TILE
    lo = np.full((ty * TILE, tx * TILE), np.inf); hi = np.zeros((ty * TILE, tx * TILE)); inv = np.zeros((ty * TILE, tx * TILE))
    lo[:H, :W] = np.where(d > 0, d, np.inf); hi[:H, :W] = d; inv[:H, :W] = d == 0      # (round 5: lo over the VALID pixels + "holds an invalid one")
    lo = lo.reshape(ty, TILE, tx, TILE).min(axis=(1, 3)); hi = hi.reshape(ty, TILE, tx, TILE).max(axis=(1, 3)); inv = inv.reshape(ty, TILE, tx, TILE).max(axis=(1, 3))
    return lo, hi, inv


class Rmq2d:
    """range min / max over rectangles of a small 2-D table (sparse table, O(1) per query, vectorised)"""

    def __init__(self, a, op):
        self.op = op
        H, W = a.shape
        self.t = {}
        ky = 0
        rows = a
        while (1 << ky) <= H:
            kx = 0
            cur = rows
            while (1 << kx) <= W:
                self.t[(ky, kx)] = cur
                nxt = op(cur[:, : cur.shape[1] - (1 << kx)], cur[:, (1 << kx):]) if cur.shape[1] > (1 << kx) else None
                kx += 1
                if nxt is None: break
                cur = nxt
            if rows.shape[0] > (1 << ky): rows = op(rows[: rows.shape[0] - (1 << ky)], rows[(1 << ky):])
            else: break
            ky += 1

    def query(self, y0, y1, x0, x1):   # inclusive, arrays
        ky = np.floor(np.log2(y1 - y0 + 1)).astype(int); kx = np.floor(np.log2(x1 - x0 + 1)).astype(int)
        out = np.empty(y0.shape)
        for (a, b), tab in self.t.items():
            m = (ky == a) & (kx == b)
            if not m.any(): continue
            ya, yb, xa, xb = y0[m], y1[m] - (1 << a) + 1, x0[m], x1[m] - (1 << b) + 1
            out[m] = self.op(self.op(tab[ya, xa], tab[ya, xb]), self.op(tab[yb, xa], tab[yb, xb]))
        return out


def classify(N, frame, shape, slack_scale, split_x=1):
    prm = synth.s1_params(N)
    vs = prm["tsdf_voxel_size"]
    T = synth.s1_transforms(frame, prm)
    R = T["Rv2c"][..., 0].astype(np.float64); t = T["tv2c"][..., 0].astype(np.float64)
    fx, fy, cx, cy = synth.FX, synth.FY, synth.CX, synth.CY
    W, H = synth.WIDTH, synth.HEIGHT
    depth = synth.s1_frame(frame)
    lo_t, hi_t, inv_t = tile_table(depth)
    rmin, rmax, rinv = Rmq2d(lo_t, np.minimum), Rmq2d(hi_t, np.maximum), Rmq2d(inv_t, np.maximum)
    trunc = synth.tranc_dist(prm)
    band = trunc * 1.001 + 1e-5 + 2e-4
    WX, WY, WZ = shape
    sx = WX // split_x
    # slack (box_slack): 2e-3 (scale - 1) * magnitude, lateral x2, axial x0.3
    ext = N
    mag = lambda r: abs(t[r]) + np.abs(R[r]).sum() * vs * ext
    k = 2e-3 * (slack_scale - 1.0)
    dX, dY, dC = 2.0 * k * mag(0), 2.0 * k * mag(1), 0.3 * k * mag(2)
    nbx, nby, nbz = N // sx, N // WY, N // WZ
    # only the part of the volume the frustum can reach: z planes between the camera and the far limit
    dmax = depth.max() / 1000.0
    far = dmax * 1.0001 + 1.05 * trunc
    res = dict(free=0, empty=0, walk=0, walk_planes=0, walk_planes_edge=0, walk_planes_band=0, lane_planes=0, free_planes=0, boxes_seen=0)
    zc = np.arange(nbz)
    falls = R[2, 2] < 0
    for bz in zc:
        z0 = bz * WZ
        # quick reject of the slab of boxes by c range
        bx, by = np.meshgrid(np.arange(nbx), np.arange(nby), indexing="xy")
        bx = bx.ravel(); by = by.ravel()
        x0, y0 = bx * sx, by * WY
        cs = []; us = []; vsn = []
        for cz in (0, 1):
            for cyy in (0, 1):
                for cxx in (0, 1):
                    vx = ((x0 + (sx - 1) * cxx) + 0.5) * vs; vy = ((y0 + (WY - 1) * cyy) + 0.5) * vs; vz = ((z0 + (WZ - 1) * cz) + 0.5) * vs
                    X = R[0, 0] * vx + R[0, 1] * vy + R[0, 2] * vz + t[0]
                    Y = R[1, 0] * vx + R[1, 1] * vy + R[1, 2] * vz + t[1]
                    c = R[2, 0] * vx + R[2, 1] * vy + R[2, 2] * vz + t[2]
                    cs.append(c); us.append(X / np.where(c > 1e-3, c, 1.0)); vsn.append(Y / np.where(c > 1e-3, c, 1.0))
        cs = np.array(cs); us = np.array(us); vsn = np.array(vsn)
        cmin = cs.min(0) - dC; cmax = cs.max(0) + dC
        if cmax.max() < 0.05 or cmin.min() > far + 0.5: continue
        ok = cmin > 1e-3
        rcm = 1.0 / np.where(ok, cmin, 1.0)
        pad_u = 1.5 + abs(fx) * (dX + np.abs(us).max(0) * dC) * rcm * 1.01
        pad_v = 1.5 + abs(fy) * (dY + np.abs(vsn).max(0) * dC) * rcm * 1.01
        u = fx * us + cx; v = fy * vsn + cy
        ulo, uhi, vlo, vhi = u.min(0) - pad_u, u.max(0) + pad_u, v.min(0) - pad_v, v.max(0) + pad_v
        px0, px1, py0, py1 = np.floor(ulo), np.ceil(uhi), np.floor(vlo), np.ceil(vhi)
        inside = (px0 >= 2) & (py0 >= 2) & (px1 <= W - 2) & (py1 <= H - 2)
        qx0, qx1, qy0, qy1 = np.maximum(px0, 0), np.minimum(px1, W - 1), np.maximum(py0, 0), np.minimum(py1, H - 1)
        seen = ok & (qx0 <= qx1) & (qy0 <= qy1) & (cmin < far)
        if not seen.any(): continue
        idx = np.nonzero(seen)[0]
        tx0, tx1, ty0, ty1 = (qx0[idx] // TILE).astype(int), (qx1[idx] // TILE).astype(int), (qy0[idx] // TILE).astype(int), (qy1[idx] // TILE).astype(int)
        lo = rmin.query(ty0, ty1, tx0, tx1); hi = rmax.query(ty0, ty1, tx0, tx1)
        ins_clean = inside[idx] & (rinv.query(ty0, ty1, tx0, tx1) == 0)    # plain free space: the range in the image AND without an invalid pixel
        # per plane: c range of the plane's four corners
        c_first_lo = cs[:4, idx].min(0); c_first_hi = cs[:4, idx].max(0)
        dzc = R[2, 2] * vs
        j = np.arange(WZ)[:, None]
        pl_lo = c_first_lo[None, :] + j * dzc - dC - 1e-5; pl_hi = c_first_hi[None, :] + j * dzc + dC + 1e-5
        p_free_ok = (lo[None, :] - pl_hi > band)          # in front of everything the box can see
        p_empty = (pl_lo - hi[None, :] > band)
        ins = ins_clean[None, :]    # ("edge" below then also counts the SPECKLE reason: scene S1's frames have no invalid pixel)
        yield dict(bz=bz, idx=idx, bx=bx[idx], by=by[idx], p_free_ok=p_free_ok, p_empty=p_empty, inside=ins, falls=falls)


def summarise(N, frame, shape, slack, edge, split_x, per_lane):
    WX, WY, WZ = shape
    tot = dict(boxes=0, free=0, empty=0, walk=0, planes_walk=0, planes_edge=0, planes_band=0, planes_free=0, lane_planes=0)
    for s in classify(N, frame, shape, slack, split_x):
        pf, pe, ins, falls = s["p_free_ok"], s["p_empty"], s["inside"], s["falls"]
        free_now = pf & ins
        free_edge = pf & ~ins & ~pe             # free space seen through the image border
        # the kernel takes free planes from the near end and empty ones from the far end (runs); emulate runs
        order = np.arange(WZ)[::-1] if falls else np.arange(WZ)
        f = (free_now | (free_edge if edge else False))[order]
        e = pe[order]
        nfree = np.cumprod(f, axis=0).sum(0)
        nempty = np.cumprod(e[::-1], axis=0).sum(0)
        nempty = np.minimum(nempty, WZ - nfree)
        walk = WZ - nfree - nempty
        # reason: walked planes that would stream with the EDGE class
        f2 = (free_now | free_edge)[order]
        nfree2 = np.cumprod(f2, axis=0).sum(0)
        walk2 = WZ - nfree2 - np.minimum(nempty, WZ - nfree2)
        if split_x > 1:
            # union over the K sub-boxes of a wave-sized box: group by (by, bx // K)
            key = s["by"].astype(np.int64) * 100000 + (s["bx"] // split_x)
            uk, inv = np.unique(key, return_inverse=True)
            # per plane walked flags
            pos = np.arange(WZ)[:, None]
            walked_pl = (pos >= nfree[None, :]) & (pos < (WZ - nempty)[None, :])
            un = np.zeros((WZ, uk.size), bool)
            np.logical_or.at(un, (slice(None), inv), walked_pl)
            wave_walk = un.sum(0)
            tot["boxes"] += uk.size; tot["walk"] += int((wave_walk > 0).sum()); tot["planes_walk"] += int(wave_walk.sum())
            tot["lane_planes"] += int(walk.sum()) * (WX // split_x) * WY
            continue
        tot["boxes"] += walk.size
        tot["free"] += int((nfree == WZ).sum()); tot["empty"] += int((nempty == WZ).sum()); tot["walk"] += int((walk > 0).sum())
        tot["planes_walk"] += int(walk.sum()); tot["planes_free"] += int(nfree.sum())
        tot["planes_edge"] += int((walk - walk2).sum()); tot["planes_band"] += int(walk2.sum())
        tot["lane_planes"] += int(walk.sum()) * WX * WY
    return tot


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1024)
    ap.add_argument("--frame", type=int, default=20)
    ap.add_argument("--shape", default="32,2,8")
    ap.add_argument("--slack", type=float, default=2.0)
    ap.add_argument("--edge", action="store_true")
    ap.add_argument("--split-x", type=int, default=1)
    ap.add_argument("--per-lane", action="store_true")
    a = ap.parse_args()
    shape = tuple(int(v) for v in a.shape.split(","))
    t = summarise(a.n, a.frame, shape, a.slack, a.edge, a.split_x, a.per_lane)
    print(f"N={a.n} frame={a.frame} box={shape} slack={a.slack} edge={a.edge} split_x={a.split_x}")
    print(t)
