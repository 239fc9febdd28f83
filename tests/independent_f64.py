"""A second, independently written checker for the hot-path kernels — TEST INFRASTRUCTURE ONLY.

Everything the parity tests otherwise rest on (oracle/, the fixtures under tests/golden/pipeline_*) is one
restatement of the reference's kernels in complex<float>, i.e. the same formulas in the same arithmetic as the
HIP kernels.  This module shares no text with it.  It was written from the reference's kernel sources alone
(file:line cited per function), in real float64 numpy, vectorised over voxels / pixels, and it never evaluates a
complex number: a quantity carried as (re, im) by the CSFD kernels is taken here as the real function
x(d) = re + d * im / h of the seed offset d, and derivatives come from central differences
(f(+d) - f(-d)) / 2d  (second order: (f(+d) - 2 f(0) + f(-d)) / d^2) of that real function, with every discrete
decision of the kernel — pixel picks, bounds and truncation tests, the zero-crossing step, trilinear cells, ICP
gates — taken once at d = 0 and held (the complex-step kernels decide on real parts only, so that is the function
they differentiate).  What the GPU tests assert with it:
  values       GPU real parts == this model in float64, to float32 rounding of the formula at hand;
  derivatives  GPU imaginary parts / h == the central difference, i.e. "gradients (and Hessians) fall out of the
               imaginary parts" is checked against a method that is not complex-step differentiation.
"""
import numpy as np

F32 = np.float32


def lin(c, d, h):
    """(..., 2) packed (re, im) float32 -> float64 value of the real function at seed offset d."""
    c = np.asarray(c, np.float64)
    return c[..., 0] + (d / h) * c[..., 1]


def _gather2d(img, y, x):
    return img[np.clip(y, 0, img.shape[0] - 1), np.clip(x, 0, img.shape[1] - 1)]


# ------------------------------------------------------------------------------------------------
# TSDF integrate — tsdfFusionKernal, XKinectFusion/src/TsdfFusion.cu:85-171
def integrate(d, h, Rv2c, tv2c, xyz, depth_m, intr, voxel_size, trunc, threshold, prev_value, prev_grad, prev_weight, dec=None):
    """One voxel update for the voxels xyz [n, 3] (integer coordinates).  Returns (new_value, dec); dec holds the
    decisions (taken when dec is None, which needs d == 0) incl. dec['update'] = voxels the kernel writes."""
    R = lin(np.asarray(Rv2c).reshape(3, 3, 2), d, h)
    t = lin(np.asarray(tv2c).reshape(3, 2), d, h)
    fx, fy, cx, cy = (float(F32(v)) for v in intr)
    vs, tr = float(F32(voxel_size)), float(F32(trunc))
    rows, cols = depth_m.shape
    vg = (np.asarray(xyz, np.float64) + 0.5) * vs                       # :108-111 voxel centre
    vc = vg @ R.T + t                                                    # :112
    inv_z = 1.0 / vc[:, 2]                                               # :113
    ix = vc[:, 0] * fx * inv_z + cx                                      # :116-117
    iy = vc[:, 1] * fy * inv_z + cy
    if dec is None:
        assert d == 0.0
        front = ~(inv_z < 0)                                             # :114-115
        with np.errstate(invalid="ignore"):
            coox = np.floor(ix - 0.5).astype(np.int64)                   # :118-119 __float2int_rd
            cooy = np.floor(iy - 0.5).astype(np.int64)
        inside = front & (coox > 1) & (cooy > 1) & (coox < cols - 1) & (cooy < rows - 1)   # :121-122
        nx, ny = np.rint(ix).astype(np.int64), np.rint(iy).astype(np.int64)                 # :123-124 __float2int_rn
        dd = [_gather2d(depth_m, cooy + j, coox + i).astype(np.float64) for j in (0, 1) for i in (0, 1)]  # d00 d10 d01 d11
        gmax, gmin = np.maximum.reduce(dd), np.minimum.reduce(dd)
        bil = (gmax - gmin < float(F32(threshold))) & (dd[0] != 0) & (dd[1] != 0) & (dd[2] != 0) & (dd[3] != 0)  # :133
        dec = dict(inside=inside, coox=coox, cooy=cooy, nx=nx, ny=ny, bil=bil)
    coox, cooy = dec["coox"], dec["cooy"]
    d00, d10 = _gather2d(depth_m, cooy, coox).astype(np.float64), _gather2d(depth_m, cooy, coox + 1).astype(np.float64)
    d01, d11 = _gather2d(depth_m, cooy + 1, coox).astype(np.float64), _gather2d(depth_m, cooy + 1, coox + 1).astype(np.float64)
    a, b = ix - (coox + 0.5), iy - (cooy + 0.5)                          # :135-136
    inter = d00 * (1 - a) * (1 - b) + d10 * a * (1 - b) + d01 * (1 - a) * b + d11 * a * b   # :137
    Dp = np.where(dec["bil"], inter, _gather2d(depth_m, dec["ny"], dec["nx"]).astype(np.float64))
    xl, yl = (ix - cx) / fx, (iy - cy) / fy                              # :142-143
    sdf = np.sqrt((Dp * xl) ** 2 + (Dp * yl) ** 2 + Dp ** 2) - np.sqrt((vc ** 2).sum(-1))  # :144-147
    if "update" not in dec:
        dec["update"] = dec["inside"] & (Dp > 0) & (sdf >= -tr)          # :148
        dec["far"] = sdf > tr                                            # :152
    tsdf = np.where(dec["far"], 1.0, sdf / tr)                           # :151-156 (the constant carries no derivative)
    w = np.asarray(prev_weight, np.float64)
    prev = np.asarray(prev_value, np.float64) + (d / h) * np.asarray(prev_grad, np.float64)
    return (prev * w + tsdf) / (w + 1.0), dec                            # :163


# ------------------------------------------------------------------------------------------------
# Raycast — RayCaster::operator(), XKinectFusion/src/RayCaster.cu:197-310 (+ :57-141 helpers)
def _cells(p, vs, res):
    """interpolateTrilineary's cell pick (:96-117) for points p [n, 3]: (lower cell [n, 3], on_border [n])."""
    with np.errstate(invalid="ignore"):
        g = np.floor(p / vs).astype(np.int64)                            # getVoxel :82-88
    r = np.asarray(res, np.int64)[None, :]
    border = ((g <= 0) | (g >= r - 1)).any(-1) | ~np.isfinite(p).all(-1)
    centre = (g + 0.5) * vs
    g = g - (p <= centre)                                                # :110-115: -(sgn(v - p) + 1) >> 1 is -1 for p <= v
    return g, border


def _trilinear(p, vol, cells, vs):
    """:117-136 on the (already offset) volume vol[z, y, x] for fixed lower cells."""
    gx, gy, gz = (np.clip(cells[:, i], 0, vol.shape[2 - i] - 2) for i in range(3))
    a0 = (p[:, 0] - (cells[:, 0] + 0.5) * vs) / vs
    b0 = (p[:, 1] - (cells[:, 1] + 0.5) * vs) / vs
    c0 = (p[:, 2] - (cells[:, 2] + 0.5) * vs) / vs
    a1, b1, c1 = 1 - a0, 1 - b0, 1 - c0
    V = lambda i, j, k: vol[gz + k, gy + j, gx + i]
    return (V(0, 0, 0) * a1 * b1 * c1 + V(0, 0, 1) * a1 * b1 * c0 + V(0, 1, 0) * a1 * b0 * c1 + V(0, 1, 1) * a1 * b0 * c0 +
            V(1, 0, 0) * a0 * b1 * c1 + V(1, 0, 1) * a0 * b1 * c0 + V(1, 1, 0) * a0 * b0 * c1 + V(1, 1, 1) * a0 * b0 * c0)


def raycast(d, h, intr, Rc2v, tc2v, Rv2w, tv2w, trunc, res, voxel_size, value, grad, px, py, dec=None):
    """Rays through pixels (px, py).  value / grad: [Z, Y, X] float arrays.  Returns (vertex_w [n, 3], normal_w [n, 3], dec);
    rows of pixels without a vertex (dec['hit'] False) / without a normal (dec['has_normal'] False) are NaN."""
    fx, fy, cx, cy = (float(F32(v)) for v in intr)
    vs = float(F32(voxel_size))
    step = float(F32(trunc) * F32(0.8))                                  # raycast() :350
    Rc = lin(np.asarray(Rc2v).reshape(3, 3, 2), d, h)
    tc = lin(np.asarray(tc2v).reshape(3, 2), d, h)
    Rw = lin(np.asarray(Rv2w).reshape(3, 3, 2), d, h)
    tw = lin(np.asarray(tv2w).reshape(3, 2), d, h)
    n = len(px)
    # readTsdf :69-78: stored value (+ i grad) + 1e-5
    vol = np.asarray(value, np.float64) + (d / h) * np.asarray(grad, np.float64) + float(F32(1e-5))
    cam = np.stack([(np.asarray(px, np.float64) - cx) / fx, (np.asarray(py, np.float64) - cy) / fy, np.ones(n)], -1)  # :57-63
    nxt = cam @ Rc.T + tc                                                # :207
    dirv = nxt - tc
    dirv = dirv / np.sqrt((dirv ** 2).sum(-1, keepdims=True))            # :208 normalized
    dirv = np.where(dirv == 0, 1e-15, dirv)                              # :210-212
    r = np.asarray(res, np.int64)
    if dec is None:
        assert d == 0.0
        t_curr = F32(0.2)                                                # :222-226
        g = np.floor((tc + dirv * float(t_curr)) / vs).astype(np.int64)
        g = np.clip(g, 0, r[None, :] - 1)                                # :227-230
        tsdf = vol[g[:, 2], g[:, 1], g[:, 0]]
        alive = np.ones(n, bool)
        cross_t = np.full(n, np.nan)
        cross_tn = np.full(n, np.nan)
        while t_curr < F32(5.0):                                         # :236, times accumulate in float
            t_next = F32(t_curr + F32(step))
            p = tc + dirv * float(t_next)                                # :238
            g = np.floor(p / vs).astype(np.int64)
            inside = ((g >= 0) & (g < r[None, :])).all(-1)               # checkInds :65-68
            alive &= inside                                              # :240-241
            gc = np.clip(g, 0, r[None, :] - 1)
            new = vol[gc[:, 2], gc[:, 1], gc[:, 0]]
            back = alive & (tsdf < 0) & (new > 0)                        # :245-246
            front = alive & (tsdf > 0) & (new < 0)                       # :247
            cross_t[front], cross_tn[front] = float(t_curr), float(t_next)
            alive &= ~(back | front)
            tsdf = np.where(alive, new, tsdf)
            t_curr = F32(t_curr + F32(step))
            if not alive.any():
                break
        cand = ~np.isnan(cross_t)
        p1, p0 = tc + dirv * cross_tn[:, None], tc + dirv * cross_t[:, None]
        c1, b1 = _cells(p1, vs, r)
        c0, b0 = _cells(p0, vs, r)
        dec = dict(cand=cand, t=cross_t, tn=cross_tn, c1=c1, c0=c0)
        ok = cand & ~b1 & ~b0                                            # :251-252, :256-257 (NaN from the border test)
        Ftdt, Ft = _trilinear(p1, vol, c1, vs), _trilinear(p0, vol, c0, vs)
        ok &= ~((Ft < 0) | (Ftdt > 0))                                   # :259-260
        dec["hit"] = ok
    hit = dec["hit"]
    t0 = np.where(hit, dec["t"], 0.0)
    tn = np.where(hit, dec["tn"], 0.0)
    Ftdt = _trilinear(tc + dirv * tn[:, None], vol, dec["c1"], vs)
    Ft = _trilinear(tc + dirv * t0[:, None], vol, dec["c0"], vs)
    with np.errstate(invalid="ignore", divide="ignore"):
        Ts = t0 - step * (Ft / (Ftdt - Ft))                              # :258, :261
    vertex = tc + dirv * Ts[:, None]                                     # :263
    vertex_w = vertex @ Rw.T + tw                                        # :264
    half = float(F32(voxel_size) * F32(0.5))                             # :274
    if "ncells" not in dec:
        with np.errstate(invalid="ignore"):
            g = np.floor(vertex / vs)
        okn = hit & ((g > 1) & (g < r[None, :] - 2)).all(-1)             # :269-271
        ncells, nb = [], np.zeros(n, bool)
        for axis in range(3):
            for sgn in (+1, -1):
                q = vertex.copy(); q[:, axis] += sgn * half
                c, b = _cells(q, vs, r)
                ncells.append(c); nb |= b
        dec["ncells"], dec["has_normal"] = ncells, okn & ~nb              # a NaN tap makes the normal NaN (:276-304)
    nrm = np.zeros((n, 3))
    k = 0
    for axis in range(3):
        taps = []
        for sgn in (+1, -1):
            q = vertex.copy(); q[:, axis] += sgn * half
            taps.append(_trilinear(q, vol, dec["ncells"][k], vs)); k += 1
        nrm[:, axis] = taps[0] - taps[1]                                 # :278, :286, :294
    with np.errstate(invalid="ignore", divide="ignore"):
        nrm = nrm / np.sqrt((nrm ** 2).sum(-1, keepdims=True))           # :301 normalized
    normal_w = nrm @ Rw.T                                                # :301
    vertex_w = np.where(hit[:, None], vertex_w, np.nan)
    normal_w = np.where(dec["has_normal"][:, None], normal_w, np.nan)
    return vertex_w, normal_w, dec


# ------------------------------------------------------------------------------------------------
# ICP rows and normal equations — Combined::search_newton / operator(), XKinectFusion/src/ICP.cu:196-281
def icp_normal_equations(d, h, Rcurr, tcurr, vmap_curr, nmap_curr, Rprev_inv, tprev, intr, vmap_prev, nmap_prev, dist_thres,
                         angle_thres, dec=None):
    """All pixels of one pyramid level.  Maps: [3 * rows, cols, 2].  Returns (sums[27] in the kernel's order
    i = 0..5, j = i..6 of row_i * row_j, inliers, dec)."""
    fx, fy, cx, cy = (float(F32(v)) for v in intr)
    rows, cols = vmap_curr.shape[0] // 3, vmap_curr.shape[1]
    planes = lambda m: lin(np.asarray(m).reshape(3, rows, cols, 2), d, h).reshape(3, -1)
    vc, nc, vp, npv = planes(vmap_curr), planes(nmap_curr), planes(vmap_prev), planes(nmap_prev)
    Rc = lin(np.asarray(Rcurr).reshape(3, 3, 2), d, h)
    tc = lin(np.asarray(tcurr).reshape(3, 2), d, h)
    Rpi = lin(np.asarray(Rprev_inv).reshape(3, 3, 2), d, h)
    tp = lin(np.asarray(tprev).reshape(3, 2), d, h)
    with np.errstate(invalid="ignore", divide="ignore"):
        vg = Rc @ vc + tc[:, None]                                       # :212
        vcp = Rpi @ (vg - tp[:, None])                                   # :213
        if dec is None:
            assert d == 0.0
            ok = ~np.isnan(nc[0])                                        # :202-204
            ux = np.rint(vcp[0] * fx / vcp[2] + cx)                      # :216-217 __float2int_rn
            uy = np.rint(vcp[1] * fy / vcp[2] + cy)
            ok &= np.isfinite(ux) & np.isfinite(uy)
            ok &= ~((ux < 0) | (uy < 0) | (ux >= cols) | (uy >= rows) | (vcp[2] < 0))   # :218-220
            idx = (np.where(ok, uy, 0).astype(np.int64) * cols + np.where(ok, ux, 0).astype(np.int64))
            ok &= ~np.isnan(npv[0][idx])                                 # :222-224
            dec = dict(idx=idx)
        idx = dec["idx"]
        n_prev, v_prev = npv[:, idx], vp[:, idx]
        if "ok" not in dec:
            dist = np.sqrt(((v_prev - vg) ** 2).sum(0))                  # :232-234
            ok &= ~(dist > float(F32(dist_thres)))
            ncg = Rc @ nc                                                # :235
            sine = np.sqrt((np.cross(ncg.T, n_prev.T) ** 2).sum(-1))     # :236-238
            ok &= ~(sine >= float(F32(angle_thres)))
            ok &= np.isfinite(dist) & np.isfinite(sine)
            dec["ok"] = ok
        ok = dec["ok"]
        s, nn, dd = vg[:, ok], n_prev[:, ok], v_prev[:, ok]              # :239-242: n = n_prev_g, d = p_prev_g, s = p_curr_g
        row = np.empty((7, s.shape[1]))
        row[0:3] = np.cross(s.T, nn.T).T                                 # :256
        row[3:6] = nn                                                    # :257
        row[6] = (nn * (dd - s)).sum(0)                                  # :258
    sums = np.array([np.dot(row[i], row[j]) for i in range(6) for j in range(i, 7)])   # :266-279
    return sums, int(ok.sum()), dec


# ------------------------------------------------------------------------------------------------
# Dual-complex local-TSDF residual — ComputeLocalTsdfHessianKernel, XKinectFusion/src/TsdfFusion.cu:204-283
def tsdf_residual_loss(p, h, Rv2c, tv2c, gt_planes, depth_m, intr, voxel_size, trunc, z0=0, dec=None):
    """sum over the voxels of gt_planes ([nz, Y, X]: planes z0 .. z0 + nz of the map, as the C ABI takes a slab) of error(p)^2,
    error = (|Dp (xl, yl, 1)| - |v_c| - gt * trunc) / trunc, for the pose x(p) = re.re + p * re.im / h (Rv2c [3, 3, 4],
    tv2c [3, 4] dual-complex groups).  Returns (loss, count, dec)."""
    Rq, tq = np.asarray(Rv2c, np.float64).reshape(3, 3, 4), np.asarray(tv2c, np.float64).reshape(3, 4)
    R = Rq[..., 0] + (p / h) * Rq[..., 1]
    t = tq[..., 0] + (p / h) * tq[..., 1]
    fx, fy, cx, cy = (float(F32(v)) for v in intr)
    vs, tr = float(F32(voxel_size)), float(F32(trunc))
    rows, cols = depth_m.shape
    if dec is None:
        assert p == 0.0
        g = np.asarray(gt_planes, np.float64)
        band = (g != 0) & ~(np.abs(g) > 0.95)                            # :221-223
        zz, yy, xx = np.nonzero(band)
        dec = dict(xyz=np.stack([xx, yy, zz + z0], -1), gt=g[band])
    xyz, g = dec["xyz"], dec["gt"]
    vg = (xyz.astype(np.float64) + 0.5) * vs                             # :224-227
    vc = vg @ R.T + t                                                    # :228
    inv_z = 1.0 / vc[:, 2]                                               # :229
    ix = vc[:, 0] * inv_z * fx + cx                                      # :232-233
    iy = vc[:, 1] * inv_z * fy + cy
    if "keep" not in dec:
        keep = ~(inv_z < 0)                                              # :230-231
        coox, cooy = np.floor(ix - 0.5).astype(np.int64), np.floor(iy - 0.5).astype(np.int64)    # :234-235
        keep &= (coox > 1) & (cooy > 1) & (coox < cols - 1) & (cooy < rows - 1)                  # :236-237
        dec.update(coox=coox, cooy=cooy, nx=np.rint(ix).astype(np.int64), ny=np.rint(iy).astype(np.int64))
        dd = [_gather2d(depth_m, cooy + j, coox + i) for j in (0, 1) for i in (0, 1)]
        dec["bil"] = (dd[0] != 0) & (dd[1] != 0) & (dd[2] != 0) & (dd[3] != 0)                  # :248-251 (threshold unused)
        dec["keep"] = keep
    coox, cooy = dec["coox"], dec["cooy"]
    d00, d10 = _gather2d(depth_m, cooy, coox).astype(np.float64), _gather2d(depth_m, cooy, coox + 1).astype(np.float64)
    d01, d11 = _gather2d(depth_m, cooy + 1, coox).astype(np.float64), _gather2d(depth_m, cooy + 1, coox + 1).astype(np.float64)
    a, b = ix - (coox + 0.5), iy - (cooy + 0.5)                          # :253-254
    inter = d00 * (1 - a) * (1 - b) + d10 * a * (1 - b) + d01 * (1 - a) * b + d11 * a * b
    Dp = np.where(dec["bil"], inter, _gather2d(depth_m, dec["ny"], dec["nx"]).astype(np.float64))
    xl, yl = (ix - cx) / fx, (iy - cy) / fy                              # :262-263
    dist = np.sqrt((Dp * xl) ** 2 + (Dp * yl) ** 2 + Dp ** 2) - np.sqrt((vc ** 2).sum(-1))    # :264-267
    err = (dist - g * tr) / tr                                           # :268-269
    if "use" not in dec:
        dec["use"] = dec["keep"] & ~((Dp > 5) | (Dp < 0.2)) & ~(np.abs(err) > 1)                # :260, :271
    use = dec["use"]
    return float((err[use] ** 2).sum()), int(use.sum()), dec             # :274-280 + the four reductions :317-324


def central(f, step):
    """(value at 0, first, second central difference) of a scalar- or array-valued f(d, dec) -> (value, dec)."""
    f0, dec = f(0.0, None)
    fp, _ = f(+step, dec)
    fm, _ = f(-step, dec)
    return f0, (fp - fm) / (2 * step), (fp - 2 * f0 + fm) / (step * step), dec
