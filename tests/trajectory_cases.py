"""GPU pipeline and the CPU oracle pipeline side by side over a whole stretch of a trajectory at BASELINE's sizes
(north_star: "camera trajectory and per-voxel TSDF within a stated float tolerance" on the configs).  Test infrastructure."""
import importlib
import time

import numpy as np

from helpers import synth


def side_by_side(torch, pl, oracle, scene, n, frames, seed=(0, 3), threshold=0.0, voxel_samples=400000, sensitivity=False):
    """Per-frame differences between the two pipelines fed the same frames, and the fused volume at the end on seeded voxels."""
    from oracle.oracle import OracleKinFu, params_from_dict
    prm = synth.s1_params(n, seed=seed, threshold=threshold)
    kf = pl.KinectFusion(prm)
    ok = OracleKinFu(oracle, params_from_dict(prm))
    # sensitivity of the scene itself: the same GPU pipeline fed the same frames, except that one pixel of frame 3 reads 1 mm more
    twin = pl.KinectFusion(prm) if sensitivity else None
    if twin is not None:
        rec_twin = []
    render = synth.s3_frame if scene == "s3" else synth.s1_frame
    rec = dict(dpose=[], dderiv_rel=[], dderiv_abs=[], deriv_scale=[], dU=[], U=[], dhits=[], hits=[], dinliers=[])
    rec_twin, rec_twin_deriv = [], []
    secs = dict(render=0.0, gpu=0.0, oracle=0.0, volumes=0.0)
    for k in range(frames):
        t0 = time.perf_counter()
        d = render(k)
        t1 = time.perf_counter()
        a = kf.process_frame(torch.from_numpy(d.view(np.int16)).cuda())
        kf.synchronize()
        t2 = time.perf_counter()
        b = ok.process_frame(d)
        t3 = time.perf_counter()
        if twin is not None:
            d2 = d.copy()
            if k == 3:
                d2[240, 320] += 1
            assert twin.process_frame(torch.from_numpy(d2.view(np.int16)).cuda()) == 1
            rec_twin.append(float(np.abs(kf.world2camera()[..., 0].astype(np.float64) - twin.world2camera()[..., 0]).max()))
            rec_twin_deriv.append(float(np.abs(kf.world2camera()[..., 1].astype(np.float64) - twin.world2camera()[..., 1]).max() / prm["csfd_seed_h"]))
        secs["render"] += t1 - t0; secs["gpu"] += t2 - t1; secs["oracle"] += t3 - t2
        assert a == 1 and b == 1, (scene, n, k, a, b)
        g, w = kf.world2camera().astype(np.float64), ok.world2camera().astype(np.float64)
        scale = max(np.abs(w[..., 1]).max(), 1e-30)
        rec["dpose"].append(float(np.abs(g[..., 0] - w[..., 0]).max()))
        rec["dderiv_rel"].append(float(np.abs(g[..., 1] - w[..., 1]).max() / scale))
        rec["dderiv_abs"].append(float(np.abs(g[..., 1] - w[..., 1]).max() / prm["csfd_seed_h"]))   # in units of d pose / d seed
        rec["deriv_scale"].append(float(scale / prm["csfd_seed_h"]))
        rec["dU"].append(int(abs(kf.last_U() - ok.last_U()))); rec["U"].append(int(ok.last_U()))
        rec["dhits"].append(int(abs(kf.last_hits() - ok.last_hits()))); rec["hits"].append(int(ok.last_hits()))
        if k > 0:
            il, wl = kf.icp_log(), ok.icp_log()
            rec["dinliers"].append(int(np.abs(il[:, 54] - wl[:, 54]).max()) if il.shape == wl.shape else -1)
    if twin is not None:
        twin.close()
        rec["sensitivity_dpose"] = rec_twin
        rec["sensitivity_dderiv_abs"] = rec_twin_deriv
    # the fused volumes on seeded voxels (the arrays are 0.5-1.5 GB each at 512^3: one at a time)
    t0 = time.perf_counter()
    rng = np.random.default_rng(0xC5FD + n)
    vox = np.sort(rng.choice(n ** 3, voxel_samples, replace=False))
    gv, gw, gg = (a[vox] for a in kf.volume())
    kf.close()
    ov, ow, og = (a[vox] for a in ok.volume())
    ok.close()
    secs["volumes"] = time.perf_counter() - t0
    rec["seconds"] = {k_: round(v, 2) for k_, v in secs.items()}
    same = gw == ow
    touched = ow > 0
    gs = max(np.abs(og).max(), 1e-30)
    rec["voxels"] = dict(sampled=int(vox.size), touched=int(touched.sum()), weight_mismatch=float((~same).mean()),
                         value_bad=float((np.abs(gv[same] - ov[same]) > 1e-4).mean()),
                         value_max=float(np.abs(gv[same] - ov[same]).max()),
                         grad_bad=float((np.abs(gg[same] - og[same]) > 1e-3 * gs).mean()),
                         grad_scale=float(gs / prm["csfd_seed_h"]))
    return rec
