"""GPU pipeline and the CPU oracle pipeline side by side over a whole stretch of a trajectory at BASELINE's sizes
(north_star: "camera trajectory and per-voxel TSDF within a stated float tolerance" on the configs).  Test infrastructure."""
import importlib

import numpy as np

from helpers import synth


def side_by_side(torch, pl, oracle, scene, n, frames, seed=(0, 3), threshold=0.0, voxel_samples=400000):
    """Per-frame differences between the two pipelines fed the same frames, and the fused volume at the end on seeded voxels."""
    from oracle.oracle import OracleKinFu, params_from_dict
    prm = synth.s1_params(n, seed=seed, threshold=threshold)
    kf = pl.KinectFusion(prm)
    ok = OracleKinFu(oracle, params_from_dict(prm))
    render = synth.s3_frame if scene == "s3" else synth.s1_frame
    rec = dict(dpose=[], dderiv_rel=[], deriv_scale=[], dU=[], U=[], dhits=[], hits=[], dinliers=[])
    for k in range(frames):
        d = render(k)
        a = kf.process_frame(torch.from_numpy(d.view(np.int16)).cuda())
        b = ok.process_frame(d)
        assert a == 1 and b == 1, (scene, n, k, a, b)
        g, w = kf.world2camera().astype(np.float64), ok.world2camera().astype(np.float64)
        scale = max(np.abs(w[..., 1]).max(), 1e-30)
        rec["dpose"].append(float(np.abs(g[..., 0] - w[..., 0]).max()))
        rec["dderiv_rel"].append(float(np.abs(g[..., 1] - w[..., 1]).max() / scale))
        rec["deriv_scale"].append(float(scale / prm["csfd_seed_h"]))
        rec["dU"].append(int(abs(kf.last_U() - ok.last_U()))); rec["U"].append(int(ok.last_U()))
        rec["dhits"].append(int(abs(kf.last_hits() - ok.last_hits()))); rec["hits"].append(int(ok.last_hits()))
        if k > 0:
            il, wl = kf.icp_log(), ok.icp_log()
            rec["dinliers"].append(int(np.abs(il[:, 54] - wl[:, 54]).max()) if il.shape == wl.shape else -1)
    # the fused volumes on seeded voxels (the arrays are 0.5-1.5 GB each at 512^3: one at a time)
    rng = np.random.default_rng(0xC5FD + n)
    vox = np.sort(rng.choice(n ** 3, voxel_samples, replace=False))
    gv, gw, gg = (a[vox] for a in kf.volume())
    kf.close()
    ov, ow, og = (a[vox] for a in ok.volume())
    ok.close()
    same = gw == ow
    touched = ow > 0
    gs = max(np.abs(og).max(), 1e-30)
    rec["voxels"] = dict(sampled=int(vox.size), touched=int(touched.sum()), weight_mismatch=float((~same).mean()),
                         value_bad=float((np.abs(gv[same] - ov[same]) > 1e-4).mean()),
                         value_max=float(np.abs(gv[same] - ov[same]).max()),
                         grad_bad=float((np.abs(gg[same] - og[same]) > 1e-3 * gs).mean()),
                         grad_scale=float(gs / prm["csfd_seed_h"]))
    return rec
