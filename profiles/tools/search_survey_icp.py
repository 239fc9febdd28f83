"""CPU only (the oracle): which inputs did the survey's level-0 search_newton call have?  SURVEY.md section 6 recorded 288 814 (256^3) /
289 412 (512^3) inliers "on the same frame" without saying which current-frame maps, pose, thresholds.  Enumerates the plausible ones
(Appendix A step 6: fields filled as ICP.cu:376-391) and prints the inlier count of each; run from the repository root:
    python profiles/tools/search_survey_icp.py [256|512]"""
import importlib, itertools, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from oracle import oracle as orc
synth = importlib.import_module('x-slam_amd.synth')
from helpers import intr_of, s1_transforms, tranc_dist

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
want = {256: 288814, 512: 289412}[n]
orc.build(ref=False)
o = orc.Oracle()
H, W = synth.HEIGHT, synth.WIDTH
res = [n, n, n]
rows = []
for seed in ((0, 3), None):
    prm = synth.s1_params(n, seed=seed)
    T0 = s1_transforms(0, prm, seed=seed)
    d0 = synth.s1_frame(0)
    v, w, g = o.new_volume(res)
    U = o.integrate(o.scale_depth(d0), v, w, g, res, tranc_dist(prm), 100, T0["Rv2c"], T0["tv2c"], intr_of(prm), prm["tsdf_voxel_size"])
    pv, pn, hits = o.raycast(intr_of(prm), T0["Rc2v"], T0["tc2v"], T0["Rv2w"], T0["tv2w"], tranc_dist(prm), res, prm["tsdf_voxel_size"], v, g, H, W)
    print(f"seed {seed}: U {U} hits {hits}", flush=True)
    for frame, src in itertools.product((0, 1, 2), ("bilateral", "raw")):
        d = synth.s1_frame(frame)
        dc = o.bilateral(d) if src == "bilateral" else np.stack([d.astype(np.float32), np.zeros_like(d, np.float32)], -1)
        cv = o.create_vmap(intr_of(prm), dc)
        cn = o.create_nmap(cv)
        for pose_k in sorted({0, frame}):
            Tk = s1_transforms(pose_k, prm, seed=seed)
            for dist, (aname, ang) in itertools.product((0.10, 0.05, 0.2), (("sin15", float(np.sin(np.float32(15.0) / np.float32(180.0) * np.pi))),
                                                                              ("sin15_f64", float(np.float32(np.sin(np.deg2rad(15.0))))),
                                                                              ("15rad", float(np.float32(15.0 * np.pi / 180.0))), ("sin20", float(np.float32(np.sin(np.deg2rad(20.0))))),
                                                                              ("sin30", 0.5), ("deg15", 15.0))):
                _, _, _, inl = o.icp_combined(Tk["Rc2w"], Tk["tc2w"], cv, cn, o.m3_inverse(T0["Rc2w"]), T0["tc2w"], intr_of(prm), pv, pn, dist, ang)
                rows.append((abs(inl - want), inl, seed, frame, src, pose_k, dist, aname))
rows.sort(key=lambda r: r[0])
print(f"wanted {want}")
for r in rows[:25]:
    print("  |diff| %6d  inliers %7d  seed %-7s current = frame %d (%s depth) at pose %d, distThres %.2f, angleThres %s" % r)
