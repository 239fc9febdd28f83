"""Round 6 soak of the two hand-off changes (GPU box, repository root): scene S3 at 256^3, a 300-frame lap of the camera path.
  1. two trackers side by side, icp_publish_pairs true / false: every pose of 600 frames identical bits;
  2. the default tracker (pairs + direct-store posts) for LAPS laps of 300 frames on fresh volumes: every frame tracked (no launch gave up
     waiting for a pose, no sums that never arrived), the last pose of every lap identical bits to the first lap's.
    python profiles/tools/soak_handoffs.py [LAPS=150]"""
import importlib, json, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
pl = importlib.import_module('x-slam_amd.pipeline'); synth = importlib.import_module('x-slam_amd.synth')
laps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
n = 256
frames = [torch.from_numpy(synth.s3_frame(k).view(np.int16)).cuda() for k in range(300)]
prm = synth.s1_params(n)
a, b = pl.KinectFusion(dict(prm, icp_publish_pairs=True)), pl.KinectFusion(dict(prm, icp_publish_pairs=False))
for k in range(600):
    f = frames[k % 300]
    assert a.process_frame(f) == 1 and b.process_frame(f) == 1, k
    assert np.array_equal(a.world2camera().view(np.int32), b.world2camera().view(np.int32)), k
a.close(); b.close()
print(json.dumps({"side_by_side_frames": 600, "poses": "identical bits"}), flush=True)
t0 = time.perf_counter()
first = None
total = 0
for lap in range(laps):
    kf = pl.KinectFusion(prm)
    for k in range(300):
        assert kf.process_frame(frames[k]) == 1, (lap, k)
    total += 300
    last = kf.world2camera().view(np.int32).copy()
    kf.close()
    if first is None:
        first = last
    assert np.array_equal(first, last), lap
    if lap % 25 == 24:
        print(json.dumps({"laps": lap + 1, "frames": total, "s": round(time.perf_counter() - t0, 1)}), flush=True)
print(json.dumps({"laps": laps, "frames": total, "every_frame_tracked": True, "last_pose_of_every_lap": "identical bits", "frames_per_s_incl_setup": round(total / (time.perf_counter() - t0), 1)}))
