cd /root/repo
timeout -k 10 600 python -m pytest tests/test_pipeline_gpu.py tests/test_trajectory_gpu.py tests/test_publish_stress_gpu.py -x -q -m gpu 2>&1 | tail -4 || exit 1
for rep in 1 2 3; do for v in true false; do
echo "== integrate_classify_predicted=$v (round $rep)"
XS_KF_DEBUG_COVERS=1 timeout -k 10 240 python3 bench.py --workload track --no-cpu-baseline --no-s2 --no-legs --steps 100 --param integrate_classify_predicted=$v 2> gpurun_out/pred_err.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  pipeline fps', d['value'], d['repetitions_fps'], 'S1 kernel ms', d['roofline']['kernel_ms'], 'icp', d['stages_ms']['icp'], 'integrate', d['stages_ms']['integrate'], 'raycast', d['stages_ms']['raycast'], 'bilinear', d['bilinear']['frames_per_s'])" || exit 1
grep "list covers" gpurun_out/pred_err.txt | sort | uniq -c | head -5
done; done
