cd /root/repo
for rep in 1 2; do for sl in 2 3 4 6; do
echo "== predicted, integrate_classify_slack=$sl (round $rep)"
XS_KF_DEBUG_COVERS=1 timeout -k 10 240 python3 bench.py --workload track --no-cpu-baseline --no-s2 --no-legs --steps 100 --param integrate_classify_predicted=true --param integrate_classify_slack=$sl 2> gpurun_out/pred_err.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  pipeline fps', d['value'], d['repetitions_fps'], 'S1 kernel ms', d['roofline']['kernel_ms'], 'icp', d['stages_ms']['icp'], 'integrate', d['stages_ms']['integrate'], 'bilinear', d['bilinear']['frames_per_s'])" || exit 1
grep "list covers" gpurun_out/pred_err.txt | sort | uniq -c | head -5
done; done
