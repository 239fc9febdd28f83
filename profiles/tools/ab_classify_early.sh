cd /root/repo
for rep in 1 2; do for early in 0 1 2; do
echo "== integrate_classify_early=$early beside=true (round $rep)"
XS_KF_DEBUG_COVERS=1 timeout -k 10 240 python3 bench.py --workload track --no-cpu-baseline --no-s2 --no-legs --steps 100 --param integrate_classify_early=$early 2> gpurun_out/early_err.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  pipeline fps', d['value'], d['repetitions_fps'], 'S1 kernel ms', d['roofline']['kernel_ms'], 'icp', d['stages_ms']['icp'], 'integrate', d['stages_ms']['integrate'], 'bilinear', d['bilinear']['frames_per_s'])" || exit 1
grep "list covers" gpurun_out/early_err.txt | sort | uniq -c | head -5
done; done
echo "== beside=false early=0"
timeout -k 10 240 python3 bench.py --workload track --no-cpu-baseline --no-s2 --no-legs --steps 100 --param integrate_classify_beside_icp=false 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  pipeline fps', d['value'], d['repetitions_fps'], 'S1 kernel ms', d['roofline']['kernel_ms'], 'icp', d['stages_ms']['icp'], 'integrate', d['stages_ms']['integrate'], 'bilinear', d['bilinear']['frames_per_s'])"
