cd /root/repo
for i in 1 2; do
for args in "--steps 20 --warmup 5" ""; do
echo "== bench.py --workload track --no-cpu-baseline --no-s2 --no-legs $args"
timeout -k 10 300 python3 bench.py --workload track --no-cpu-baseline --no-s2 --no-legs $args 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  fps', d['value'], d['repetitions_fps'], 'steps', d['steps'], 'warmup', d['warmup'])"
done; done
