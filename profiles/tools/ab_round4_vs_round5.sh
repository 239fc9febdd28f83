#!/bin/bash
# GPU box: the round-4 tree (mkdir -p scratch/r04 && git archive 048ef56 | tar -x -C scratch/r04, then make in its csrc and host directories) against the current tree, alternating on one box:
# bench.py --workload track (headline workload only), the driver's region shape (--steps 20 --warmup 5) and a longer one (--steps 60).
cd "$(dirname "$0")/../.."
ROOT=$PWD
run() { (cd $1 && timeout -k 10 200 python3 bench.py --workload track --steps $2 --warmup 5 --no-s2 --no-cpu-baseline --no-legs 2>/dev/null) | python3 -c "
import json,sys
p=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$3 steps $2: frames/s', p['repetitions_fps'], ' integrate kernel ms', p['roofline']['kernel_ms'], ' stages', {k: v for k, v in (p.get('stages_ms') or {}).items() if k in ('icp', 'integrate', 'raycast')})"; }
for rep in 1 2 3; do
  run $ROOT/scratch/r04 20 "round 4 (048ef56)"
  run $ROOT 20 "round 5           "
done
for rep in 1 2; do
  run $ROOT/scratch/r04 60 "round 4 (048ef56)"
  run $ROOT 60 "round 5           "
done
