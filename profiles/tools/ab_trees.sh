#!/bin/bash
# GPU box: two built trees alternating on one box (A = scratch/prev: `mkdir -p scratch/prev && git archive <commit> | tar -x -C scratch/prev`, then make in its
# csrc and host directories; B = the working tree).  bench.py --workload track: the headline (nearest pixel) and the bilinear run beside it.
#   profiles/tools/ab_trees.sh [labelA] [labelB]
cd "$(dirname "$0")/../.."
ROOT=$PWD
LA=${1:-previous}; LB=${2:-current}
run() { (cd $1 && timeout -k 10 200 python3 bench.py --workload track --steps $2 --warmup 5 --no-s2 --no-cpu-baseline --no-legs 2>/dev/null) | python3 -c "
import json,sys
p=json.loads(sys.stdin.read().strip().splitlines()[-1])
b=p.get('bilinear') or {}
print('$3 steps $2: frames/s', p['repetitions_fps'], ' integrate kernel ms', p['roofline']['kernel_ms'], ' bilinear', {k: b.get(k) for k in ('frames_per_s', 'integrate_kernel_ms')})"; }
for rep in 1 2 3; do
  run $ROOT/scratch/prev 20 "$LA"
  run $ROOT 20 "$LB"
done
