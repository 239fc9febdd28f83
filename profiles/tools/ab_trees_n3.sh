#!/bin/bash
# GPU box: two built trees (A = scratch/prev, see ab_trees.sh; B = the working tree) alternating on one box, the two N^3-bound workloads:
# the Hessian pass per frame at 512^3 (config 4) and the Gauss-Newton relocalisation at 1024^3 (config 5).
cd "$(dirname "$0")/../.."
ROOT=$PWD
run() { for w in hessian reloc; do (cd $1 && timeout -k 10 300 python3 bench.py --workload $w --no-cpu-baseline 2>/dev/null) | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r={k: {kk: v.get(kk) for kk in ('kernel_ms', 'frac', 'achieved')} for k, v in d.items() if k.startswith('roofline') and isinstance(v, dict)}
print('$2 $w:', d['value'], d['unit'], r)"; done; }
for rep in 1 2 3; do
  run $ROOT/scratch/prev "${1:-previous}"
  run $ROOT "${2:-current }"
done
