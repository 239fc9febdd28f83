#!/bin/bash
# Round 6: the orchestrator's scheduling knobs once more at the final tree (posts and sums are cheaper now: has any balance shifted?).
# GPU box, repository root, product library; alternating, 200 frames each.
for round in 1 2 3; do
  for cfg in "" "--icp-lookahead 2" "--integrate-post" "--param integrate_classify_early=1" "--param raycast_pyramid=false"; do
    python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-s2 --no-csfd --no-legs --workload track $cfg 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
i = d['icp_us_per_iteration']
print(('$cfg' or 'default').ljust(40), 'round $round:', 'frames/s', d['value'], ' ICP us', i['level0'], i['level1'], i['level2'], 'first', i['first_iteration_of_frame'])
"
  done
done
