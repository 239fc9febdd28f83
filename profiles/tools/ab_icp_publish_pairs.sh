#!/bin/bash
# Round 6: the ICP sums published as 55 self-validating 16-byte pairs {sequence number, sum} (icp_publish_pairs: true) against 55 doubles + a
# completion word behind a wait, a barrier and a release store (false).  GPU box, repository root, product library; alternating, 200 frames each.
for round in 1 2 3 4; do
  for pairs in false true; do
    python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-s2 --no-csfd --no-legs --workload track --param icp_publish_pairs=$pairs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
i = d['icp_us_per_iteration']
print('pairs' if '$pairs' == 'true' else 'word ', 'round $round:', 'frames/s', d['value'], ' ICP iteration us', i['level0'], i['level1'], i['level2'], ' first', i['first_iteration_of_frame'])
"
  done
done
