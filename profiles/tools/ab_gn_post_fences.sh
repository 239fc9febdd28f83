#!/bin/bash
# Round 6: the six-pose post of the Gauss-Newton loop with three store fences (all payloads, sequence words of boxes 0..4, those of box 5) against
# twelve (six xs_icp_post_pose calls).  GPU box, repository root; builds the library with -DXS_EXPERIMENTS and restores the product build.
set -e
make -C x-slam_amd/csrc clean > /dev/null; make -C x-slam_amd/csrc EXTRAFLAGS=-DXS_EXPERIMENTS > /dev/null 2>&1
for round in 1 2 3; do
  for twelve in 1 0; do
    if [ $twelve = 1 ]; then export XS_GN_POST_TWELVE_FENCES=1; else unset XS_GN_POST_TWELVE_FENCES; fi
    python bench.py --workload reloc --steps 30 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d.get('workloads', {}).get('reloc', d)
h = r['host_side']
print('twelve fences' if $twelve else 'three fences ', 'round $round:', 'host_us_per_pass', h['host_us_per_pass'], 'posted period', h['posted_pass_period_us'], 'after-solve period - kernel', h['launched_after_the_solve']['period_minus_kernel_us'], 'fps', r['value'])
"
  done
done
unset XS_GN_POST_TWELVE_FENCES
make -C x-slam_amd/csrc clean > /dev/null; make -C x-slam_amd/csrc > /dev/null 2>&1
