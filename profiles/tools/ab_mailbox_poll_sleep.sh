#!/bin/bash
# Round 6: s_sleep between two polls of the pose mailbox (x 64 clocks): 8 (product), 2, 0.  Builds of libxslam_hip.so swapped in place, alternating.
set -e
for v in 8 2 0; do
  make -C x-slam_amd/csrc clean > /dev/null; make -C x-slam_amd/csrc EXTRAFLAGS=-DXS_MAILBOX_POLL_SLEEP=$v > /dev/null 2>&1; cp x-slam_amd/libxslam_hip.so /tmp/hip_sleep$v.so
done
for round in 1 2 3; do
  for v in 8 2 0; do
    cp /tmp/hip_sleep$v.so x-slam_amd/libxslam_hip.so
    python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-s2 --no-csfd --no-legs --workload track 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
i = d['icp_us_per_iteration']
print('s_sleep $v', 'round $round:', 'frames/s', d['value'], ' ICP iteration us', i['level0'], i['level1'], i['level2'])
"
  done
done
make -C x-slam_amd/csrc clean > /dev/null; make -C x-slam_amd/csrc > /dev/null 2>&1
