#!/bin/bash
# (HISTORICAL: the compile switch -DXS_MAILBOX_NO_REREAD no longer exists — the layout that followed, a sequence word per 32-byte sector, made the
# second read unnecessary; kept as the record of how profiles/r06_ab_mailbox_reread.txt was measured.)
# Round 6, measurement only: what the mailbox poller's second read (the payload is read again after both sequence words were seen) costs a frame.
# Two builds of libxslam_hip.so (with / without -DXS_MAILBOX_NO_REREAD) swapped in place, alternating; the product build is restored at the end.
set -e
make -C x-slam_amd/csrc clean > /dev/null; make -C x-slam_amd/csrc EXTRAFLAGS=-DXS_MAILBOX_NO_REREAD > /dev/null 2>&1; cp x-slam_amd/libxslam_hip.so /tmp/hip_noreread.so
make -C x-slam_amd/csrc clean > /dev/null; make -C x-slam_amd/csrc > /dev/null 2>&1; cp x-slam_amd/libxslam_hip.so /tmp/hip_product.so
for round in 1 2 3 4; do
  for v in product noreread; do
    cp /tmp/hip_$v.so x-slam_amd/libxslam_hip.so
    python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-s2 --no-csfd --no-legs --workload track 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
i = d['icp_us_per_iteration']
print('$v'.ljust(9), 'round $round:', 'frames/s', d['value'], ' ICP iteration us', i['level0'], i['level1'], i['level2'])
"
  done
done
cp /tmp/hip_product.so x-slam_amd/libxslam_hip.so   # (the switch existed for this measurement only: the layout that followed made the second read unnecessary and the switch is gone)
