#!/bin/bash
# (HISTORICAL: the switch -DXS_ICP_GATHER_COHERENT_LOADS was removed after this measurement; see profiles/r06_ab_icp_gather_loads.txt)
# Round 6: the last workgroup of an ICP launch reads the other workgroups' records after an agent-scope acquire fence (product) or, without the fence,
# with agent-scope loads (-DXS_ICP_GATHER_COHERENT_LOADS).  Two builds of libxslam_hip.so swapped in place, alternating; the product build is restored.
set -e
make -C x-slam_amd/csrc clean > /dev/null; make -C x-slam_amd/csrc EXTRAFLAGS=-DXS_ICP_GATHER_COHERENT_LOADS > /dev/null 2>&1; cp x-slam_amd/libxslam_hip.so /tmp/hip_coherent.so
make -C x-slam_amd/csrc clean > /dev/null; make -C x-slam_amd/csrc > /dev/null 2>&1; cp x-slam_amd/libxslam_hip.so /tmp/hip_product.so
for round in 1 2 3 4; do
  for v in product coherent; do
    cp /tmp/hip_$v.so x-slam_amd/libxslam_hip.so
    python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-s2 --no-csfd --no-legs --workload track 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
i = d['icp_us_per_iteration']
print('$v'.ljust(9), 'round $round:', 'frames/s', d['value'], ' ICP iteration us', i['level0'], i['level1'], i['level2'])
"
  done
done
cp /tmp/hip_coherent.so x-slam_amd/libxslam_hip.so
timeout -k 10 300 python -m pytest tests/test_publish_stress_gpu.py tests/test_kernels_gpu.py -x -q -k "icp" 2>&1 | tail -2
cp /tmp/hip_product.so x-slam_amd/libxslam_hip.so
