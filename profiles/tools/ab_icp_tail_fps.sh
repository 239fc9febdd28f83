#!/bin/bash
# GPU box, repository root: headline frames/s and the level-0 iteration period with the ICP tail fix (default) against -DXS_ICP_TAIL_BASELINE
cd "$(dirname "$0")/../.."
F="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
for v in new base new base new base; do
  touch x-slam_amd/csrc/xs_icp.hip
  if [ $v = base ]; then make -C x-slam_amd/csrc HIPFLAGS="$F -DXS_ICP_TAIL_BASELINE" > /dev/null 2>&1; else make -C x-slam_amd/csrc > /dev/null 2>&1; fi
  timeout -k 10 300 python3 bench.py --workload track --no-s2 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v fps', d['value'], d['repetitions_fps'], 'level0 us', d['icp_us_per_iteration']['level0'], 'sustained', d['sustained']['frames_per_s'])" || exit 1
done
touch x-slam_amd/csrc/xs_icp.hip; make -C x-slam_amd/csrc > /dev/null 2>&1
