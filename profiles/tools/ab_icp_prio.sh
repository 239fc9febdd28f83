#!/bin/bash
# GPU box, repository root: the ICP kernels' waves at issue priority 3 (default) against 0 (-DXS_ICP_NO_PRIO), with the next frame announced
cd "$(dirname "$0")/../.."
F="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
for v in prio base prio base prio base; do
  touch x-slam_amd/csrc/xs_icp.hip
  if [ $v = base ]; then make -C x-slam_amd/csrc HIPFLAGS="$F -DXS_ICP_NO_PRIO" > /dev/null 2>&1; else make -C x-slam_amd/csrc > /dev/null 2>&1; fi
  timeout -k 10 300 python3 bench.py --workload track --no-s2 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); i=d['icp_us_per_iteration']; print('$v fps', d['value'], d['repetitions_fps'], 'icp us', i['level0'], i['level1'], i['level2'], 'first', i['first_iteration_of_frame'], 'sustained', d['sustained']['frames_per_s'])" || exit 1
done
touch x-slam_amd/csrc/xs_icp.hip; make -C x-slam_amd/csrc > /dev/null 2>&1
