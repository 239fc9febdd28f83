#!/bin/bash
# GPU box: the crossing kernels' minimum waves per SIMD (XS_RAYCAST_CROSS_WAVES: the register budget the compiler gets), raycast alone on the tracked 512^3 volume
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
F="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -fno-slp-vectorize"
for rep in 1 2; do for w in 5 4 6 7 8; do
  touch x-slam_amd/csrc/xs_raycast.hip; make -C x-slam_amd/csrc HIPFLAGS="$F -DXS_RAYCAST_CROSS_WAVES=$w" xs_raycast.o > /dev/null 2>&1 && make -C x-slam_amd/csrc > /dev/null 2>&1 || exit 1
  echo -n "cross waves $w: "
  timeout -k 10 240 python3 bench.py --workload track --no-cpu-baseline --no-s2 --no-legs --steps 60 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['raycast']; print('raycast alone ms', r['ms_per_frame_alone'], 'every step', r['every_step']['ms_per_frame_alone'], 'stage', d['stages_ms']['raycast'], 'fps', d['value'])" || exit 1
done; done
touch x-slam_amd/csrc/xs_raycast.hip; make -C x-slam_amd/csrc > /dev/null 2>&1
