#!/bin/bash
# GPU box: the per-voxel walk of the brick kernel G planes at a time (XS_WALK_GROUP; integrate_span_grouped) against the plain walk,
# at the workgroups-per-CU each form needs: S1 + S2 + 1024^3 probes; with "pipeline" as first argument also bench.py's track.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
F="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
VARIANTS=${VARIANTS:-"-DXS_WALK_GROUP=0;-DXS_WALK_GROUP=4;-DXS_WALK_GROUP=4 -DXS_INTEGRATE_WAVES=5;-DXS_WALK_GROUP=8 -DXS_INTEGRATE_WAVES=4;-DXS_WALK_GROUP=0"}
IFS=';' read -ra V <<< "$VARIANTS"
for v in "${V[@]}"; do
  touch x-slam_amd/csrc/xs_tsdf.hip; make -C x-slam_amd/csrc HIPFLAGS="$F $v" > /dev/null 2>&1 || exit 1
  echo "== $v"
  timeout -k 10 120 python3 profiles/tools/probe_integrate.py 2>/dev/null | tail -1 | cut -c1-100 || exit 1
  timeout -k 10 120 python3 profiles/tools/probe_s2_r4.py 20 2>/dev/null | tail -1 || exit 1
  XS_PROBE_N=1024 timeout -k 10 120 python3 profiles/tools/probe_integrate.py 2>/dev/null | tail -1 | cut -c1-100 || exit 1
  if [ "$1" = pipeline ]; then
  timeout -k 10 240 python3 bench.py --workload track --no-cpu-baseline --no-s2 --no-legs --steps 100 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  pipeline fps', d['value'], d['repetitions_fps'], 'S1 kernel ms', d['roofline']['kernel_ms'], 'stage', d['stages_ms']['integrate'], 'bilinear', d['bilinear']['frames_per_s'], d['bilinear']['integrate_kernel_ms'])" || exit 1
  fi
done
touch x-slam_amd/csrc/xs_tsdf.hip; make -C x-slam_amd/csrc > /dev/null 2>&1
