#!/bin/bash
# GPU box: the brick kernel compiled for 8 / 7 / 6 workgroups per CU (64 / 72 / 80 VGPRs; scratch per lane 12-48 / 0-12 / 0 bytes):
# S1 + S2 probes and the pipeline (bench.py track: the sign-map-marking instance, frames/s, S1 kernel time)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
F="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
for rep in 1 2; do for w in 8 7 6; do
  touch x-slam_amd/csrc/xs_tsdf.hip; make -C x-slam_amd/csrc HIPFLAGS="$F -DXS_INTEGRATE_WAVES=$w" > /dev/null 2>&1 || exit 1
  echo "== waves $w (round $rep)"
  timeout -k 10 120 python3 profiles/tools/probe_integrate.py 2>/dev/null | tail -1 | cut -c1-80 || exit 1
  timeout -k 10 120 python3 profiles/tools/probe_s2_r4.py 20 2>/dev/null | tail -1 || exit 1
  timeout -k 10 240 python3 bench.py --workload track --no-cpu-baseline --no-s2 --no-legs --steps 100 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  pipeline fps', d['value'], d['repetitions_fps'], 'S1 kernel ms', d['roofline']['kernel_ms'], 'stage', d['stages_ms']['integrate'], 'bilinear', d['bilinear']['frames_per_s'], d['bilinear']['integrate_kernel_ms'])" || exit 1
done; done
touch x-slam_amd/csrc/xs_tsdf.hip; make -C x-slam_amd/csrc > /dev/null 2>&1
