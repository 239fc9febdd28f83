#!/bin/bash
# GPU box, repository root: the integrate kernel's launch knobs on the S2 / S1 probes after the store elision (brick planes, grid, brick shape)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
F="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
OUT=gpurun_out/r03_ab_integrate_knobs.txt; : > $OUT
run() { echo "== $1" >> $OUT
  timeout -k 10 120 python profiles/tools/probe_s2.py 20 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-300 >> $OUT || exit 1
  timeout -k 10 120 python profiles/tools/probe_s1.py 20 only 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-300 >> $OUT || exit 1; }
for bx in 32 64; do
  touch x-slam_amd/csrc/xs_tsdf.hip; make -C x-slam_amd/csrc HIPFLAGS="$F -DXS_BRICK_X=$bx" > /dev/null 2>&1
  for bz in 8 16 4; do for g in 8192 2048 4096 16384; do
    [ $bz != 8 ] && [ $g != 8192 ] && continue
    XS_BRICK_Z=$bz XS_BRICK_GRID=$g run "brick ${bx}x$((256 / bx))x$bz grid $g"
  done; done
done
touch x-slam_amd/csrc/xs_tsdf.hip; make -C x-slam_amd/csrc > /dev/null 2>&1
cat $OUT
