#!/bin/bash
# GPU box, repository root: brick shapes 32x8 ... 256x1 (x contiguous bytes per workgroup plane: 128 B ... 1 KB) for the round-2 walk
# and the round-3 grouped kernels, S2 probe + S1 probe; product build restored afterwards
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
F="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
OUT=gpurun_out/r03_ab_brick_shape.txt; : > $OUT
for bx in ${SHAPES:-32 64 128 256}; do
  touch x-slam_amd/csrc/xs_tsdf.hip; make -C x-slam_amd/csrc HIPFLAGS="$F -DXS_BRICK_X=$bx" > /dev/null 2>&1
  for v in ${VARIANTS:-walk default g4w7}; do
    if [ $v = default ]; then unset XS_INTEGRATE_KERNEL; else export XS_INTEGRATE_KERNEL=$v; fi
    echo "== brick ${bx} x $((256 / bx))  kernel $v" >> $OUT
    timeout -k 10 120 python profiles/tools/probe_s2.py 20 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-300 >> $OUT || exit 1
    timeout -k 10 120 python profiles/tools/probe_s1.py 20 only 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-300 >> $OUT || exit 1
  done
done
touch x-slam_amd/csrc/xs_tsdf.hip; make -C x-slam_amd/csrc > /dev/null 2>&1
cat $OUT
