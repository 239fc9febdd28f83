#!/bin/bash
# GPU box, repository root: the integrate kernel with 64x4, 32x8 and 16x16 column bricks (S1 probe, S2 probe), product build restored afterwards
cd "$(dirname "$0")/../.."
F="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
for bx in 64 32 16; do
  touch x-slam_amd/csrc/xs_tsdf.hip; make -C x-slam_amd/csrc HIPFLAGS="$F -DXS_BRICK_X=$bx" > /dev/null 2>&1
  echo "brick ${bx} x $((256 / bx))"; python profiles/tools/probe_integrate.py 2>&1 | grep "bricks listed"
  python profiles/tools/probe_s2.py 8 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-300
done
touch x-slam_amd/csrc/xs_tsdf.hip; make -C x-slam_amd/csrc > /dev/null 2>&1
