#!/bin/bash
# GPU box, repository root: the ICP kernels alone with the last workgroup's final additions kept in step (default) against the compiler's
# own order (-DXS_ICP_TAIL_BASELINE: spills), rebuilt and timed alternately on this box
cd "$(dirname "$0")/../.."
F="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
for v in new base new base new base; do
  touch x-slam_amd/csrc/xs_icp.hip
  if [ $v = base ]; then make -C x-slam_amd/csrc HIPFLAGS="$F -DXS_ICP_TAIL_BASELINE" > /dev/null 2>&1; else make -C x-slam_amd/csrc > /dev/null 2>&1; fi
  echo "== $v"; profiles/tools/time_icp_kernels.sh | grep "16, 2"
done
touch x-slam_amd/csrc/xs_icp.hip; make -C x-slam_amd/csrc > /dev/null 2>&1
