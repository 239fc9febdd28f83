#!/bin/bash
cd "$(dirname "$0")/../.."
F="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
touch x-slam_amd/csrc/xs_icp.hip; make -C x-slam_amd/csrc HIPFLAGS="$F -DXS_ICP_NO_PREFETCH" > /dev/null 2>&1
echo "no prefetch"; bash profiles/tools/bench3.sh | cut -c1-120
touch x-slam_amd/csrc/xs_icp.hip; make -C x-slam_amd/csrc > /dev/null 2>&1
echo "vertex prefetch"; bash profiles/tools/bench3.sh | cut -c1-120
