#!/bin/bash
# GPU box, repository root: stamps build, the pose wait of a launch inside the tracking loop, product build again.
cd "$(dirname "$0")/../.."
F="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
touch x-slam_amd/csrc/xs_icp.hip; make -C x-slam_amd/csrc HIPFLAGS="$F -DXS_ICP_TRACE" > /dev/null 2>&1
timeout -k 10 300 python profiles/tools/trace_icp_loop.py 2>&1 | grep -v amdgpu.ids
touch x-slam_amd/csrc/xs_icp.hip; make -C x-slam_amd/csrc > /dev/null 2>&1
