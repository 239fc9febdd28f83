#!/bin/bash
# GPU box, repository root: rebuild xs_icp.hip with phase stamps, print the timelines, rebuild the product kernel.
cd "$(dirname "$0")/../.."
F="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
touch x-slam_amd/csrc/xs_icp.hip; make -C x-slam_amd/csrc HIPFLAGS="$F -DXS_ICP_TRACE" > /dev/null 2>&1
python profiles/tools/trace_icp.py 2>&1 | grep -v amdgpu.ids
touch x-slam_amd/csrc/xs_icp.hip; make -C x-slam_amd/csrc > /dev/null 2>&1
