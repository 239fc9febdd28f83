#!/bin/bash
# Does the ICP reduction kernel respond to occupancy?  Rebuilds xs_icp.hip on the GPU box with the kernel forced to one wave
# per SIMD (default: two, set by its 210 registers and 64 KB of LDS) and times the launches per level both ways.
cd "$(dirname "$0")/../.."
F="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
echo "== default build (2 waves per SIMD)"; python profiles/tools/probe_icp.py 2>/dev/null | grep "^{"
touch x-slam_amd/csrc/xs_icp.hip; make -C x-slam_amd/csrc HIPFLAGS="$F -DXS_ICP_WAVES_PER_EU=1" > /dev/null 2>&1
echo "== 1 wave per SIMD"; python profiles/tools/probe_icp.py 2>/dev/null | grep "^{"
touch x-slam_amd/csrc/xs_icp.hip; make -C x-slam_amd/csrc > /dev/null 2>&1
