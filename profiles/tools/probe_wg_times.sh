#!/bin/bash
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
F="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
touch x-slam_amd/csrc/xs_tsdf.hip; make -C x-slam_amd/csrc HIPFLAGS="$F -DXS_PROBE_WG_TIMES $EXTRA" > /dev/null 2>&1 || exit 1
timeout -k 10 120 python3 profiles/tools/probe_wg_times.py 2>&1 | grep -v amdgpu.ids | tail -40
touch x-slam_amd/csrc/xs_tsdf.hip; make -C x-slam_amd/csrc > /dev/null 2>&1
