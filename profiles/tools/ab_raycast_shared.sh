#!/bin/bash
# GPU box: what fetching shared corners once could buy the raycast crossing at most (profiles/tools/probe_raycast_shared.py).
# The probe is a patch (profiles/tools/patches/raycast_shared_loads.patch: a device switch that makes the second sample of each of the
# normal's three pairs reuse the first one's corners), applied for this build only; the product source carries none of it.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
patch -p1 -s < profiles/tools/patches/raycast_shared_loads.patch || exit 1
make -C x-slam_amd/csrc EXTRAFLAGS="-DXS_EXPERIMENTS -DXS_RAY_SHARED_LOADS" > /dev/null 2>&1 || exit 1
timeout -k 10 300 python3 profiles/tools/probe_raycast_shared.py 2>&1 | grep -v amdgpu.ids
patch -p1 -s -R < profiles/tools/patches/raycast_shared_loads.patch
make -C x-slam_amd/csrc > /dev/null 2>&1
