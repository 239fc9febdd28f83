#!/bin/bash
# GPU box: the ordered brick list taken alternately from both ends (XS_PROBE_ZIP = group size) against front to back
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
F="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
for v in "" "-DXS_PROBE_ZIP=8" "-DXS_PROBE_ZIP=256" "-DXS_PROBE_ZIP=2048" ""; do
  touch x-slam_amd/csrc/xs_tsdf.hip; make -C x-slam_amd/csrc HIPFLAGS="$F $v" > /dev/null 2>&1 || exit 1
  echo "== $v"
  timeout -k 10 120 python3 profiles/tools/probe_integrate.py 2>/dev/null | tail -1 | cut -c1-100 || exit 1
  timeout -k 10 120 python3 profiles/tools/probe_s2_r4.py 20 2>/dev/null | tail -1 || exit 1
  XS_PROBE_N=1024 timeout -k 10 120 python3 profiles/tools/probe_integrate.py 2>/dev/null | tail -1 | cut -c1-100 || exit 1
done
touch x-slam_amd/csrc/xs_tsdf.hip; make -C x-slam_amd/csrc > /dev/null 2>&1
