#!/bin/bash
# GPU box, repository root: the integrate kernels' r04 knobs — planes per request group of the free-space path (XS_FREE_CHUNK) and the
# walk's all-planes LDS-DMA prefetch (XS_WALK_PREFETCH) — on S1 (24 tracked frames at 512^3) and the S2 probe; product build restored
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
F="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
OUT=gpurun_out/ab_integrate_r4.txt; : > $OUT
IFS=';' read -ra VARS <<< "${VARIANTS:--DXS_FREE_CHUNK=4;-DXS_FREE_CHUNK=2;-DXS_FREE_CHUNK=8;-DXS_WALK_PREFETCH=1;-DXS_FREE_CHUNK=4}"
for d in "${VARS[@]}"; do
  touch x-slam_amd/csrc/xs_tsdf.hip; make -C x-slam_amd/csrc HIPFLAGS="$F $d" > /dev/null 2>&1 || { echo "build failed $d"; exit 1; }
  echo "== $d" >> $OUT
  timeout -k 10 120 python3 profiles/tools/probe_integrate.py 2>/dev/null | tail -1 >> $OUT || exit 1
  timeout -k 10 120 python3 profiles/tools/probe_s2_r4.py 20 2>/dev/null | tail -1 >> $OUT || exit 1
done
touch x-slam_amd/csrc/xs_tsdf.hip; make -C x-slam_amd/csrc > /dev/null 2>&1
cat $OUT
