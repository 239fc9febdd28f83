#!/bin/bash
# GPU box, repository root: what the round-2 walk kernel costs without its memory traffic (S2 probe) — the instruction-issue floor.
#   nomem   = no state loads, no stores (the depth gathers stay);  nomem+nodepth = arithmetic alone
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
F="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
OUT=gpurun_out/r03_integrate_floor.txt; : > $OUT
export XS_INTEGRATE_KERNEL=walk
for d in "" "-DXS_PROBE_NOMEM" "-DXS_PROBE_NOMEM -DXS_PROBE_NODEPTH"; do
  touch x-slam_amd/csrc/xs_tsdf.hip; make -C x-slam_amd/csrc HIPFLAGS="$F $d" > /dev/null 2>&1
  echo "== walk $d" >> $OUT
  timeout -k 10 120 python profiles/tools/probe_s2.py 20 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-300 >> $OUT || exit 1
done
touch x-slam_amd/csrc/xs_tsdf.hip; make -C x-slam_amd/csrc > /dev/null 2>&1
cat $OUT
