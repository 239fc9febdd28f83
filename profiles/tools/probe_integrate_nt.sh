#!/bin/bash
# GPU box, repository root: the integrate kernel with non-temporal stores (global_store_dword ... nt) against plain ones, S2 + S1 probes,
# with and without the store elision; product build restored afterwards
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
F="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
OUT=gpurun_out/r03_integrate_nt.txt; : > $OUT
for d in "" "-DXS_PROBE_NT_STORES"; do
  touch x-slam_amd/csrc/xs_tsdf.hip; make -C x-slam_amd/csrc HIPFLAGS="$F $d" > /dev/null 2>&1
  for st in elide always; do
    if [ $st = always ]; then export XS_INTEGRATE_ALWAYS_STORE=1; else unset XS_INTEGRATE_ALWAYS_STORE; fi
    echo "== stores: ${d:-plain} $st" >> $OUT
    timeout -k 10 120 python profiles/tools/probe_s2.py 20 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-300 >> $OUT || exit 1
    timeout -k 10 120 python profiles/tools/probe_s1.py 20 only 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-300 >> $OUT || exit 1
  done
done
touch x-slam_amd/csrc/xs_tsdf.hip; make -C x-slam_amd/csrc > /dev/null 2>&1
cat $OUT
