#!/bin/bash
# GPU box: when every workgroup of the S1 integrate kernel begins and ends, per class and per CU (XS_PROBE_N = 512 | 1024).
# Builds an -DXS_EXPERIMENTS -DXS_WG_TIMES library and restores the product build afterwards.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
touch x-slam_amd/csrc/xs_tsdf.hip; make -C x-slam_amd/csrc EXTRAFLAGS="-DXS_EXPERIMENTS -DXS_WG_TIMES $EXTRA" > /dev/null 2>&1 || exit 1
for n in ${SIZES:-512 1024}; do
  echo "== n = $n =="
  XS_PROBE_N=$n timeout -k 10 200 python3 profiles/tools/probe_wg_times.py 2>&1 | grep -v amdgpu.ids | tail -40
done
touch x-slam_amd/csrc/xs_tsdf.hip; make -C x-slam_amd/csrc > /dev/null 2>&1
