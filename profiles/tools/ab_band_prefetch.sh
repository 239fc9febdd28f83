#!/bin/bash
# GPU box: the Hessian / loss / Gauss-Newton kernels' scan of the dense ground-truth TSDF with the next chunk of 32 planes requested before the
# current one is scanned (-DXS_EXPERIMENTS -DXS_BAND_PREFETCH: two register sets, one wave per SIMD fewer) against the product build.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
for rep in 1 2; do
  for v in product prefetch; do
    if [ $v = prefetch ]; then F="-DXS_EXPERIMENTS -DXS_BAND_PREFETCH"; else F=""; fi
    touch x-slam_amd/csrc/xs_tsdf.hip; make -C x-slam_amd/csrc EXTRAFLAGS="$F" > /dev/null 2>&1 || exit 1
    echo "== $v =="
    timeout -k 10 200 python3 profiles/tools/probe_hess.py 2>&1 | grep "^{"
    timeout -k 10 200 python3 profiles/tools/probe_gn.py 1024 2>&1 | grep "^{"
  done
done
touch x-slam_amd/csrc/xs_tsdf.hip; make -C x-slam_amd/csrc > /dev/null 2>&1
