#!/bin/bash
# GPU box, repository root: round-3 integrate kernels side by side on the S2 probe (the HBM claim) and the S1 probe
# (XS_INTEGRATE_KERNEL picks the instance: walk = round 2's kernel, gGwW = G planes per group at W waves per SIMD).
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
OUT=gpurun_out/r03_ab_integrate.txt; : > $OUT
for v in ${VARIANTS:-walk g8w8 g8w7 default g8w5 g4w8 g4w7 g4w6}; do
  for st in elide always; do
    if [ $st = always ]; then export XS_INTEGRATE_ALWAYS_STORE=1; else unset XS_INTEGRATE_ALWAYS_STORE; fi
    if [ $v = default ]; then unset XS_INTEGRATE_KERNEL; else export XS_INTEGRATE_KERNEL=$v; fi
    [ $v = walk ] && [ $st = always ] && continue
    echo "== $v $st" >> $OUT
    timeout -k 10 120 python profiles/tools/probe_s2.py 20 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-300 >> $OUT || exit 1
    timeout -k 10 120 python profiles/tools/probe_s1.py 20 only 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-300 >> $OUT || exit 1
  done
done
cat $OUT
