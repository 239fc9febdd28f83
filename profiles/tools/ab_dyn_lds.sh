#!/bin/bash
# GPU box: S1 integrate kernel with the resident workgroups per CU capped through dynamic LDS (XS_INTEGRATE_DYN_LDS bytes per workgroup)
cd "$(dirname "$0")/../.."
for rep in 1 2; do for lds in 0 20000 26000 32000 40000 53000 80000; do
  echo -n "dyn lds $lds: "; XS_INTEGRATE_DYN_LDS=$lds XS_BRICK_GRID=${GRID:-0} timeout -k 10 120 python3 profiles/tools/probe_integrate.py 2>/dev/null | tail -1 | cut -c1-80 || exit 1
done; done
