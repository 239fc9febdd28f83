#!/bin/bash
# GPU box: the fused classification against bricks-then-boxes (k_classify_boxes: reservations per batch of list neighbours), same box, alternating
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
for rep in 1 2; do for two in 0 1; do
  if [ $two = 1 ]; then export XS_CLASSIFY_TWO_KERNELS=1; else unset XS_CLASSIFY_TWO_KERNELS; fi
  echo "== two kernels: $two (round $rep)"
  timeout -k 10 120 python3 profiles/tools/probe_integrate.py 2>/dev/null | tail -1 | cut -c1-100
  timeout -k 10 120 python3 profiles/tools/probe_s2_r4.py 20 2>/dev/null | tail -1
  XS_PROBE_N=1024 timeout -k 10 120 python3 profiles/tools/probe_integrate.py 2>/dev/null | tail -1 | cut -c1-100
done; done
