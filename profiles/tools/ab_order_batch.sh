#!/bin/bash
# GPU box: bricks per reservation of k_classify_boxes beyond 4 096 listed bricks (XS_ORDER_BATCH, compile time: 64 / 32 / 16): a workgroup takes its
# batch in rounds of 8 bricks, so 64 is eight dependent rounds; fewer bricks per batch = fewer rounds but more same-address atomics.
# 1024^3 tracking (12.5 K bricks) and the S2 whole call (23 K bricks).
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
for b in 64 32 16 64 32; do
  touch x-slam_amd/csrc/xs_tsdf.hip; make -C x-slam_amd/csrc EXTRAFLAGS="-DXS_ORDER_BATCH=$b" > /dev/null 2>&1 || exit 1
  timeout -k 10 300 python3 bench.py --workload track --size 1024 --steps 40 --warmup 5 --no-cpu-baseline --no-legs 2>/dev/null | python3 -c "
import json,sys
p=json.loads(sys.stdin.read().strip().splitlines()[-1]); s2=p['roofline_s2']
print('batch $b: 1024^3 frames/s', p['repetitions_fps'], 'integrate kernel ms', p['roofline']['kernel_ms'], '| S2 kernel', s2['kernel_ms'], 'whole call', s2['whole_call_ms'], 'first touch', s2['first_touch']['kernel_ms'])"
done
touch x-slam_amd/csrc/xs_tsdf.hip; make -C x-slam_amd/csrc > /dev/null 2>&1
