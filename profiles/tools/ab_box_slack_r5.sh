#!/bin/bash
# GPU box, repository root: the axial / lateral pads of box classes decided ahead of the final pose (box_slack in xs_tsdf.hip: lateral 2.0,
# axial 0.3 of the frustum slack) against frames/s, the integrate kernel and how often the classes held, at 512^3 and 1024^3.
# Builds an XS_EXPERIMENTS library (the switches are environment variables there only) and restores the product build afterwards.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
touch x-slam_amd/csrc/xs_tsdf.hip; make -C x-slam_amd/csrc EXTRAFLAGS=-DXS_EXPERIMENTS > /dev/null 2>&1 || exit 1
run() {
  XS_BOX_SLACK_AXIAL=$1 XS_BOX_SLACK_LATERAL=$2 timeout -k 10 200 python3 bench.py --workload track --size $3 --steps 60 --warmup 5 --no-s2 --no-cpu-baseline --no-legs 2>/dev/null | python3 -c "
import json,sys
p=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('axial $1 lateral $2 size $3: fps', p['value'], 'median', p['value_median_of_repetitions'], 'integrate kernel ms', p['roofline']['kernel_ms'], 'held', p['config']['classes_decided_ahead_held'])"
}
for size in 512 1024; do
  for ax in 0.3 0.15 0.08; do run $ax 2.0 $size; done
  run 0.3 1.0 $size
  run 0.15 1.0 $size
done
touch x-slam_amd/csrc/xs_tsdf.hip; make -C x-slam_amd/csrc > /dev/null 2>&1
