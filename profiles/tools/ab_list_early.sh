#!/bin/bash
# GPU box: the brick list at the frame's start (integrate_list_early, round 5) against both classification kernels behind the last ICP
# launch (round 4's order), alternating, at 512^3 and 1024^3; frames/s of three timed regions each, the integrate kernel, how often the list held.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
run() { timeout -k 10 200 python3 bench.py --workload track --size $2 --steps 60 --warmup 5 --no-s2 --no-cpu-baseline --no-legs --param integrate_list_early=$1 ${3:+--param integrate_list_slack=$3} 2>/dev/null | python3 -c "
import json,sys
p=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('size $2 integrate_list_early=$1 slack ${3:-8}: frames/s', p['repetitions_fps'], 'integrate kernel ms', p['roofline']['kernel_ms'], 'held', p['config']['classes_decided_ahead_held'], 'tail host', p['tail_host_us']['timed_region_profiling_level_1'])"; }
for rep in 1 2; do
  run true 512; run false 512
done
run true 512 4; run true 512 16
run true 1024; run false 1024
