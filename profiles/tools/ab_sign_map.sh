#!/bin/bash
# GPU box, repository root: frames/s of the headline workload with the ray march started from the sign map (default; bricks of 8^3 and 16^3
# voxels) against the march of every step; no profiler
cd "$(dirname "$0")/../.."
for mode in map3 full map4 map3 full map4; do
  case $mode in full) F="--no-sign-map";; map4) F="--sign-map-shift 4";; *) F="";; esac
  timeout -k 10 300 python3 bench.py --workload track --no-s2 --no-cpu-baseline $F 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$mode fps', d['value'], d['repetitions_fps'], 'integrate kernel ms', d['roofline']['kernel_ms'], 'stages', d['stages_ms']['integrate'], d['stages_ms']['raycast'], 'sustained', d['sustained']['frames_per_s'])" || exit 1
done
