#!/bin/bash
# GPU box, repository root: with the sign map, the march + crossing as ONE launch (default) against two (XS_RAY_MAP_TWO_KERNELS=1); no profiler
cd "$(dirname "$0")/../.."
for mode in one two one two one two; do
  if [ $mode = two ]; then export XS_RAY_MAP_TWO_KERNELS=1; else unset XS_RAY_MAP_TWO_KERNELS; fi
  timeout -k 10 300 python3 bench.py --workload track --no-s2 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$mode fps', d['value'], d['repetitions_fps'], 'stages', d['stages_ms']['integrate'], d['stages_ms']['raycast'], 'sustained', d['sustained']['frames_per_s'], 'raycast alone ms', d['raycast']['ms_per_frame_alone'])" || exit 1
done
