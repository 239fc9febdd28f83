#!/bin/bash
# GPU box, repository root: the next frame announced (default: its bilateral filter and depth pyramid run during this frame's ICP loop) against not
cd "$(dirname "$0")/../.."
for v in ahead plain ahead plain ahead plain; do
  if [ $v = plain ]; then F="--no-look-ahead"; else F=""; fi
  timeout -k 10 300 python3 bench.py --workload track --no-s2 --no-cpu-baseline $F 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); i=d['icp_us_per_iteration']; print('$v fps', d['value'], d['repetitions_fps'], 'icp us', i['level0'], i['level1'], i['level2'], 'first', i['first_iteration_of_frame'], 'sustained', d['sustained']['frames_per_s'])" || exit 1
done
