#!/bin/bash
# GPU box, repository root: the integrate kernel posted its pose (--integrate-post: classification, gate and kernel go in while the last ICP launch runs) against launched after the final pose
cd "$(dirname "$0")/../.."
for v in post plain post plain post plain; do
  if [ $v = plain ]; then F=""; else F="--integrate-post"; fi
  timeout -k 10 300 python3 bench.py --workload track --no-s2 --no-cpu-baseline $F 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v fps', d['value'], d['repetitions_fps'], 'posted', d['config']['integrate_posted'], 'sustained', d['sustained']['frames_per_s'])" || exit 1
done
