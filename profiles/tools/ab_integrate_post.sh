#!/bin/bash
# GPU box, repository root: frames/s of the headline workload with the integrate kernel posted its pose (default) against launched after it; no profiler
cd "$(dirname "$0")/../.."
for mode in post nopost post nopost post nopost; do
  if [ $mode = nopost ]; then F="--no-integrate-post"; else F=""; fi
  timeout -k 10 300 python3 bench.py --workload track --no-s2 --no-cpu-baseline $F 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$mode fps', d['value'], d['repetitions_fps'], 'integrate kernel ms', d['roofline']['kernel_ms'], 'stages', d['stages_ms']['integrate'], d['stages_ms']['raycast'], 'sustained', d['sustained']['frames_per_s'])" || exit 1
done
