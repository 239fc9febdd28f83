#!/bin/bash
# GPU box: bench.py's track (the headline pipeline) with an orchestrator parameter set two ways, alternating; usage:
#   bash profiles/tools/ab_param_track.sh integrate_classify_beside_icp true false [rounds]
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
V=$1; A=$2; B=$3; R=${4:-2}
for rep in $(seq 1 $R); do for val in $A $B; do
  echo "== $V=$val (round $rep)"
  timeout -k 10 240 python3 bench.py --workload track --no-cpu-baseline --no-s2 --no-legs --steps 100 --param $V=$val 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  pipeline fps', d['value'], d['repetitions_fps'], 'S1 kernel ms', d['roofline']['kernel_ms'], 'stages', {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d['stages_ms'].items()}, 'bilinear', d['bilinear']['frames_per_s'], d['bilinear']['integrate_kernel_ms'])" || exit 1
done; done
