#!/bin/bash
# GPU box: how wide the ahead-of-time box classes are padded (XS_BOX_SLACK_LATERAL x / XS_BOX_SLACK_AXIAL x the frustum planes' extra slack):
# wider = fewer frames whose classes must be decided again, more boxes on the per-voxel walk
cd "$(dirname "$0")/../.."
for rep in 1 2; do for v in "2.0 0.3" "1.0 0.3" "0.5 0.3" "1.0 0.15" "4.0 0.3"; do
  set -- $v
  echo -n "lateral $1 axial $2: "
  XS_BOX_SLACK_LATERAL=$1 XS_BOX_SLACK_AXIAL=$2 timeout -k 10 240 python3 bench.py --workload track --no-cpu-baseline --no-s2 --no-legs --steps 100 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('fps', d['value'], d['repetitions_fps'], 'S1 kernel ms', d['roofline']['kernel_ms'], 'integrate stage', d['stages_ms']['integrate'], 'bilinear', d['bilinear']['frames_per_s'], d['bilinear']['integrate_kernel_ms'])" || exit 1
done; done
