#!/bin/bash
# GPU box, repository root: HIP_FORCE_DEV_KERNARG=1 / 0 / unset on the headline workload
cd "$(dirname "$0")/../.."
for v in unset 1 0 unset 1 0; do
  if [ $v = unset ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$v; fi
  timeout -k 10 300 python3 bench.py --workload track --no-s2 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); i=d['icp_us_per_iteration']; print('HIP_FORCE_DEV_KERNARG=$v fps', d['value'], d['repetitions_fps'], 'icp us', i['level0'], i['level1'], i['level2'], 'sustained', d['sustained']['frames_per_s'])" || exit 1
done
