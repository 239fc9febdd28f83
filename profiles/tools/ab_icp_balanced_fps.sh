#!/bin/bash
# GPU box, repository root: frames/s of the headline workload, level-0 ICP as 512 balanced eight-wave workgroups (1, default), 256 sixteen-wave
# ones (2) or 600 one-tile-per-wave ones (0); no profiler
cd "$(dirname "$0")/../.."
for mode in ${MODES:-1 0 1 0 1 0}; do
  export XS_ICP_BALANCED=$mode
  timeout -k 10 300 python3 bench.py --workload track --no-s2 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('XS_ICP_BALANCED=$mode fps', d['value'], d['repetitions_fps'], 'level0 us', d['icp_us_per_iteration']['level0'], 'icp stage ms', d['stages_ms']['icp'], 'sustained', d['sustained']['frames_per_s'])" || exit 1
done
