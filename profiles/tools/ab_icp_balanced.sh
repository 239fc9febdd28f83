#!/bin/bash
# GPU box, repository root: level-0 ICP launches as 512 balanced workgroups (nine or ten tiles each) against 600 one-tile-per-wave ones
cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
for mode in 1 0 1 0; do
  OUT=gpurun_out/prof_icpbal_$mode; rm -rf $OUT; mkdir -p $OUT
  export XS_ICP_BALANCED=$mode
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/track -- python3 bench.py --workload track --no-s2 --no-cpu-baseline > $OUT/track.log 2>&1 || exit $?
  find $OUT/track -name '*kernel_stats.csv' -exec cp {} $OUT/track_kernel_stats.csv \;
  echo "== XS_ICP_BALANCED=$mode"; grep -h "k_icp" $OUT/track_kernel_stats.csv | cut -c1-110
  grep -h '^{' $OUT/track.log | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('fps', d['value'], d['repetitions_fps'], d['icp_us_per_iteration']['level0'], d['stages_ms']['icp'])"
done
