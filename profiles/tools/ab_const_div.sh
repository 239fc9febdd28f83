#!/bin/bash
# GPU box, repository root: the tracking workload's kernel trace with the short division by launch constants on and off (same box)
cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
for mode in on off on off; do
  OUT=gpurun_out/prof_cd_$mode; rm -rf $OUT; mkdir -p $OUT
  if [ $mode = off ]; then export XS_CONST_DIV_OFF=1; else unset XS_CONST_DIV_OFF; fi
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/track -- python3 bench.py --workload track --no-s2 --no-cpu-baseline > $OUT/track.log 2>&1 || exit $?
  find $OUT/track -name '*kernel_stats.csv' -exec cp {} $OUT/track_kernel_stats.csv \;
  echo "== short division $mode"; grep -h "raycast\|integrate_bricks" $OUT/track_kernel_stats.csv | cut -d, -f1-4 | cut -c1-120
  grep -h '^{' $OUT/track.log | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('fps', d['value'], d['stages_ms'])"
done
