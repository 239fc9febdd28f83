#!/bin/bash
# GPU box: kernel timelines of single frames of the track workload with an orchestrator parameter set; usage: timeline_param.sh KEY=VALUE [frame ...]
cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
OUT=gpurun_out/timeline_$(echo $1 | tr '=' '_'); rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 bench.py --workload track --no-s2 --no-cpu-baseline --no-legs --steps 100 --param $1 > $OUT/run.log 2>&1 || exit 1
shift
for f in ${@:-100}; do
  echo "---- frame $f" >> $OUT/frame_timeline.txt
  python3 profiles/tools/frame_timeline.py $(find $OUT/t -name '*kernel_trace.csv') $f >> $OUT/frame_timeline.txt 2>&1
done
rm -rf $OUT/t
cat $OUT/frame_timeline.txt
