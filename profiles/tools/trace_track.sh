#!/bin/bash
# GPU box, repository root: kernel trace of the track workload only; prints the per-kernel averages
cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
OUT=gpurun_out/prof_track; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/track -- python3 bench.py --workload track --no-s2 --no-cpu-baseline > $OUT/track.log 2>&1 || exit $?
find $OUT/track -name '*kernel_stats.csv' -exec cp {} $OUT/track_kernel_stats.csv \;
cut -d, -f1-4 $OUT/track_kernel_stats.csv | cut -c1-150 | head -16
