#!/bin/bash
# Kernel traces of the final tree (GPU box, repository root): the track workload alone (k_integrate_bricks = the S1 launches only: the figure
# to set beside bench.py's roofline.kernel_ms), the Hessian / loss probe and the Gauss-Newton probes after the band queue; one frame's timeline.
set -u
cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
OUT=gpurun_out/prof_final; mkdir -p $OUT
run() { local name=$1; shift; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- "$@" > $OUT/$name.log 2>&1; local rc=$?; echo "$name rc=$rc"; [ $rc -ge 124 ] && exit $rc; find $OUT/$name -name '*kernel_stats.csv' -exec cp {} $OUT/${name}_kernel_stats.csv \; ; grep -h "^{" $OUT/$name.log | tail -1 | cut -c1-400; }
run track python3 bench.py --workload track --no-s2 --no-cpu-baseline
run hess_probe python3 profiles/tools/probe_hess.py
run gn_512 python3 profiles/tools/probe_gn.py 512
run gn_1024 python3 profiles/tools/probe_gn.py 1024
run reloc python3 bench.py --workload reloc --steps 20 --warmup 2
python3 profiles/tools/frame_timeline.py $(find $OUT/track -name '*kernel_trace.csv') 100 > $OUT/frame_timeline.txt 2>&1
head -8 $OUT/*_kernel_stats.csv | cut -c1-200
