#!/bin/bash
# Kernel traces of the round-6 tree (GPU box, repository root): the track workload (k_integrate_bricks<false,.> = the S1 launches, <true,.> = the
# bilinear leg's, the balanced level-0 ICP instance, the raycast pair), the S2 probe, the Hessian / Gauss-Newton probes; one frame's timeline.
set -u
cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
OUT=gpurun_out/prof_r06; mkdir -p $OUT
run() { local name=$1; shift; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- "$@" > $OUT/$name.log 2>&1; local rc=$?; echo "$name rc=$rc"; [ $rc -ge 124 ] && exit $rc; find $OUT/$name -name '*kernel_stats.csv' -exec cp {} $OUT/${name}_kernel_stats.csv \; ; grep -h "^{" $OUT/$name.log | tail -1 | cut -c1-300; }
run track python3 bench.py --workload track --no-s2 --no-cpu-baseline --no-legs
run driver_shape python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline
run s2_probe python3 profiles/tools/probe_s2_pmc.py
run hess_probe python3 profiles/tools/probe_hess.py
run gn_1024 python3 profiles/tools/probe_gn.py 1024
run reloc python3 bench.py --workload reloc --steps 20 --warmup 2
python3 profiles/tools/frame_timeline.py $(find $OUT/track -name '*kernel_trace.csv') 100 > $OUT/frame_timeline.txt 2>&1
head -12 $OUT/*_kernel_stats.csv | cut -c1-160
