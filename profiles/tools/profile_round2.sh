#!/bin/bash
# Round-2 profiles, run on the GPU box from the repository root:  bash profiles/tools/profile_round2.sh
# rocprofv3 --kernel-trace --stats of (a) the default bench, (b) the S2 integrate probe alone, (c) the Hessian / loss /
# Gauss-Newton probe (512^3, and GN at 1024^3); then the counter passes of collect_pmc.sh (separate --pmc runs, no other
# trace domains).  Summaries land under gpurun_out/prof_r02/; the ones to keep are copied to profiles/ by hand.
set -u
cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
OUT=gpurun_out/prof_r02; mkdir -p $OUT
run() { local name=$1; shift; echo "== $name"; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- "$@" > $OUT/$name.log 2>&1; local rc=$?; echo "$name rc=$rc"; [ $rc -ge 124 ] && exit $rc; find $OUT/$name -name '*kernel_stats.csv' -exec cp {} $OUT/${name}_kernel_stats.csv \; ; tail -2 $OUT/$name.log | cut -c1-600; }
run bench python3 bench.py --no-cpu-baseline --no-csfd
run s2_probe python3 profiles/tools/probe_s2.py 20
run hess_probe python3 profiles/tools/probe_hess.py
run gn_512 python3 profiles/tools/probe_gn.py 512
run gn_1024 python3 profiles/tools/probe_gn.py 1024
timeout -k 10 900 bash profiles/tools/collect_pmc.sh $OUT/pmc > $OUT/pmc.log 2>&1; echo "pmc rc=$?"; tail -30 $OUT/pmc/summary.json
head -12 $OUT/*_kernel_stats.csv
