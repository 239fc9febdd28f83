#!/bin/bash
# GPU box: kernel trace of the S2 integrate probe (every launch of the whole call)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
rm -rf gpurun_out/ts2; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ts2 -- python3 profiles/tools/probe_s2.py 10 > gpurun_out/ts2.log 2>&1; tail -1 gpurun_out/ts2.log
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/ts2/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    print(r["Name"][:70].ljust(70), r["Calls"].rjust(5), r["AverageNs"][:9].rjust(10), r["MinNs"].rjust(8), r["MaxNs"].rjust(8))
PY
