#!/bin/bash
# GPU box, repository root: durations of the three ICP kernel instances alone (kernel trace of profiles/tools/time_icp_kernels.py: 50 launches per level)
cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
O=gpurun_out/icp_time; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/p -- python3 profiles/tools/time_icp_kernels.py > $O/p.log 2>&1 || { tail -20 $O/p.log; exit 1; }
python3 - <<"PY"
import csv, glob, collections
f = glob.glob("gpurun_out/icp_time/p/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "k_icp" in r["Kernel_Name"]: d[(r["Kernel_Name"].split("(")[0], r["Grid_Size"] if "Grid_Size" in r else "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items()):
    v = v[5:]
    print(f"{k[0]:34s} grid {k[1]:>8s}  launches {len(v):3d}  avg {sum(v) / len(v):6.2f} us  min {min(v):6.2f}  median {sorted(v)[len(v) // 2]:6.2f}")
PY
