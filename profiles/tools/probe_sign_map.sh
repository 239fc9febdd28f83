#!/bin/bash
# GPU box, repository root: the march kernel alone (kernel-trace durations) on a tracked 512^3 volume: every step against the sign-map start
cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
O=gpurun_out/prof_sm2; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/p -- python3 profiles/tools/probe_sign_map.py "$@" > $O/p.log 2>&1 || { tail -20 $O/p.log; exit 1; }
grep "^shift" $O/p.log
python3 - <<"PY"
import csv,glob
f=glob.glob("gpurun_out/prof_sm2/p/**/*kernel_trace.csv",recursive=True)[0]
for k in ("k_raycast<2","k_raycast<3"):
    rows=[r for r in csv.DictReader(open(f)) if k in r["Kernel_Name"]]
    d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows][-204:]
    for i,name in enumerate(("every step","map 8^3","map 16^3","map 32^3")):
        g=d[i*51:(i+1)*51]; print("%-12s %-11s avg %.1f us  min %.1f  max %.1f"%(k+">",name,sum(g)/len(g),min(g),max(g)))
PY
