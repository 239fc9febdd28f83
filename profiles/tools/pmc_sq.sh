#!/bin/bash
# GPU box, repository root: SQ counters of the integrate kernel on scenes S2 and S1 (one --pmc pass each, kernel trace only)
OUT=${1:-gpurun_out/pmc_sq}; mkdir -p $OUT; export TMPDIR=/tmp
C="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU"
rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/s2 -- python3 profiles/tools/probe_s2.py 4 > $OUT/s2.log 2>&1
rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/s1 -- python3 profiles/tools/probe_s1.py 20 only > $OUT/s1.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, os, sys, json
out = sys.argv[1]
for sc in ("s2", "s1"):
    acc = {}
    for f in glob.glob(os.path.join(out, sc, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "k_integrate_bricks" in row["Kernel_Name"]:
                acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    print(sc, json.dumps({k: round(sum(v) / len(v)) for k, v in sorted(acc.items())}), "launches", {k: len(v) for k, v in acc.items()}.get("SQ_WAVES"))
PY
