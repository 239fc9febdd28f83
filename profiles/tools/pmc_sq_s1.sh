#!/bin/bash
# GPU box, repository root: SQ counters of the S1 integrate launch on its own (profiles/tools/probe_integrate.py), one --pmc pass, kernel trace only
OUT=${1:-gpurun_out/pmc_sq_s1}; mkdir -p $OUT; export TMPDIR=/tmp
C="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU"   # (eight: two more and the profiler refuses the set, then hangs in its own shutdown)
timeout -k 10 150 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/s1 -- python3 profiles/tools/probe_integrate.py > $OUT/s1.log 2>&1
tail -1 $OUT/s1.log | cut -c1-120
python3 - $OUT <<'PY'
import csv, glob, os, sys, json
out = sys.argv[1]
for kern in ("k_integrate_bricks", "k_classify_boxes", "k_classify_bricks"):
    acc = {}
    for f in glob.glob(os.path.join(out, "s1", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if kern in row["Kernel_Name"]:
                acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    print(kern, json.dumps({k: round(sum(v) / len(v)) for k, v in sorted(acc.items())}), "launches", {k: len(v) for k, v in acc.items()}.get("SQ_WAVES"))
PY
