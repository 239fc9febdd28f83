#!/bin/bash
# GPU box, repository root: SQ counters of the ICP reduction kernel, one launch shape per level (profiles/tools/probe_icp.py)
OUT=${1:-gpurun_out/pmc_icp}; mkdir -p $OUT; export TMPDIR=/tmp
C="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU"
rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/icp -- python3 profiles/tools/probe_icp.py > $OUT/icp.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, os, sys, json
out = sys.argv[1]
acc = {}
for f in glob.glob(os.path.join(out, "icp", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_icp<" in row["Kernel_Name"]:
            key = (row["Kernel_Name"].split("(")[0], row.get("Grid_Size") or row.get("Workgroup_Size"))
            acc.setdefault(key, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
for key, cs in sorted(acc.items()):
    print(key, json.dumps({k: round(sum(v) / len(v)) for k, v in sorted(cs.items())}), "launches", len(next(iter(cs.values()))))
PY
