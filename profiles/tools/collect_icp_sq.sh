#!/bin/bash
# GPU box, repository root: SQ / instruction-cache counters of the ICP kernels in the headline workload (separate passes; kernel trace only)
cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
OUT=gpurun_out/icp_sq; rm -rf $OUT; mkdir -p $OUT
pass() { local n=$1; shift; timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$n -- python3 bench.py --workload track --no-s2 --no-cpu-baseline --steps 60 > $OUT/$n.log 2>&1 || { tail -5 $OUT/$n.log; exit 1; }; }
pass a SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
pass b SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_IFETCH SQ_BUSY_CYCLES
python3 - <<'PY'
import csv, glob, collections
for n in "ab":
    f = glob.glob(f"gpurun_out/icp_sq/{n}/**/*counter_collection.csv", recursive=True)
    if not f: print("no csv for", n); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0]
        if not ("k_icp" in k or "k_raycast" in k or "k_integrate_bricks" in k): continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r["Dispatch_Id"])
        if key not in seen: seen.add(key); cnt[k] += 1
    for k in acc:
        print(k, "launches", cnt[k], "  ".join(f"{c} {v / cnt[k]:.4g}" for c, v in sorted(acc[k].items())))
PY
