#!/bin/bash
# GPU box, repository root: the two N^3-bound workloads (Hessian pass per frame at 512^3; Gauss-Newton relocalisation at 1024^3), one line each
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
for w in hessian reloc; do
  timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline "$@" > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err
  rc=$?; [ $rc -ge 124 ] && { echo "killed"; exit $rc; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/bench_$w.json").read().strip().splitlines()[-1])
print("$w", d["value"], d["unit"], {k: d[k] for k in d if k.startswith("roofline")})
PY
done
