#!/bin/bash
# A/B of the ICP fold location on the GPU box (run from the repository root): bench track workload, device fold vs host fold,
# three runs each, interleaved; then the per-stage times of both.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
for i in 1 2 3; do
  for mode in device host; do
    timeout -k 10 120 python bench.py --workload track --no-s2 --no-cpu-baseline --icp-fold $mode > gpurun_out/ab_fold_${mode}_$i.json 2> gpurun_out/ab_fold_${mode}_$i.err
    rc=$?; [ $rc -ge 124 ] && { echo "killed"; exit $rc; }
    python - <<PY
import json
d=json.loads(open("gpurun_out/ab_fold_${mode}_$i.json").read().strip().splitlines()[-1])
print("$mode $i", d["value"], d["repetitions_fps"], d["stages_ms"]["icp"] if d["stages_ms"] else None)
PY
  done
done
