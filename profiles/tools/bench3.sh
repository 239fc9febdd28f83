#!/bin/bash
# three runs of the track workload (scene S1 512^3, 200 timed frames x 2 repetitions each), optional extra bench flags
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
for i in 1 2 3; do
  timeout -k 10 120 python bench.py --workload track --no-s2 --no-cpu-baseline "$@" > gpurun_out/bench3_$i.json 2> gpurun_out/bench3_$i.err
  rc=$?; [ $rc -ge 124 ] && { echo "killed"; exit $rc; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/bench3_$i.json").read().strip().splitlines()[-1])
print("run $i", d["value"], d["repetitions_fps"], d["stages_ms"], d["roofline"]["kernel_ms"], d["roofline"]["frac"])
PY
done
