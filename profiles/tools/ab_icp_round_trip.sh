#!/bin/bash
# GPU box, repository root: what stands between two ICP reductions, per pyramid level, and what a solve on the device would put there instead.
#   (1) iteration period inside the timed region with the next frame announced (default) and without (--no-look-ahead: no bilateral filter
#       on the same SIMDs), host solve + posted pose;
#   (2) the reduction kernels alone (kernel trace of 50 back-to-back launches per level);
#   (3) the same pipeline with the pose update on the device (--icp-solve device: k_icp + k_icp_solve per iteration, no host in the loop),
#       under the kernel trace: k_icp_solve's own duration = the device's serial Cholesky chain.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
OUT=gpurun_out/r04_ab_icp_round_trip.txt; : > $OUT
P='import json,sys
d=json.loads(sys.stdin.read()); i=d["icp_us_per_iteration"]
print("fps", d["value"], d["repetitions_fps"], "| period us: level0", i["level0"], "level1", i["level1"], "level2", i["level2"], "first", i["first_iteration_of_frame"], "| icp stage ms", d["stages_ms"]["icp"])'
for rep in 1 2; do
  for v in "" "--no-look-ahead" "--icp-solve device" "--icp-solve device --no-look-ahead"; do
    echo "== bench.py --workload track $v (round $rep)" >> $OUT
    timeout -k 10 240 python3 bench.py --workload track --no-s2 --no-cpu-baseline --no-legs --steps 100 $v 2>/dev/null | python3 -c "$P" >> $OUT || exit 1
  done
done
echo "== reduction kernels alone (profiles/tools/time_icp_kernels.sh)" >> $OUT
bash profiles/tools/time_icp_kernels.sh >> $OUT 2>&1 || exit 1
echo "== kernel trace of bench.py --workload track --icp-solve device --no-look-ahead" >> $OUT
rm -rf gpurun_out/icp_dev; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/icp_dev -- python3 bench.py --workload track --no-s2 --no-cpu-baseline --no-legs --steps 60 --icp-solve device --no-look-ahead > gpurun_out/icp_dev.log 2>&1 || { tail -5 gpurun_out/icp_dev.log; exit 1; }
python3 - >> $OUT <<'PY'
import csv, glob
f = glob.glob("gpurun_out/icp_dev/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_icp" in r["Name"]:
        print(f'{r["Name"][:60]:60s} calls {r["Calls"]:>6s}  avg {float(r["AverageNs"]) / 1e3:7.2f} us  min {float(r["MinNs"]) / 1e3:7.2f}  max {float(r["MaxNs"]) / 1e3:7.2f}')
PY
cat $OUT
