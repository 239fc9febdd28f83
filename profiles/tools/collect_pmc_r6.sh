#!/bin/bash
# GPU box, repository root:  bash profiles/tools/collect_pmc_r6.sh [out_dir]
# Round 6: HBM traffic of the integrate kernel (FETCH_SIZE and WRITE_SIZE in separate --pmc passes, kernel trace only; corrected with the factors
# a known-bytes kernel of the same access width gives on the same box: calib_stream.hip) on
#   s2     scene S2 (probe_s2_pmc.py: the first launch into an empty volume, then repeats of the same frame: the steady state of a static camera)
#   track  the headline workload (bench.py --workload track: k_integrate_bricks<false,.> in the pipeline, every frame a new pose) and its
#          bilinear leg (k_integrate_bricks<true,.>)
# + the SQ counters of the S2 launches.  profiles/tools/pmc_summary_r5.py folds the CSVs into JSON.
set -e
OUT=${1:-gpurun_out/pmc_r6}
mkdir -p $OUT
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o $OUT/calib_stream profiles/tools/calib_stream.hip
export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/s2_$C -- python3 profiles/tools/probe_s2_pmc.py > $OUT/s2_$C.log 2>&1
  S2_NOISY=1 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/s2_noisy_$C -- python3 profiles/tools/probe_s2_pmc.py > $OUT/s2_noisy_$C.log 2>&1
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/calib_$C -- $OUT/calib_stream > $OUT/calib_$C.log 2>&1
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/track_$C -- python3 bench.py --workload track --no-s2 --no-cpu-baseline --no-legs --steps 60 > $OUT/track_$C.log 2>&1
done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/s2_SQ -- python3 profiles/tools/probe_s2_pmc.py > $OUT/s2_SQ.log 2>&1
python3 profiles/tools/pmc_summary_r5.py $OUT > $OUT/summary.json
cat $OUT/summary.json
# round 6: the residual kernels' read (FETCH_SIZE of the Hessian / loss probe at 512^3 and of the Gauss-Newton pass at 1024^3)
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/hess_$C -- python3 profiles/tools/probe_hess.py > $OUT/hess_$C.log 2>&1
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/gn_$C -- python3 profiles/tools/probe_gn.py 1024 > $OUT/gn_$C.log 2>&1
done
python3 - $OUT <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
cal = json.load(open(os.path.join(out, "summary.json"))).get("calibration", {})
res = {}
for tag, kernels in (("hess", ("k_tsdf_hessian", "k_tsdf_loss")), ("gn", ("k_tsdf_gauss_newton",))):
    for C in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(os.path.join(out, f"{tag}_{C}", "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                for k in kernels:
                    if k in row["Kernel_Name"] and row["Counter_Name"] == C:
                        res.setdefault(k, {}).setdefault(C, []).append(float(row["Counter_Value"]))
summary = {k: {C: {"launches": len(v), "mean_raw": sum(v) / len(v)} for C, v in d.items()} for k, d in res.items()}
json.dump({"calibration_of_summary_json": cal, "raw_counter_units_as_collected": summary}, open(os.path.join(out, "residual_kernels_pmc.json"), "w"), indent=1)
print(json.dumps(summary))
PY
