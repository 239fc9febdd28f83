#!/bin/bash
# GPU box, repository root:  bash profiles/tools/collect_pmc_r5.sh [out_dir]
# Round 5: HBM traffic of the integrate kernel (FETCH_SIZE and WRITE_SIZE in separate --pmc passes, kernel trace only; corrected with the factors
# a known-bytes kernel of the same access width gives on the same box: calib_stream.hip) on
#   s2     scene S2 (probe_s2_pmc.py: the first launch into an empty volume, then repeats of the same frame: the steady state of a static camera)
#   track  the headline workload (bench.py --workload track: k_integrate_bricks<false,.> in the pipeline, every frame a new pose) and its
#          bilinear leg (k_integrate_bricks<true,.>)
# + the SQ counters of the S2 launches.  profiles/tools/pmc_summary_r5.py folds the CSVs into JSON.
set -e
OUT=${1:-gpurun_out/pmc_r5}
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
