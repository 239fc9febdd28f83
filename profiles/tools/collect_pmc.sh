#!/bin/bash
# Run on the GPU box from the repo root:  bash profiles/tools/collect_pmc.sh <out_dir>
# Separate counter passes (FETCH_SIZE and WRITE_SIZE do not fit one pass) for the integrate kernel on
# scenes S2 (frustum-filling) and S1 (the benchmark's scene, frame 20) and for the calibration kernel; profiles/tools/pmc_summary.py turns the CSVs into JSON.
set -e
OUT=${1:-gpurun_out/pmc}
mkdir -p $OUT
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o $OUT/calib_stream profiles/tools/calib_stream.hip
export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/s2_$C -- python3 profiles/tools/probe_s2.py 4 > $OUT/s2_$C.log 2>&1
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/s1_$C -- python3 profiles/tools/probe_s1.py 20 only > $OUT/s1_$C.log 2>&1
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/calib_$C -- $OUT/calib_stream > $OUT/calib_$C.log 2>&1
done
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU"; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/s2_SQ -- python3 profiles/tools/probe_s2.py 4 > $OUT/s2_SQ.log 2>&1
done
python3 profiles/tools/pmc_summary.py $OUT > $OUT/summary.json
cat $OUT/summary.json
