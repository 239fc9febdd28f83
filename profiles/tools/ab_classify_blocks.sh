#!/bin/bash
# GPU box: the fused classification with contiguous blocks of bricks per workgroup (unit = bricks per workgroup) against the interleaved
# deal and against bricks-then-boxes; integrate probes (512^3 only: the workgroup count is given for that volume) + the classification's time
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
run() {
  echo "== $1"
  timeout -k 10 120 python3 profiles/tools/probe_integrate.py 2>/dev/null | tail -1 | cut -c1-100
  timeout -k 10 120 python3 profiles/tools/probe_s2_r4.py 20 2>/dev/null | tail -1
  rm -rf gpurun_out/cu_trace; XS_PROBE_NO_COUNT=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cu_trace -- python3 profiles/tools/probe_integrate.py > /dev/null 2>&1
  python3 profiles/tools/kstat.py gpurun_out/cu_trace k_classify
  rm -rf gpurun_out/cu_trace; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cu_trace -- python3 profiles/tools/probe_s2_r4.py 20 > /dev/null 2>&1
  python3 profiles/tools/kstat.py gpurun_out/cu_trace k_classify
}
for rep in 1 2; do
unset XS_CLASSIFY_TWO_KERNELS XS_CLASSIFY_UNIT XS_CLASSIFY_WGS
run "fused, interleaved (round $rep)"
XS_CLASSIFY_UNIT=64 XS_CLASSIFY_WGS=1024 run "fused, 1024 blocks of 64 (round $rep)"
XS_CLASSIFY_UNIT=128 XS_CLASSIFY_WGS=512 run "fused, 512 blocks of 128 (round $rep)"
XS_CLASSIFY_UNIT=32 XS_CLASSIFY_WGS=2048 run "fused, 2048 blocks of 32 (round $rep)"
XS_CLASSIFY_TWO_KERNELS=1 run "two kernels (round $rep)"
done
rm -rf gpurun_out/cu_trace
