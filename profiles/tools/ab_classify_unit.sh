#!/bin/bash
# GPU box: the fused classification's dealing unit (runs of that many x-neighbouring bricks per workgroup): the integrate probes and the
# classification's own time (rocprofv3 kernel trace of the S1 probe)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
for u in 1 4 16 1 4 16; do
  echo "== XS_CLASSIFY_UNIT=$u"
  XS_CLASSIFY_UNIT=$u timeout -k 10 120 python3 profiles/tools/probe_integrate.py 2>/dev/null | tail -1 | cut -c1-100
  XS_CLASSIFY_UNIT=$u timeout -k 10 120 python3 profiles/tools/probe_s2_r4.py 20 2>/dev/null | tail -1
  XS_CLASSIFY_UNIT=$u XS_PROBE_N=1024 timeout -k 10 120 python3 profiles/tools/probe_integrate.py 2>/dev/null | tail -1 | cut -c1-100
done
for u in 1 4 16; do
  export XS_CLASSIFY_UNIT=$u
  rm -rf gpurun_out/cu_trace; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cu_trace -- python3 profiles/tools/probe_integrate.py > /dev/null 2>&1
  echo "== unit $u: S1 512^3"; grep -h "k_classify\|k_integrate_bricks" $(find gpurun_out/cu_trace -name '*kernel_stats.csv') | cut -d, -f1-4 | cut -c1-120
  rm -rf gpurun_out/cu_trace; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cu_trace -- python3 profiles/tools/probe_s2_r4.py 20 > /dev/null 2>&1
  echo "== unit $u: S2"; grep -h "k_classify\|k_integrate_bricks" $(find gpurun_out/cu_trace -name '*kernel_stats.csv') | cut -d, -f1-4 | cut -c1-120
done
rm -rf gpurun_out/cu_trace
