#!/bin/bash
# GPU box: how long the classification kernels take on the three probe scenes, per XS_CLASSIFY_UNIT given as arguments
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
for u in "$@"; do
  export XS_CLASSIFY_UNIT=$u
  for sc in s1 s2 s1_1024; do
    rm -rf gpurun_out/cu_trace
    case $sc in
      s1) timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cu_trace -- python3 profiles/tools/probe_integrate.py > /dev/null 2>&1 ;;
      s2) timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cu_trace -- python3 profiles/tools/probe_s2_r4.py 20 > /dev/null 2>&1 ;;
      s1_1024) XS_PROBE_N=1024 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cu_trace -- python3 profiles/tools/probe_integrate.py > /dev/null 2>&1 ;;
    esac
    echo "== unit $u, $sc"; python3 profiles/tools/kstat.py gpurun_out/cu_trace k_classify k_integrate_bricks
  done
done
rm -rf gpurun_out/cu_trace
