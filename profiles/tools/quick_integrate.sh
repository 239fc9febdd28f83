#!/bin/bash
# GPU box: the integrate / sign-map tests, then the three standalone integrate probes (S1 512^3, S2, S1 1024^3)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_integrate_gpu.py tests/test_signmap_gpu.py -x -q -m gpu 2>&1 | tail -15 || exit 1
timeout -k 10 120 python3 profiles/tools/probe_integrate.py 2>/dev/null | tail -1 | cut -c1-100
timeout -k 10 120 python3 profiles/tools/probe_s2_r4.py 20 2>/dev/null | tail -1
XS_PROBE_N=1024 timeout -k 10 120 python3 profiles/tools/probe_integrate.py 2>/dev/null | tail -1 | cut -c1-100
