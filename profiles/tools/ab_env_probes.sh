#!/bin/bash
# GPU box: the three standalone integrate probes with an environment variable set to each of the given values, twice; usage:
#   bash profiles/tools/ab_env_probes.sh XS_BRICK_GRID 8192 2048 4096
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
V=$1; shift
for rep in 1 2; do for val in "$@"; do
  echo "== $V=$val (round $rep)"
  env $V=$val timeout -k 10 120 python3 profiles/tools/probe_integrate.py 2>/dev/null | tail -1 | cut -c1-100
  env $V=$val timeout -k 10 120 python3 profiles/tools/probe_s2_r4.py 20 2>/dev/null | tail -1
  env $V=$val XS_PROBE_N=1024 timeout -k 10 120 python3 profiles/tools/probe_integrate.py 2>/dev/null | tail -1 | cut -c1-100
done; done
