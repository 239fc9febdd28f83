#!/bin/bash
# GPU box: a brick whose boxes walk >= XS_SPLIT_MIN_PLANES planes taken as two list entries (XS_SPLIT_HEAVY, list_append in xs_tsdf.hip) against
# bricks taken whole, alternating builds on one box: the S1 integrate kernel alone (probe_edge.py: volumes compared with the walk everywhere,
# bit for bit) and the tracking pipeline (bench.py --workload track).
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
for v in ${VARIANTS:-"0 6" "1 6" "1 4" "0 6" "1 6"}; do
  set -- $v
  touch x-slam_amd/csrc/xs_tsdf.hip; make -C x-slam_amd/csrc EXTRAFLAGS="-DXS_SPLIT_HEAVY=$1 -DXS_SPLIT_MIN_PLANES=$2" > /dev/null 2>&1 || exit 1
  echo "== XS_SPLIT_HEAVY=$1 XS_SPLIT_MIN_PLANES=$2 =="
  timeout -k 10 200 python3 profiles/tools/probe_edge.py 512 2>&1 | grep -E "^n 512 (own|ahead)|identical" | cut -c1-230 || exit 1
  [ -n "$NO_TRACK" ] || timeout -k 10 200 python3 bench.py --workload track --steps 20 --warmup 5 --no-s2 --no-cpu-baseline --no-legs 2>/dev/null | python3 -c "
import json,sys
p=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('   track: frames/s', p['repetitions_fps'], ' integrate kernel ms', p['roofline']['kernel_ms'], ' stages', {k: v for k, v in (p.get('stages_ms') or {}).items() if k in ('icp', 'integrate', 'raycast')})" || exit 1
done
touch x-slam_amd/csrc/xs_tsdf.hip; make -C x-slam_amd/csrc > /dev/null 2>&1
