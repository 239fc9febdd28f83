#!/bin/bash
# GPU box: how many planes a streaming wave has in flight — the rolling pipeline of the free-space column with 1 or 2 planes per group
# (XS_FREE_CHUNK) against all of the column's planes at once by LDS-DMA (XS_FREE_DMA) — alternating builds on one box:
# the S1 kernel alone (probe_edge.py 512 / 1024: volumes against the walk everywhere, bit for bit) and the whole bench line.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
for v in ${VARIANTS:-"1 0" "1 1" "2 0" "1 0" "1 1"}; do
  set -- $v
  touch x-slam_amd/csrc/xs_tsdf.hip; make -C x-slam_amd/csrc EXTRAFLAGS="-DXS_FREE_CHUNK=$1 -DXS_FREE_DMA=$2" > /dev/null 2>&1 || exit 1
  echo "== XS_FREE_CHUNK=$1 XS_FREE_DMA=$2 =="
  timeout -k 10 300 python3 profiles/tools/probe_edge.py 512 1024 2>&1 | grep -E "^n (512|1024) (own|ahead)|identical" | cut -c1-100 | sort | uniq -c || exit 1
  timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
p=json.loads(sys.stdin.read().strip().splitlines()[-1])
s2=p.get('roofline_s2') or {}; l=(p.get('legs') or {}); t=l.get('track_1024') or {}; b=p.get('bilinear') or {}
print('   frames/s', p['value'], p.get('repetitions_fps'), '| integrate kernel ms', p['roofline']['kernel_ms'])
print('     1024^3:', {k: t.get(k) for k in ('frames_per_s', 'integrate_kernel_ms')})
print('     S2: kernel', s2.get('kernel_ms'), 'frac', s2.get('frac'), 'whole call', s2.get('whole_call_ms'), 'first touch', (s2.get('first_touch') or {}).get('kernel_ms'), 'every word', (s2.get('every_word_stored') or {}).get('kernel_ms'), 'noisy', (s2.get('noisy') or {}).get('kernel_ms'))
print('     bilinear:', {k: b.get(k) for k in ('frames_per_s', 'integrate_kernel_ms')})" || exit 1
done
touch x-slam_amd/csrc/xs_tsdf.hip; make -C x-slam_amd/csrc > /dev/null 2>&1
