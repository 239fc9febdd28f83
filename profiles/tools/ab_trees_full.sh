#!/bin/bash
# GPU box: two built trees (A = scratch/prev, see ab_trees.sh; B = the working tree) alternating on one box, the whole default bench.py line of each:
# headline, integrate kernel, the 1024^3 leg, scene S2's regimes, the bilinear run.
cd "$(dirname "$0")/../.."
ROOT=$PWD
run() { (cd $1 && timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null) | python3 -c "
import json,sys
p=json.loads(sys.stdin.read().strip().splitlines()[-1])
s2=p.get('roofline_s2') or {}; l=(p.get('legs') or {}); t=l.get('track_1024') or {}; b=p.get('bilinear') or {}
print('$2: frames/s', p['value'], p.get('repetitions_fps'), '| integrate kernel ms', p['roofline']['kernel_ms'], 'frac', p['roofline']['frac'])
print('     1024^3:', {k: t.get(k) for k in ('frames_per_s', 'integrate_kernel_ms')}, (t.get('stages_ms') or {}).get('integrate'))
print('     S2: kernel', s2.get('kernel_ms'), 'frac', s2.get('frac'), 'whole call', s2.get('whole_call_ms'), 'first touch', (s2.get('first_touch') or {}).get('kernel_ms'), 'every word', (s2.get('every_word_stored') or {}).get('kernel_ms'), 'noisy', (s2.get('noisy') or {}).get('kernel_ms'))
print('     bilinear:', {k: b.get(k) for k in ('frames_per_s', 'integrate_kernel_ms')}, '| sustained', (l.get('sustained') or {}).get('frames_per_s'), '| no look-ahead', (l.get('no_look_ahead') or {}).get('frames_per_s'))"; }
for rep in 1 2; do
  run $ROOT/scratch/prev "${1:-previous}"
  run $ROOT "${2:-current }"
done
