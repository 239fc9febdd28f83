#!/bin/bash
# Round 6: does a row pitch that is no power of two (plane stride 1 MiB + pad x 2 KiB at 512^3) change the integrate kernel's S2 regimes?  (The residual
# kernels' scan lost 10 % to a power-of-two distance between a lane's requests: profiles/r06_hess_scan.txt 7.)  GPU box, repository root, product library.
for round in 1 2; do
  for pad in 0 64 32 96; do
    echo "== pitch 512 + $pad floats (round $round)"
    PROBE_PITCH_PAD_FLOATS=$pad python profiles/tools/probe_s2_modes.py 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    try: d = json.loads(line)
    except Exception: continue
    if 'regime' in d: print('   %-55s median %.4f ms  min %.4f  frac %.3f' % (d['regime'][:55], d['kernel_ms']['median'], d['kernel_ms']['min'], d['frac_of_8TBs_algorithmic']))
"
  done
done
