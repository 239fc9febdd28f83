#!/bin/bash
# Round 6: nontemporal loads / stores for the integrate kernel's FREE planes (GPU box, repository root, library built with -DXS_EXPERIMENTS).
# XS_INTEGRATE_STREAM_NT: 0 = never (round 5), 1 = launches whose list exceeds 8192 bricks (scene S2: 23 K), 2 = always (scene S1 too: its
# state lives in the Infinity Cache from frame to frame).  Scene S2 by regime (probe_s2_modes.py), scene S1 alone (probe_edge.py).  Alternating.
for round in 1 2; do
  for nt in 0 1; do
    echo "== S2 512^3, XS_INTEGRATE_STREAM_NT=$nt (round $round)"
    XS_INTEGRATE_STREAM_NT=$nt python profiles/tools/probe_s2_modes.py 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    try: d = json.loads(line)
    except Exception: continue
    if 'regime' in d: print('   %-55s median %.4f ms  min %.4f  frac %.3f' % (d['regime'][:55], d['kernel_ms']['median'], d['kernel_ms']['min'], d['frac_of_8TBs_algorithmic']))
"
  done
  for nt in 0 2; do
    echo "== S1 512^3 / 1024^3, XS_INTEGRATE_STREAM_NT=$nt (round $round)"
    XS_INTEGRATE_STREAM_NT=$nt PROBE_QUICK=1 python profiles/tools/probe_edge.py 512 1024 2>/dev/null | grep "ahead  +sign" | cut -c1-100
  done
done
