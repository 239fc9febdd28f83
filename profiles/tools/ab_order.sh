#!/bin/bash
# GPU box: the brick list taken from its near end (plane order) or from its far end (KF_FAR_FIRST), S1 kernel and the S2 probe
cd "$(dirname "$0")/../.."
for rep in 1 2; do for o in near far; do
  echo "== order $o"; XS_INTEGRATE_ORDER=$o timeout -k 10 120 python3 profiles/tools/probe_integrate.py 2>/dev/null | tail -1 | cut -c1-80 || exit 1
  XS_INTEGRATE_ORDER=$o timeout -k 10 120 python3 profiles/tools/probe_s2_r4.py 20 2>/dev/null | tail -1 || exit 1
done; done
