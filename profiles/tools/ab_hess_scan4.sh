#!/bin/bash
# Round 6 A/B, fourth series (GPU box, repository root, library built with -DXS_EXPERIMENTS): the number of interleaved z groups G per tile column
# (XS_HESS_GROUPS) — is the slower scan of the interleaved form at 512^3 (0.0985 ms against 0.0888 for runs; none at 1024^3) a matter of
# power-of-two strides between a batch's requests?  Lines: all-zero gt (scan alone) Gauss-Newton / Hessian / loss, then the probes with a real map.
for g in 4 3 5 6 2 8; do
  for il in 2 1; do
    echo "== G=$g il=$il"
    XS_HESS_GROUPS=$g XS_HESS_IL=$il python profiles/tools/probe_residual_scan_alone.py 512 2>/dev/null | tr '\n' ' '; echo
    XS_HESS_GROUPS=$g XS_HESS_IL=$il python profiles/tools/probe_hess.py 2>/dev/null | head -1
    XS_HESS_GROUPS=$g XS_HESS_IL=$il python profiles/tools/probe_gn.py 512 2>/dev/null | head -1
  done
done
