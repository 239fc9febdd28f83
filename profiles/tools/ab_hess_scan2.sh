#!/bin/bash
# Round 6 A/B, second series (GPU box, repository root, library built with EXTRAFLAGS=-DXS_EXPERIMENTS): granularity of the plane interleave
# of the wide scan — XS_HESS_IL consecutive planes per workgroup before its neighbours' — at 1024 workgroups.  Alternating, two rounds.
for round in 1 2; do
  for v in "wide_runs:XS_HESS_INTERLEAVE=0 XS_HESS_BLOCKS=1024" "il1:XS_HESS_IL=1 XS_HESS_BLOCKS=1024" "il2:XS_HESS_IL=2 XS_HESS_BLOCKS=1024" "il4:XS_HESS_IL=4 XS_HESS_BLOCKS=1024" "il8:XS_HESS_IL=8 XS_HESS_BLOCKS=1024"; do
    name=${v%%:*}; envs=${v#*:}
    echo "== $name (round $round)"
    env $envs python profiles/tools/probe_hess.py 2>/dev/null | grep -v amdgpu.ids
    env $envs python profiles/tools/probe_gn.py 512 2>/dev/null | grep '"n"'
  done
done
