#!/bin/bash
# Round 6 A/B of the ground-truth scan of the Hessian / loss / Gauss-Newton kernels (GPU box, repository root, a library built with
# EXTRAFLAGS=-DXS_EXPERIMENTS): one column per lane (XS_HESS_NARROW) against four columns per lane with 16-byte loads, a workgroup's planes
# in runs (XS_HESS_INTERLEAVE=0) or interleaved with its neighbours' (=1), by number of workgroups (XS_HESS_BLOCKS).  Alternating, two rounds.
for round in 1 2; do
  for v in "narrow:XS_HESS_NARROW=1" "wide_runs_1024:XS_HESS_INTERLEAVE=0 XS_HESS_BLOCKS=1024" "wide_runs_4096:XS_HESS_INTERLEAVE=0 XS_HESS_BLOCKS=4096" \
           "wide_interleaved_1024:XS_HESS_INTERLEAVE=1 XS_HESS_BLOCKS=1024" "wide_interleaved_2048:XS_HESS_INTERLEAVE=1 XS_HESS_BLOCKS=2048" "wide_interleaved_4096:XS_HESS_INTERLEAVE=1 XS_HESS_BLOCKS=4096"; do
    name=${v%%:*}; envs=${v#*:}
    echo "== $name (round $round)"
    env $envs python profiles/tools/probe_hess.py 2>/dev/null | grep -v amdgpu.ids
    env $envs python profiles/tools/probe_gn.py 1024 2>/dev/null | grep '"n"'
  done
done
