#!/bin/bash
# Round 6 A/B, third series (GPU box, repository root, library built with EXTRAFLAGS=-DXS_EXPERIMENTS): planes per z group of the wide scan's
# tiles (XS_HESS_TILE_PLANES: G = planes / that) and workgroups (XS_HESS_BLOCKS) for the kernels with an expensive body.
# Per variant: Hessian / loss at 512^3 (scene S1), Gauss-Newton terms at 1024^3 and 512^3 (scene S1), relocalisation workload at 1024^3 (scene S3: a box room,
# walls in every orientation): frames/s, ms per pass, host side.
for v in "tp32_b1024:XS_HESS_TILE_PLANES=32 XS_HESS_BLOCKS=1024" "tp64_b1024:XS_HESS_TILE_PLANES=64 XS_HESS_BLOCKS=1024" "tp128_b1024:XS_HESS_TILE_PLANES=128 XS_HESS_BLOCKS=1024" \
         "tp128_b2048:XS_HESS_TILE_PLANES=128 XS_HESS_BLOCKS=2048" "tp128_b4096:XS_HESS_TILE_PLANES=128 XS_HESS_BLOCKS=4096" "tp256_b1024:XS_HESS_TILE_PLANES=256 XS_HESS_BLOCKS=1024" \
         "tp256_b4096:XS_HESS_TILE_PLANES=256 XS_HESS_BLOCKS=4096" "tp512_b4096:XS_HESS_TILE_PLANES=512 XS_HESS_BLOCKS=4096" "il1_tp128_b1024:XS_HESS_IL=1 XS_HESS_TILE_PLANES=128 XS_HESS_BLOCKS=1024" \
         "il4_tp128_b1024:XS_HESS_IL=4 XS_HESS_TILE_PLANES=128 XS_HESS_BLOCKS=1024" "il4_tp64_b1024:XS_HESS_IL=4 XS_HESS_TILE_PLANES=64 XS_HESS_BLOCKS=1024"; do
  name=${v%%:*}; envs=${v#*:}
  echo "== $name"
  env $envs python profiles/tools/probe_hess.py 2>/dev/null | grep hessian_ms
  env $envs python profiles/tools/probe_gn.py 1024 2>/dev/null | grep '"n"'
  env $envs python profiles/tools/probe_gn.py 512 2>/dev/null | grep '"n"'
  env $envs python bench.py --workload reloc --steps 20 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['workloads']['reloc']; print('reloc', r['value'], r['ms_per_pass_incl_allreduce_and_host_solve'], r['host_side'])"
done
