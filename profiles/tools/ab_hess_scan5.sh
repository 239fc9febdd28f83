#!/bin/bash
# Round 6 A/B, fifth series (as ab_hess_scan4.sh): G and il at 1024^3 — Gauss-Newton terms on scene S1 and the relocalisation workload on scene S3.
for g in 4 3 5 6; do
  for il in 2 1; do
    echo "== G=$g il=$il"
    XS_HESS_GROUPS=$g XS_HESS_IL=$il python profiles/tools/probe_gn.py 1024 2>/dev/null | head -1
    XS_HESS_GROUPS=$g XS_HESS_IL=$il python bench.py --workload reloc --steps 20 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['workloads']['reloc']; print('reloc', r['value'], r['ms_per_pass_incl_allreduce_and_host_solve'], r['host_us_per_pass'])"
  done
done
