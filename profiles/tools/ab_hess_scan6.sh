#!/bin/bash
# Round 6 A/B, sixth series (as ab_hess_scan4.sh): the adopted G = 4 / pairs against G = 5 / single planes, alternating, both sizes, both scenes.
for round in 1 2; do
  for v in "G4_il2:XS_HESS_GROUPS=4 XS_HESS_IL=2" "G5_il1:XS_HESS_GROUPS=5 XS_HESS_IL=1" "G3_il1:XS_HESS_GROUPS=3 XS_HESS_IL=1" "G7_il1:XS_HESS_GROUPS=7 XS_HESS_IL=1"; do
    name=${v%%:*}; envs=${v#*:}
    echo "== $name (round $round)"
    env $envs python profiles/tools/probe_hess.py 2>/dev/null | head -1
    env $envs python profiles/tools/probe_gn.py 512 2>/dev/null | head -1
    env $envs python profiles/tools/probe_gn.py 1024 2>/dev/null | head -1
    env $envs python bench.py --workload reloc --steps 20 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['workloads']['reloc']; print('reloc', r['value'], r['ms_per_pass_incl_allreduce_and_host_solve'], r['host_us_per_pass'])"
    env $envs python bench.py --workload hessian --steps 200 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hessian workload', d['value'], d['roofline']['frac'], d['roofline']['kernel_ms'])"
  done
done
