#!/bin/bash
# (HISTORICAL: bench.py's --no-numa-bind / XS_BENCH_BIND_REMOTE were removed after this measurement: profiles/r06_ab_numa_bind.txt)
# Round 6: bench.py binds the rank's main thread to the CPUs of its GPU's NUMA node (default) or leaves it to the scheduler (--no-numa-bind).
# `remote` = bound to the OTHER socket on purpose (XS_BENCH_BIND_REMOTE=1): what the binding protects from.
# GPU box, repository root, product library; alternating, 200 frames each + the hessian / reloc workloads' short runs.
for round in 1 2 3; do
  for flag in --no-numa-bind "" remote; do
    if [ "$flag" = remote ]; then export XS_BENCH_BIND_REMOTE=1; f=""; else unset XS_BENCH_BIND_REMOTE; f=$flag; fi
    python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-s2 --no-csfd --no-legs $f 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
i = d['icp_us_per_iteration']; r = d['workloads']['reloc']; h = d['host_affinity']
print(({'--no-numa-bind': 'scheduler   ', '': 'bound       ', 'remote': 'other socket'})['$flag'], 'round $round:', 'frames/s', d['value'], ' ICP iteration us', i['level0'], i['level1'], i['level2'], ' reloc fps', r['value'], 'host_us_per_pass', r['host_us_per_pass'],
      ' gpu node', h.get('gpu_numa_node'), 'bound', h.get('bound'), 'cpu at start', h.get('thread_started_on_cpu'))
"
  done
done
