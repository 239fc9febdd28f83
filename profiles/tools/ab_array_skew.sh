#!/bin/bash
# GPU box: the S2 probe with the three volume arrays (value, weight, grad: 512 MiB each) carved out of one allocation at base addresses
# a multiple of 512 MiB apart plus a skew — does the spread between processes (every word stored: 0.179 .. 0.215 ms) come from the
# arrays' same-offset accesses landing on the same channels?  Four fresh processes per skew.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
for sk in 0 256 1024 4096 65536 1048576; do for rep in 1 2 3 4; do
  echo -n "skew $sk run $rep: "
  XS_PROBE_ARRAY_SKEW=$sk timeout -k 10 120 python3 profiles/tools/probe_s2_r4.py 20 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('S2', d['S2 ms'], 'always', d['always ms'], 'first', d['first ms'], 'first always', d['first always ms'], 'exact', d['exact ms'])"
done; done
