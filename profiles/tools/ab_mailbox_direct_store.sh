#!/bin/bash
# Round 6: mailbox posts as MOVDIR64B direct stores (one 64-byte write per line, no fence between payload and sequence word) against
# payload / sfence / sequence words / sfence.  GPU box, repository root; builds the library with -DXS_EXPERIMENTS and restores the product build.
set -e
grep -m1 "model name" /proc/cpuinfo; echo "movdir64b in /proc/cpuinfo flags: $(grep -c movdir64b /proc/cpuinfo) of $(grep -c ^processor /proc/cpuinfo) processors"
make -C x-slam_amd/csrc clean > /dev/null; make -C x-slam_amd/csrc EXTRAFLAGS=-DXS_EXPERIMENTS > /dev/null 2>&1
for round in 1 2 3; do
  for fenced in 1 0; do
    if [ $fenced = 1 ]; then export XS_MAILBOX_NO_DIRECT_STORE=1; else unset XS_MAILBOX_NO_DIRECT_STORE; fi
    python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-s2 --no-csfd --no-legs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
i = d['icp_us_per_iteration']; r = d['workloads']['reloc']
print('fenced stores ' if $fenced else 'direct stores ', 'round $round:', 'frames/s', d['value'], ' ICP iteration us', i['level0'], i['level1'], i['level2'], ' reloc fps', r['value'], 'host_us_per_pass', r['host_us_per_pass'])
"
  done
done
unset XS_MAILBOX_NO_DIRECT_STORE
make -C x-slam_amd/csrc clean > /dev/null; make -C x-slam_amd/csrc > /dev/null 2>&1
