#!/bin/bash
# CPU: registers / scratch / occupancy of every k_integrate_bricks instance for a set of -D flags (hipcc remarks, nothing is run).
#   profiles/tools/kernel_resources.sh "-DXS_WALK_GROUP=4 -DXS_INTEGRATE_WAVES=5" [kernel-name-substring]
cd "$(dirname "$0")/../../x-slam_amd/csrc"
F="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function"
/opt/rocm/bin/hipcc $F $1 -c xs_tsdf.hip -o /tmp/kernel_resources.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
name=None; want='${2:-k_integrate_bricks}'
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)', l)
    if m: name=m.group(1); vals={}
    m=re.search(r'remark:\s+(VGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\d+)', l)
    if m and name: vals[m.group(1).split()[0]]=int(m.group(2))
    if name and 'LDS Size' in l:
        if want in name: print(name[:60].ljust(60), vals)
        name=None
"
