#!/bin/bash
# Round 6: pieces per round (XS_CSFD_UNROLL 1 / 2 / 4) and workgroups (XS_CSFD_BLOCKS) of the CSFD array kernels on 64 M elements (1.5 GB per launch).
# GPU box, repository root, library built with EXTRAFLAGS=-DXS_EXPERIMENTS (the script builds it and restores the product build).
set -e
make -C x-slam_amd/csrc clean > /dev/null; make -C x-slam_amd/csrc EXTRAFLAGS=-DXS_EXPERIMENTS > /dev/null 2>&1
cat > /tmp/probe_csfd.py <<'PY'
import importlib, sys
sys.path.insert(0, '.')
import torch
capi = importlib.import_module('x-slam_amd.capi')
s = torch.cuda.current_stream()
out = {}
for nb in (1000000, 64 << 20):
    ba = torch.empty((nb, 2), dtype=torch.float32, device="cuda").uniform_(-2, 2); bb = torch.empty((nb, 2), dtype=torch.float32, device="cuda").uniform_(0.05, 2)
    ba[:, 1] = 1e-6; bb[:, 1] = 1e-6
    bo = torch.empty_like(ba)
    row = {}
    for name in ("mul", "div", "exp", "sin"):
        reps = 50 if nb < (1 << 22) else 8
        capi.csfd_array_op(name, "our", ba, bb, bo, nb, stream=s)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps):
            capi.csfd_array_op(name, "our", ba, bb, bo, nb, stream=s)
        e1.record(s); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        row[name] = (round(ms, 5), round(24.0 * nb / ms / 1e6, 1))
    print(nb, row, flush=True)
PY
for round in 1 2; do
  for cfg in ${CSFD_AB_CONFIGS:-1,2048 2,2048 4,2048 1,4096 2,4096 4,4096 2,1020 4,1020 4,510 2,8192}; do   # unroll,workgroups
    set -- ${cfg/,/ }
    echo "== unroll $1 blocks $2 (round $round)"
    XS_CSFD_UNROLL=$1 XS_CSFD_BLOCKS=$2 python /tmp/probe_csfd.py
  done
done
make -C x-slam_amd/csrc clean > /dev/null; make -C x-slam_amd/csrc > /dev/null 2>&1
