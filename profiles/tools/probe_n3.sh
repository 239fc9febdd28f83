#!/bin/bash
# The N^3-bound kernels at the final tree (run on the GPU box from the repository root): Hessian / loss probe at 512^3, the
# Gauss-Newton pass at 512^3 and 1024^3 (scene S1: a wall across z), and the two N^3 workloads of the bench (reloc on the box room).
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
python profiles/tools/probe_hess.py 2>/dev/null | grep "^{"
python profiles/tools/probe_gn.py 512 2>/dev/null | grep "^{"
python profiles/tools/probe_gn.py 1024 2>/dev/null | grep "^{"
python bench.py --workload hessian --steps 200 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hessian workload', d['value'], d['roofline'])"
python bench.py --workload reloc --steps 20 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); w=d['workloads']['reloc']; print('reloc workload', w['value'], w['ms_per_pass_incl_allreduce_and_host_solve'], w['gt_read_GBs_per_rank'], w['loss_first_to_last'], w['position_error_mm_start_to_end'])"
