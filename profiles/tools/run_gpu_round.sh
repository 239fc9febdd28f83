#!/bin/bash
# One GPU-box session of the round: the whole GPU test suite with durations, the default bench, a 2-rank rehearsal of the
# self-launching N > 1 path on one card (gloo).  A step that is killed at its limit ends the session (no further GPU step).
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
step() { local limit=$1 log=$2; shift 2; timeout -k 10 "$limit" "$@" > "gpurun_out/$log" 2> "gpurun_out/$log.err"; local rc=$?; echo "$log rc=$rc"; if [ $rc -ge 124 ]; then echo "killed at its limit: stopping"; exit $rc; fi; }
step 1000 r02_gpu_tests.log python -m pytest tests -m gpu -q --durations=20
tail -25 gpurun_out/r02_gpu_tests.log
step 500 r02_bench.json python bench.py
tail -c 3000 gpurun_out/r02_bench.json; tail -5 gpurun_out/r02_bench.json.err
step 400 r02_bench_n2_gloo.json python bench.py --gpus 2 --backend gloo --same-gpu --steps 20 --warmup 3 --size 256 --reloc-size 256
tail -c 2500 gpurun_out/r02_bench_n2_gloo.json; tail -5 gpurun_out/r02_bench_n2_gloo.json.err
