export TMPDIR=/tmp
OUT=gpurun_out/fo; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 bench.py --workload track --no-s2 --no-cpu-baseline --no-legs --steps 100 > $OUT/run.log 2>&1 || exit 1
python3 profiles/tools/fold_offsets.py $(find $OUT/t -name '*kernel_trace.csv')
rm -rf $OUT/t
