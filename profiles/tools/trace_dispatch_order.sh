export TMPDIR=/tmp
OUT=gpurun_out/fo; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 bench.py --workload track --no-s2 --no-cpu-baseline --no-legs --steps 100 > $OUT/run.log 2>&1 || exit 1
python3 - $(find $OUT/t -name '*kernel_trace.csv') <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(list(rows[0].keys()))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0][:30]
idx = [i for i, r in enumerate(rows) if name(r).startswith("k_bilateral")]
a, b = idx[150], idx[151]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a - 4:b + 2]:
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:6.1f}  q{r['Queue_Id']} stream {r.get('Stream_Id', '?')} thread {r.get('Thread_Id', '?')} dispatch {r.get('Dispatch_Id', '?')} corr {r.get('Correlation_Id', '?')}  {name(r)}")
PY
rm -rf $OUT/t
