"""Per-frame kernel timeline from a rocprofv3 --kernel-trace CSV: prints the kernels of one steady-state frame
(start offset, duration, gap to the previous kernel's end) and the average idle time per frame.
usage: python profiles/tools/frame_timeline.py <kernel_trace.csv> [frame_index]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0][:40]
# a frame starts at k_scale_depth... use k_bilateral as the marker on the aux stream; main-stream frame = first k_icp after a k_resize_pyramid
starts = [i for i, r in enumerate(rows) if name(r).startswith("k_bilateral")]
fi = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) // 2
a, b = starts[fi], starts[fi + 1]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = None
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = "" if prev_end is None else f"{(s - prev_end) / 1e3:7.1f}"
    print(f"{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:6.1f}  gap {gap:>7}  q{r.get('Queue_Id', '?')}  {name(r)}")
    prev_end = max(prev_end or 0, e)
print("frame period (us):", (int(rows[b]["Start_Timestamp"]) - t0) / 1e3)
