"""From a rocprofv3 kernel trace of the track workload: when each k_fold_count (auxiliary stream: the previous frame's voxel count, enqueued with the
header clear) starts relative to the latest raycast launch's start and end, and the gap between the integrate kernel's end and that raycast's begin —
over the whole run, in order.  usage: python3 profiles/tools/fold_offsets.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0]
last_ray = None; last_int_end = None; out = []; gaps = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = name(r)
    if n.startswith("void k_integrate_bricks"): last_int_end = e
    elif n.startswith("void k_raycast<0"):
        last_ray = (s, e)
        if last_int_end is not None: gaps.append((s - last_int_end) / 1e3)
    elif n.startswith("k_fold_count") and last_ray:
        out.append(((s - last_ray[0]) / 1e3, (s - last_ray[1]) / 1e3))
def hist(v, edges):
    return [sum(1 for x in v if a <= x < b) for a, b in zip(edges[:-1], edges[1:])]
print("fold_count start - raycast start (us), in run order, every 20th:", [round(a, 1) for a, _ in out[::20]])
print("fold_count start - raycast end   (us), in run order, every 20th:", [round(b, 1) for _, b in out[::20]])
print("integrate end -> raycast begin (us), in run order, every 20th:", [round(g, 1) for g in gaps[::20]])
