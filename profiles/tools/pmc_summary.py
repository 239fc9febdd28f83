"""Fold the rocprofv3 counter CSVs written by collect_pmc.sh into one JSON summary.

traffic = FETCH_SIZE * cf + WRITE_SIZE * cw per launch of the integrate kernel, where cf / cw are
the known-bytes / reported-bytes ratios of the calibration kernel (same one-dword-per-lane access
pattern; MI355X_MICROARCH.md, HBM section: gfx950 FETCH_SIZE under-reports coalesced reads and
other widths must be calibrated).  FETCH_SIZE / WRITE_SIZE are in KiB.
"""
import csv, glob, json, os, sys

def counters(d, kernel_substr):
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if kernel_substr not in row["Kernel_Name"]:
                continue
            acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}

out = sys.argv[1]
res = {"kernel": "k_integrate_bricks<false>, scene S2 512^3 (scratch/probe_s2.py), per launch"}
cal_known = 12.0 * (96 << 20)
s2, n = {}, {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    a, cnt = counters(os.path.join(out, "s2_" + c), "k_integrate_bricks"); s2.update(a); n.update(cnt)
    k, _ = counters(os.path.join(out, "calib_" + c), "k_calib_state_update")
    res["calib_" + c + "_KiB"] = k.get(c)
    res["calib_factor_" + c] = cal_known / (k[c] * 1024.0) if k.get(c) else None
a, cnt = counters(os.path.join(out, "s2_SQ"), "k_integrate_bricks"); s2.update(a); n.update(cnt)
res["counters"] = s2
res["launches_averaged"] = n
if s2.get("FETCH_SIZE") and s2.get("WRITE_SIZE") and res["calib_factor_FETCH_SIZE"]:
    res["traffic_read_bytes"] = s2["FETCH_SIZE"] * 1024.0 * res["calib_factor_FETCH_SIZE"]
    res["traffic_write_bytes"] = s2["WRITE_SIZE"] * 1024.0 * res["calib_factor_WRITE_SIZE"]
    res["traffic_bytes_per_launch"] = res["traffic_read_bytes"] + res["traffic_write_bytes"]
res["calib_known_bytes_each_way"] = cal_known
print(json.dumps(res, indent=1))
