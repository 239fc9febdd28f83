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
cal_known = 12.0 * (96 << 20)
res = {"calib_known_bytes_each_way": cal_known}
fac = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    k, _ = counters(os.path.join(out, "calib_" + c), "k_calib_state_update")
    res["calib_" + c + "_KiB"] = k.get(c)
    fac[c] = cal_known / (k[c] * 1024.0) if k.get(c) else None
    res["calib_factor_" + c] = fac[c]
for scene, what in (("s2", "scene S2 512^3 (profiles/tools/probe_s2.py)"), ("s1", "scene S1 512^3 frame 20 (profiles/tools/probe_s1.py), brick-list launches")):
    e = {"kernel": "k_integrate_bricks<false>, " + what + ", per launch"}
    cs, n = {}, {}
    for sub in ("FETCH_SIZE", "WRITE_SIZE", "SQ"):
        d = os.path.join(out, scene + "_" + sub)
        if os.path.isdir(d):
            a, cnt = counters(d, "k_integrate_bricks"); cs.update(a); n.update(cnt)
    e["counters"] = cs
    # U (voxels written per launch) of the very runs the counters come from: the probes print it
    import re
    us = []
    for sub in ("FETCH_SIZE", "WRITE_SIZE"):
        try:
            us += [int(m) for m in re.findall(r"['\"]U['\"]: (\d+)", open(os.path.join(out, scene + "_" + sub + ".log")).read())]
        except OSError:
            pass
    e["U"] = us[0] if us and all(u == us[0] for u in us) else (us or None)
    if isinstance(e["U"], int):
        e["algorithmic_bytes_per_launch"] = 24 * e["U"] + 2 * 640 * 480
    e["launches_averaged"] = n
    if cs.get("FETCH_SIZE") and cs.get("WRITE_SIZE") and fac["FETCH_SIZE"]:
        e["traffic_read_bytes"] = cs["FETCH_SIZE"] * 1024.0 * fac["FETCH_SIZE"]
        e["traffic_write_bytes"] = cs["WRITE_SIZE"] * 1024.0 * fac["WRITE_SIZE"]
        e["traffic_bytes_per_launch"] = e["traffic_read_bytes"] + e["traffic_write_bytes"]
    res[scene] = e
print(json.dumps(res, indent=1))
