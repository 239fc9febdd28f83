"""Fold the rocprofv3 counter CSVs written by collect_pmc_r5.sh into one JSON summary (round 5: as rounds 3 and 4 — the first launch of a scene — every stored
word changes — is kept apart from the later ones, where the kernel stores only the words that change).

traffic = FETCH_SIZE * cf + WRITE_SIZE * cw per launch, cf / cw = known bytes / reported bytes of the calibration kernel (one dword per lane,
three arrays read and written: MI355X_MICROARCH.md, HBM section — gfx950 FETCH_SIZE under-reports coalesced reads).  Counters are in KiB."""
import csv, glob, json, os, re, sys


def dispatches(d, kernel_substr):
    """{counter: [value per dispatch, in dispatch order]} for the kernels whose name contains kernel_substr"""
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if kernel_substr in row["Kernel_Name"]:
                rows.append((int(row["Dispatch_Id"]), row["Counter_Name"], float(row["Counter_Value"])))
    rows.sort()
    acc = {}
    for _, name, v in rows:
        acc.setdefault(name, []).append(v)
    return acc


out = sys.argv[1]
cal_known = 12.0 * (96 << 20)
res = {"calib_known_bytes_each_way": cal_known}
fac = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    k = dispatches(os.path.join(out, "calib_" + c), "k_calib_state_update").get(c)
    fac[c] = cal_known / (sum(k) / len(k) * 1024.0) if k else None
    res["calib_factor_" + c] = fac[c]


def traffic(fetch_kib, write_kib):
    r, w = fetch_kib * 1024.0 * fac["FETCH_SIZE"], write_kib * 1024.0 * fac["WRITE_SIZE"]
    return {"read_bytes": round(r), "write_bytes": round(w), "bytes_per_launch": round(r + w)}


# S2 (and its noisy variant): first launch and the steady state of the static camera
for s2tag in ("s2", "s2_noisy"):
    if not os.path.isdir(os.path.join(out, s2tag + "_FETCH_SIZE")):
        continue
    f, w = dispatches(os.path.join(out, s2tag + "_FETCH_SIZE"), "k_integrate_bricks").get("FETCH_SIZE"), dispatches(os.path.join(out, s2tag + "_WRITE_SIZE"), "k_integrate_bricks").get("WRITE_SIZE")
    us = [int(m) for m in re.findall(r"['\"]U['\"]: (\d+)", open(os.path.join(out, s2tag + "_FETCH_SIZE.log")).read())]
    if f and w and fac["FETCH_SIZE"]:
        U = us[0] if us else None
        alg = 24 * U + 2 * 640 * 480 if U else None
        e = {"kernel": "k_integrate_bricks<false, true>, scene S2 512^3 (profiles/tools/probe_s2_pmc.py)" + (" with 2 mm noise, holes and 0.2 % speckle" if s2tag == "s2_noisy" else ""), "U": U, "algorithmic_bytes_per_launch": alg,
             "first_launch_into_an_empty_volume": traffic(f[0], w[0]),
             "later_launches_same_frame": dict(traffic(sum(f[1:]) / len(f[1:]), sum(w[1:]) / len(w[1:])), launches_averaged=len(f) - 1)}
        for k in ("first_launch_into_an_empty_volume", "later_launches_same_frame"):
            if alg:
                e[k]["ratio_to_algorithmic"] = round(e[k]["bytes_per_launch"] / alg, 3)
        # bench.py's traffic_from_profile reads these two
        e["traffic_bytes_per_launch"] = e["later_launches_same_frame"]["bytes_per_launch"]
        # the first touch as an entry of its own (bench.py: roofline_s2.first_touch.traffic_from_profile)
        res[s2tag + "_first_touch"] = {"kernel": e["kernel"] + ", launch into a freshly initialised volume", "U": U, "algorithmic_bytes_per_launch": alg,
                                 "traffic_bytes_per_launch": e["first_launch_into_an_empty_volume"]["bytes_per_launch"],
                                 "read_bytes": e["first_launch_into_an_empty_volume"]["read_bytes"], "write_bytes": e["first_launch_into_an_empty_volume"]["write_bytes"]}
        sq = dispatches(os.path.join(out, s2tag + "_SQ"), "k_integrate_bricks")
        e["sq_counters_later_launches"] = {k: round(sum(v[1:]) / len(v[1:])) for k, v in sorted(sq.items())}
        res[s2tag] = e
# the pipeline: every launch of the tracked stream
for tag, name in (("s1", "k_integrate_bricks<false"), ("s1_bilinear", "k_integrate_bricks<true")):
    f = dispatches(os.path.join(out, "track_FETCH_SIZE"), name).get("FETCH_SIZE")
    w = dispatches(os.path.join(out, "track_WRITE_SIZE"), name).get("WRITE_SIZE")
    if not (f and w and fac["FETCH_SIZE"]):
        continue
    try:
        line = [l for l in open(os.path.join(out, "track_FETCH_SIZE.log")) if l.startswith("{")][-1]
        b = json.loads(line)
        U = b["roofline"]["U_per_frame"] if tag == "s1" else b["bilinear"]["U_per_frame"]
    except Exception:
        U = None
    n = min(len(f), len(w))
    e = {"kernel": name + ", .> inside the pipeline (bench.py --workload track: scene S1 512^3, a new pose every frame)", "launches_averaged": n,
         "U": U, "algorithmic_bytes_per_launch": round(24 * U + 2 * 640 * 480) if U else None}
    e.update(traffic(sum(f[:n]) / n, sum(w[:n]) / n))
    e["traffic_bytes_per_launch"] = e["bytes_per_launch"]
    if U:
        e["ratio_to_algorithmic"] = round(e["bytes_per_launch"] / e["algorithmic_bytes_per_launch"], 3)
    res[tag] = e
print(json.dumps(res, indent=1))
