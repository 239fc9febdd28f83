"""Fold what profiles/tools/s2_modes.sh wrote into a text table: per process and regime the kernel time and the card's state, then the
address-translation counters per launch of k_integrate_bricks in dispatch order (the regimes run in the order the probe prints them)."""
import csv, glob, json, os, sys
out = sys.argv[1]
print("== kernel time per regime and process (ms: min / median / max), card state while the regime ran (min / mean / max) ==")
for f in sorted(glob.glob(os.path.join(out, "process_*.jsonl"))):
    for line in open(f):
        try:
            r = json.loads(line)
        except ValueError:
            continue
        if "tag" in r:
            print(f"\n-- {r['tag']} pid {r['pid']} arrays {r['array_addresses']} idle {r['idle']}")
            continue
        k, c = r["kernel_ms"], r["card"]
        card = "  ".join(f"{key} {c[key][0]:.0f}/{c[key][1]:.0f}/{c[key][2]:.0f}" for key in ("sclk", "dpm_sclk", "mclk", "dpm_fclk", "power", "t_junction", "t_mem") if key in c)
        print(f"{r['regime'][:52]:52s} n {r['launches']:3d}  {k['min']:.4f} / {k['median']:.4f} / {k['max']:.4f}  frac {r['frac_of_8TBs_algorithmic']:.3f}   {card}  ({c.get('samples', 0)} samples)")
print("\n== counters per launch of k_integrate_bricks (dispatch order; the probe's regimes, 12 launches each at most) ==")
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    rows = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "k_integrate_bricks" in row["Kernel_Name"]:
                rows.setdefault(int(row["Dispatch_Id"]), {})[row["Counter_Name"]] = float(row["Counter_Value"])
    names = sorted({n for v in rows.values() for n in v})
    print(f"\n-- {os.path.basename(d)}: {len(rows)} launches; columns {names}")
    for i, k in enumerate(sorted(rows)):
        print(f"   launch {i:3d}  " + "  ".join(f"{rows[k].get(n, float('nan')):14.0f}" for n in names))
