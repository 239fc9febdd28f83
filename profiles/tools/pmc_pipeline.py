"""Per-kernel HBM read traffic of the whole pipeline from one rocprofv3 --pmc FETCH_SIZE pass over bench.py.
usage (GPU box):  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT -- python3 bench.py --steps 30 --no-s2 --no-cpu-baseline
                  python3 profiles/tools/pmc_pipeline.py OUT
FETCH_SIZE is in KiB and under-reports by 2 on gfx950 for these dword-per-lane streams (profiles/r01_integrate_pmc_v3.json)."""
import csv, glob, json, os, sys
acc = {}
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] != "FETCH_SIZE":
            continue
        name = row["Kernel_Name"].split("(")[0]
        key = (name, row.get("Grid_Size", "?"))
        acc.setdefault(key, []).append(float(row["Counter_Value"]))
out = []
for (name, grid), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    out.append({"kernel": name, "grid_threads": grid, "launches": len(v), "read_MB_per_launch_x2_corrected": round(sum(v) / len(v) * 1024 * 2 / 1e6, 2)})
print(json.dumps(out[:24], indent=1))
