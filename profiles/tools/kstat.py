"""Average duration (us) and calls of the kernels whose name contains one of the given substrings, from rocprofv3 --stats CSVs under a directory.
usage: python3 profiles/tools/kstat.py <dir> substring [substring ...]"""
import csv, glob, os, sys
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if any(k in row["Name"] for k in sys.argv[2:]):
            print(f"  {row['Name'][:70]:70s} calls {row['Calls']:>5s}  avg {float(row['AverageNs']) / 1e3:7.2f} us  min {float(row['MinNs']) / 1e3:7.2f}  max {float(row['MaxNs']) / 1e3:7.2f}")
