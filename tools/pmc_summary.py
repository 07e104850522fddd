"""Summarise rocprofv3 csv output: per-kernel stats and per-kernel counter sums."""
import csv, glob, os, sys, collections
d = sys.argv[1]
for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
    print("==", f)
    for row in csv.DictReader(open(f)):
        print(f"{row['Name'][:70]:70s} calls={row['Calls']:>6} total_ns={row['TotalDurationNs']:>12} avg_ns={float(row['AverageNs']):>12.0f} pct={row['Percentage']}")
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    print("==", f)
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:60]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    for k, v in acc.items():
        print(k)
        for c, x in sorted(v.items()):
            print(f"    {c:28s} {x:16.0f}")
