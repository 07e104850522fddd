"""ns per key of every msd kernel from the kernel_stats.csv files tools/cache_probe.py left behind: one column per run."""
import csv
import glob
import sys

runs = sys.argv[1:]
table, keys_of = {}, {}
for spec in runs:                       # directory:keys_per_rep_total
    d, nk = spec.split(":")
    f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        name = r["Name"]
        if "msd::" not in name or "gen_" in name or "check_kernel" in name:
            continue
        short = name.replace("void msd::", "").replace("msd::", "").split("(")[0]
        table.setdefault(short, {})[d] = float(r["TotalDurationNs"]) / float(nk)
print("kernel".ljust(52) + "".join(x.split(":")[0].split("/")[-1].rjust(10) for x in runs))
tot = {}
for k, v in sorted(table.items(), key=lambda kv: -max(kv[1].values())):
    print(k[:50].ljust(52) + "".join((f"{v.get(x.split(':')[0], 0):.4f}").rjust(10) for x in runs))
    for x in runs:
        tot[x] = tot.get(x, 0) + v.get(x.split(":")[0], 0)
print("sum".ljust(52) + "".join(f"{tot[x]:.4f}".rjust(10) for x in runs))
