"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel (mean per dispatch)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2:] or None
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"].split("(")[0]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if flt and not any(f in k for f in flt):
        continue
    print(k)
    for c, vals in sorted(v.items()):
        print("   %-24s mean %.5g  n=%d" % (c, sum(vals) / len(vals), len(vals)))
