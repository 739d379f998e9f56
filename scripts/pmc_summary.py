"""Sums rocprofv3 counter_collection.csv per kernel: python scripts/pmc_summary.py <csv> [substring ...]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
want = sys.argv[2:] or [""]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in rows:
    k = r["Kernel_Name"]
    if any(w in k for w in want):
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
for k, v in agg.items():
    n = len(disp[k])
    print(k[:90], "dispatches", n)
    for c, val in sorted(v.items()):
        print(f"   {c:32s} {val / n:16.1f} per dispatch")
