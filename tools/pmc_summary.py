"""Summarise a rocprofv3 counter_collection.csv: per kernel name, mean counter values per call."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for r in rows:
    k = r["Kernel_Name"][:70]
    if pat and pat not in k: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k, d in agg.items():
    print(k)
    for n, v in sorted(d.items()): print(f"    {n:32s} {v / cnt[k][n]:16.0f}  (calls {cnt[k][n]})")
