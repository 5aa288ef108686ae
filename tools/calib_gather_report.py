"""Per-dispatch counters of pf_calib_gather_kernel from rocprofv3 --pmc output directories, in dispatch order."""
import collections, csv, glob, os, sys
tab = collections.defaultdict(dict)
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "pf_calib_gather_kernel" in r["Kernel_Name"]:
                per[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        for c, v in per.items():
            for i, (_, x) in enumerate(sorted(v)):
                tab[i][c] = x
for i in sorted(tab):
    print(f"launch {i:2d}: " + "  ".join(f"{c} {v:.0f}" for c, v in sorted(tab[i].items())))
