#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection CSVs per kernel (profiling helper)."""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        cnt = collections.defaultdict(int)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void prt::", "")[:24]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(k, r["Counter_Name"])] += 1
        for k in agg:
            if k.startswith("__amd") or k.startswith("void at"): continue
            print(d.split("/")[-1], k, {c: "%.4g" % v for c, v in agg[k].items()}, "dispatches", max(cnt[(k, c)] for c in agg[k]))
