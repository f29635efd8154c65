#!/usr/bin/env python3
"""Print the per-kernel timeline of one frame from a rocprofv3 --kernel-trace CSV (profiling helper)."""
import csv, glob, sys
d = sys.argv[1]
frame = int(sys.argv[2]) if len(sys.argv) > 2 else 1
f = glob.glob(d + "/*/*_kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a frame ends with its k_resolve; it starts right after the previous one
ends = [i for i, r in enumerate(rows) if "k_resolve" in r["Kernel_Name"]]
a = ends[frame - 1] + 1 if frame > 0 else 0
b = ends[frame] + 1
t0 = int(rows[a]["Start_Timestamp"])
tot = {}
for r in rows[a:b]:
    n = r["Kernel_Name"].split("(")[0].replace("void prt::", "")[:28]
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    tot[n] = tot.get(n, 0) + dur
    print("%-30s start %8.3f ms dur %8.3f ms grid %s vgpr %s lds %s" % (n, (int(r["Start_Timestamp"]) - t0) / 1e6, dur, r.get("Grid_Size_X"), r.get("VGPR_Count"), r.get("LDS_Block_Size")))
print({k: round(v, 3) for k, v in tot.items()})
