#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py into profiles/traffic_<workload>.json.

usage: make_traffic.py <fetch_dir> <write_dir> <workload> <out.json> [frames]
FETCH_SIZE / WRITE_SIZE are KiB.  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reads 1/2 of a wide coalesced
stream and is uncalibrated for other widths, so raw = FETCH + WRITE is a lower bound and corrected = 2*FETCH + WRITE
an upper bound of the HBM bytes.
"""
import collections, csv, glob, json, sys

def per_kernel(d, counter):
    f = glob.glob(d + "/*/*_counter_collection.csv")[0]
    agg = collections.defaultdict(float); n = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter: continue
        k = r["Kernel_Name"].split("(")[0].replace("void prt::", "").replace("prt::", "")
        agg[k] += float(r["Counter_Value"]); n[k] += 1
    return agg, n

fetch_dir, write_dir, workload, out_path = sys.argv[1:5]
fetch, nf = per_kernel(fetch_dir, "FETCH_SIZE")
write, _ = per_kernel(write_dir, "WRITE_SIZE")
frames = int(sys.argv[5]) if len(sys.argv) > 5 else max(nf.get("k_resolve", 1), 1)
out = {"command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline",
       "workload": workload, "frames_in_profile": frames,
       "unit_note": "FETCH_SIZE / WRITE_SIZE are KiB; raw = FETCH + WRITE (lower bound), corrected = 2*FETCH + WRITE (gfx950 FETCH_SIZE halving, upper bound)",
       "kernels": {}}
for k in sorted(fetch):
    if not k.startswith("k_"): continue
    # kernels of the counting frame (COUNT = true instantiations) run once; everything else every frame
    fr = 1 if k.endswith("true>") else (frames - 1 if "k_trace<256, false>" in k or "k_trace_overflow<false>" in k else frames)
    fr = max(fr, 1)
    out["kernels"][k] = {"fetch_KiB_per_frame": fetch[k] / fr, "write_KiB_per_frame": write.get(k, 0.0) / fr,
                         "hbm_bytes_per_frame_raw": int((fetch[k] + write.get(k, 0.0)) / fr * 1024),
                         "hbm_bytes_per_frame_corrected": int((2 * fetch[k] + write.get(k, 0.0)) / fr * 1024)}
kt = out["kernels"].get("k_trace<256, false>")
if kt:
    out["hbm_bytes_per_frame_k_trace"] = kt["hbm_bytes_per_frame_corrected"]
for k, v in out["kernels"].items():
    if k.startswith("k_pool<") and ", false, false, false>" in k and "true, false, false, false>" not in k.replace("k_pool<256, 5, false, true,", ""):
        out["hbm_bytes_per_frame_k_pool"] = v["hbm_bytes_per_frame_corrected"]
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps({k: (v["hbm_bytes_per_frame_raw"], v["hbm_bytes_per_frame_corrected"]) for k, v in out["kernels"].items()}, indent=1))
