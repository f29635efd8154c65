#!/bin/bash
# HBM traffic (FETCH_SIZE, WRITE_SIZE; separate passes) of the headline frame's kernels with and without one environment switch.
#   tools/pmc_traffic_env.sh PRT_ACCUM_PER_SAMPLE=1
root="$(cd "$(dirname "$0")/.." && pwd)"
out="$root/gpurun_out/pmc_env"
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $root/bench.py --no-cpu-baseline --no-other-workloads --steps 3 --warmup 1"
for tag in default switched; do
    if [ $tag = switched ]; then export "$1"; fi
    for c in FETCH_SIZE WRITE_SIZE; do
        timeout -k 10 240 rocprofv3 --pmc $c --output-format csv -d "$out/$tag/$c" -- $BENCH > "$out/$tag.$c.log" 2>&1 || echo "pass $tag $c failed"
    done
done
python3 - "$out" "$1" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for tag in ("default", "switched"):
    tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(out + "/" + tag + "/" + c + "/*/*_counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == c:
                    k = r["Kernel_Name"].split("(")[0].replace("void prt::", "")[-64:]
                    tot[k][c] += float(r["Counter_Value"]); n[k][c] += 1
    for k in tot:
        fr = n[k]["FETCH_SIZE"] or 1
        f, w = tot[k]["FETCH_SIZE"] * 1024 / fr, tot[k]["WRITE_SIZE"] * 1024 / (n[k]["WRITE_SIZE"] or 1)
        if f + w > 5e7:
            print("%-9s %-64s per launch: fetch %.2f GB  write %.2f GB  corrected (2 x fetch + write) %.2f GB   (%d launches)" % (
                tag if tag == "default" else sys.argv[2], k, f / 1e9, w / 1e9, (2 * f + w) / 1e9, fr))
PY
