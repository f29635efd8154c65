#!/bin/bash
# L1 / L2 hit rates of the headline frame's kernels (C4, pool pipeline): counters only, one group per pass.
root="$(cd "$(dirname "$0")/.." && pwd)"
out="$root/gpurun_out/pmc_cache"
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $root/bench.py --no-cpu-baseline --no-other-workloads --steps 3 --warmup 1"
pass() {
    name="$1"; shift
    echo "pmc pass $name: $*" | tee -a "$out/progress.txt"
    timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -- $BENCH > "$out/$name.log" 2>&1 || echo "pass $name failed" | tee -a "$out/failed.txt"
}
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
pass lat TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for d in ("tcp", "tcc", "lat"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(out + "/" + d + "/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0][-70:]][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in agg.items():
        if max(v.values()) > 1e6:
            print(d, k, dict((a, "%.4g" % b) for a, b in v.items()))
PY
