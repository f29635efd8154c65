#!/bin/bash
# Issue counters of the frame's kernel for several library builds, one box: tools/pmc_sq.sh <tag> ...
# (a build is par_raytracer_amd/libprt_hip_<tag>.so, selected with PRT_HIP_LIB; "default" = libprt_hip.so, which is never touched)
# (two rocprofv3 --pmc passes of bench.py per build, counters only; summary: vector instructions per frame, vector pipe busy,
# share of wave cycles spent waiting)
root="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
    tag="$v"
    if [ "$v" = default ]; then export PRT_HIP_LIB=libprt_hip.so; else export PRT_HIP_LIB=libprt_hip_$v.so; fi
    out="$root/gpurun_out/pmc_sq_$tag"; rm -rf "$out"; mkdir -p "$out"
    BENCH="python3 $root/bench.py --no-cpu-baseline --no-other-workloads --pipeline 4 --steps 3 --warmup 1"
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d "$out/sq1" -- $BENCH > "$out/sq1.log" 2>&1
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS GRBM_GUI_ACTIVE TA_TA_BUSY_sum --output-format csv -d "$out/sq2" -- $BENCH > "$out/sq2.log" 2>&1
    python3 - "$out" "$tag" <<'PY'
import csv, glob, sys, collections
out, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out + "/sq*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_pool<" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", 0))[:1]:
    c = {a: agg[k][a] / max(1, n[k][a]) for a in agg[k]}
    print("%-14s VALU insts %.3f G  SALU %.3f G  VMEM rd %.3f G  LDS %.3f G | wave cycles: %.1f%% waiting, %.1f%% issuing vector; vector pipe busy x5 waves %.1f%%; TA busy %.1f%%" % (
        tag, c.get("SQ_INSTS_VALU", 0) / 1e9, c.get("SQ_INSTS_SALU", 0) / 1e9, c.get("SQ_INSTS_VMEM_RD", 0) / 1e9, c.get("SQ_INSTS_LDS", 0) / 1e9,
        100 * c.get("SQ_WAIT_ANY", 0) / max(1, c.get("SQ_WAVE_CYCLES", 1)), 100 * c.get("SQ_ACTIVE_INST_VALU", 0) / max(1, c.get("SQ_WAVE_CYCLES", 1)),
        500 * c.get("SQ_ACTIVE_INST_VALU", 0) / max(1, c.get("SQ_WAVE_CYCLES", 1)),
        100 * c.get("TA_TA_BUSY_sum", 0) / 256.0 / max(1, c.get("GRBM_GUI_ACTIVE", 1) / 8.0)))
PY
done
