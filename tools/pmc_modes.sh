#!/bin/bash
# Instruction and wait counters of k_pool in adaptive mode (10..50 spp) and at fixed 50 spp, C4: where does adaptive mode's
# time per ray go?  Counters only, one group per pass.
root="$(cd "$(dirname "$0")/.." && pwd)"
out="$root/gpurun_out/pmc_modes"
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for mode in "10 50" "50 0"; do
    tag="m$(echo $mode | tr ' ' '_')"
    for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum"; do
        g="$(echo $grp | cut -d' ' -f1)"
        timeout -k 10 240 rocprofv3 --pmc $grp --output-format csv -d "$out/$tag/$g" -- python3 $root/tools/one_mode.py $mode 2 > "$out/$tag.$g.log" 2>&1 || echo "pass $tag $g failed"
        tail -1 "$out/$tag.$g.log"
    done
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for tag in ("m10_50", "m50_0"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
    for f in glob.glob(out + "/" + tag + "/*/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-60:]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in agg.items():
        if v.get("SQ_INSTS_VALU", 0) > 1e9:
            print(tag, k, " ".join("%s=%.4g" % (a, b / 2) for a, b in sorted(v.items())))
PY
