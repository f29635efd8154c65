#!/bin/bash
# Issue / stall breakdown of the frame's kernels: two rocprofv3 --pmc passes over bench.py (counters only, no tracing).
#   tools/pmc_stalls.sh [bench args]        output: gpurun_out/pmc_stalls/{a,b}/.../*_counter_collection.csv + summary.txt
root="$(cd "$(dirname "$0")/.." && pwd)"
out="$root/gpurun_out/pmc_stalls"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVES \
    --output-format csv -d "$out/a" -- python3 "$root/bench.py" --no-cpu-baseline --steps 3 --warmup 1 "$@" > "$out/a.log" 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS \
    --output-format csv -d "$out/b" -- python3 "$root/bench.py" --no-cpu-baseline --steps 3 --warmup 1 "$@" > "$out/b.log" 2>&1 || exit 1
python3 "$root/tools/pmc_summary.py" "$out/a" "$out/b" > "$out/summary.txt"
cat "$out/summary.txt"
