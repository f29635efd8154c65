#!/bin/bash
# The round's profiles of the headline command (C4, pool pipeline), written to gpurun_out/prof_round/ and summarised into
# profiles/ by tools/profile_summary.py (run on the build side afterwards):
#   kernel trace + stats, HBM traffic (FETCH_SIZE and WRITE_SIZE in separate passes, as the gfx950 guide prescribes),
#   issue / stall counters, texture-addresser counters.  Counters only in the PMC passes (no tracing).
root="$(cd "$(dirname "$0")/.." && pwd)"
out="$root/gpurun_out/prof_round"
rm -rf "$out"; mkdir -p "$out"       # (gpurun MERGES this directory into the local one: delete the local copy before a new round of passes)
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $root/bench.py --no-cpu-baseline --no-other-workloads --no-live-traffic --steps 5 --warmup 1"
echo "kernel trace" | tee -a "$out/progress.txt"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- $BENCH > "$out/trace.log" 2>&1 || echo "trace failed" >> "$out/failed.txt"
pass() {
    name="$1"; shift
    echo "pmc pass $name: $*" | tee -a "$out/progress.txt"
    rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -- $BENCH > "$out/$name.log" 2>&1 || echo "pass $name failed" >> "$out/failed.txt"
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVES
pass sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
pass ta GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr
pass tcp TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum
echo "adaptive trace" | tee -a "$out/progress.txt"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_other" -- python3 $root/bench.py --no-cpu-baseline --no-live-traffic --steps 2 --warmup 1 > "$out/trace_other.log" 2>&1 || echo "trace_other failed" >> "$out/failed.txt"
echo done | tee -a "$out/progress.txt"
cat "$out/failed.txt" 2>/dev/null
