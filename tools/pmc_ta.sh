#!/bin/bash
# Texture-addresser / L1 pressure of the frame's kernels: rocprofv3 --pmc passes over bench.py (counters only, no tracing).
#   tools/pmc_ta.sh [bench args]        output: gpurun_out/pmc_ta/<pass>/... + summary.txt
root="$(cd "$(dirname "$0")/.." && pwd)"
out="$root/gpurun_out/pmc_ta"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$out/counters_available.txt" 2>&1
pass() {
    name="$1"; shift
    rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -- python3 "$root/bench.py" --no-cpu-baseline --steps 3 --warmup 1 $BENCH_ARGS > "$out/$name.log" 2>&1 || echo "pass $name failed" >> "$out/failed.txt"
}
pass a GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr
pass b TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
pass c TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum
pass d TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
pass e TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum
pass f SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
python3 "$root/tools/pmc_summary.py" "$out/a" "$out/b" "$out/c" "$out/d" "$out/e" "$out/f" > "$out/summary.txt" 2>&1
cat "$out/summary.txt"; cat "$out/failed.txt" 2>/dev/null
