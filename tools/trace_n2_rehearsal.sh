#!/bin/bash
# The advisor's check for the N > 1 frame loop (round 2, finding 2): with CU-masked - i.e. blocking - render streams, does
# rank 0's torch work (assemble of frame k) still run beside the persistent render kernel of frame k + 1?  Two ranks on ONE
# GPU (gloo, --share-device; RCCL needs distinct devices), PRT_RESERVE_CUS=8 as bench.py sets it for RCCL runs; rank 0 runs
# under rocprofv3 --kernel-trace (started directly: no launcher between the profiler and python), rank 1 plain.
root="$(cd "$(dirname "$0")/.." && pwd)"
out="$root/gpurun_out/trace_n2"
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 WORLD_SIZE=2 LOCAL_WORLD_SIZE=2 PRT_RESERVE_CUS=8 HSA_ENABLE_IPC_MODE_LEGACY=0
ARGS="--gpus 2 --backend gloo --share-device --steps 8 --warmup 2"
RANK=1 LOCAL_RANK=1 timeout -k 10 300 python3 $root/bench.py $ARGS > "$out/rank1.log" 2>&1 &
r1=$!
RANK=0 LOCAL_RANK=0 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$out/rank0" -- python3 $root/bench.py $ARGS > "$out/rank0.log" 2>&1
wait $r1
tail -n 1 "$out/rank0.log" | cut -c1-300
python3 - "$out" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/rank0/*/*_kernel_trace.csv"):
    rows += list(csv.DictReader(open(f)))
iv = lambda r: (int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
pool = sorted(iv(r) for r in rows if "k_pool<" in r["Kernel_Name"] and r["Kernel_Name"].rstrip().endswith("false>(prt::PoolArgs const*, prt::DevCounters*)"))
torch_k = [(r["Kernel_Name"].split("(")[0][-50:], iv(r)) for r in rows if "prt::" not in r["Kernel_Name"]]
inside = [(n, a, b) for n, (a, b) in torch_k if any(a < pe and b > ps for ps, pe in pool)]
print("rank 0: %d fast k_pool launches (mean %.2f ms); %d kernels that are not the library's (torch: assemble / copies), %d of them ran while a k_pool was executing" % (
    len(pool), sum(b - a for a, b in pool) / max(1, len(pool)) / 1e6, len(torch_k), len(inside)))
for n, a, b in inside[:6]:
    ps, pe = next((ps, pe) for ps, pe in pool if a < pe and b > ps)
    print("  %-50s started %.3f ms into a k_pool of %.3f ms, ran %.3f ms" % (n, (a - ps) / 1e6, (pe - ps) / 1e6, (b - a) / 1e6))
PY
