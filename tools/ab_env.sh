#!/bin/bash
# A/B of one environment switch on ONE GPU box: bench.py (C4, pool pipeline) with and without it, alternating, two rounds.
#   tools/ab_env.sh PRT_NO_XCD_SEGMENTS=1 [extra bench.py arguments]
cd "$(dirname "$0")/.."
sw="$1"; shift
run() {
    env $1 python bench.py --no-cpu-baseline --no-other-workloads --pipeline 4 --steps ${AB_STEPS:-8} --warmup 2 "${@:2}" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('%-28s %8.1f Mrays/s  %7.3f ms/frame  kernel %7.3f ms  rays %d' % ('$1', j['value'], j['ms_per_step'], r['kernel_ms_per_frame'], r['per_frame']['rays']))"
}
for i in 1 2; do run "PRT_AB_DEFAULT=1" "$@"; run "$sw" "$@"; done
