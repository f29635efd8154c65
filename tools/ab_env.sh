#!/bin/bash
# A/B of an environment knob on the GPU box: tools/ab_env.sh VAR value1 value2 ...   (bench on both production pipelines; C4, then C3 on the pool pipeline)
cd "$(dirname "$0")/.."
var=$1; shift
for v in "$@"; do
    for wl in "C4 4" "C4 2" "C3 4"; do
        set -- $wl
        env $var=$v python bench.py --no-cpu-baseline --workload $1 --pipeline $2 --steps 8 --warmup 2 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=j['roofline']
print('%-28s %-3s %-9s %8.1f Mrays/s  %7.3f ms/frame  kernel %7.3f ms  nodes %d tris %d rays %d' % ('$var=$v', '$1', j['config']['pipeline'], j['value'], j['ms_per_step'], r['kernel_ms_per_frame'], r['per_frame']['node_visits'], r['per_frame']['tri_tests'], r['per_frame']['rays']))"
    done
done
