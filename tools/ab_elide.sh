#!/bin/bash
# dead-shadow-ray elision on / off (PRT_TRACE_DEAD_SHADOW_RAYS), C4, both production pipelines, same box
cd "$(dirname "$0")/.."
run() {
  name=$1; shift
  env "$@" python bench.py --no-cpu-baseline --no-other-workloads --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('%-40s %-9s %8.1f Mrays/s %7.3f ms/frame kernel %7.3f ms rays %d' % ('$name', j['config']['pipeline'], j['value'], j['ms_per_step'], r['kernel_ms_per_frame'], r['per_frame']['rays']))"
}
run "elide (default), pool" X=1
run "trace all, pool" PRT_TRACE_DEAD_SHADOW_RAYS=1
run "elide (default), pool" X=1
run "trace all, pool" PRT_TRACE_DEAD_SHADOW_RAYS=1
