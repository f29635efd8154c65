#!/bin/bash
# shadow rays that cannot change the image: counted, not traced (PRT_TRACE_DEAD_SHADOW_RAYS=1 traces them).  C4, pool pipeline, same box.
cd "$(dirname "$0")/.."
run() {
  name=$1; shift
  env "$@" python bench.py --no-cpu-baseline --no-other-workloads --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('%-44s %-9s %8.1f Mrays/s %7.3f ms/frame kernel %7.3f ms rays %d nodes %d' % ('$name', j['config']['pipeline'], j['value'], j['ms_per_step'], r['kernel_ms_per_frame'], r['per_frame']['rays'], r['per_frame']['node_visits']))"
}
run "both (default)" X=1
run "trace everything" PRT_TRACE_DEAD_SHADOW_RAYS=1
run "both (default)" X=1
run "trace everything" PRT_TRACE_DEAD_SHADOW_RAYS=1
