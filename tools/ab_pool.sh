#!/bin/bash
# A/B of libprt_hip.so builds on ONE GPU box (box-to-box differences are larger than most effects): pool pipeline, C4.
#   tools/ab_pool.sh <tag> ...      one line per build (par_raytracer_amd/libprt_hip_<tag>.so via PRT_HIP_LIB; "default" = the product
#   library, never overwritten); list a build twice to see the run-to-run noise
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for v in "$@"; do
    lib=libprt_hip.so; [ "$v" != default ] && lib=libprt_hip_$v.so
    PRT_HIP_LIB=$lib python bench.py --no-cpu-baseline --no-other-workloads --pipeline 4 --steps ${AB_STEPS:-10} --warmup 2 ${AB_ARGS} 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=j['roofline']
print('%-32s %8.1f Mrays/s  %7.3f ms/frame  kernel %7.3f ms  nodes %d tris %d rays %d' % ('$v', j['value'], j['ms_per_step'], r['kernel_ms_per_frame'], r['per_frame']['node_visits'], r['per_frame']['tri_tests'], r['per_frame']['rays']))" | tee -a gpurun_out/ab_pool.log
done
