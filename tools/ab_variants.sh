#!/bin/bash
# A/B of libprt_hip.so builds on the GPU box: tools/ab_variants.sh <tag> ...   (libprt_hip_<tag>.so via PRT_HIP_LIB, "default" = the product library; each runs bench on both production pipelines)
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for v in "$@"; do
    lib=libprt_hip.so; [ "$v" != default ] && lib=libprt_hip_$v.so
    for p in 4 2; do
        PRT_HIP_LIB=$lib python bench.py --no-cpu-baseline --pipeline $p --steps 8 --warmup 2 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=j['roofline']
print('%-40s %-9s %8.1f Mrays/s  %7.3f ms/frame  kernel %7.3f ms  nodes %d tris %d rays %d' % ('$v', j['config']['pipeline'], j['value'], j['ms_per_step'], r['kernel_ms_per_frame'], r['per_frame']['node_visits'], r['per_frame']['tri_tests'], r['per_frame']['rays']))"
    done
done
