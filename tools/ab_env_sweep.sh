#!/bin/bash
# One library build, one box: bench.py (C4, pool pipeline) under several settings of one environment knob.
#   tools/ab_env_sweep.sh <tag> <ENV_NAME> <value> ...        ("-" = unset; the build is par_raytracer_amd/libprt_hip_<tag>.so,
#   selected with PRT_HIP_LIB - "default" = libprt_hip.so, which is never overwritten)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
if [ "$1" = default ]; then export PRT_HIP_LIB=libprt_hip.so; else export PRT_HIP_LIB=libprt_hip_$1.so; fi
name=$2; shift 2
for val in "$@"; do
    if [ "$val" = "-" ]; then unset $name; else export $name=$val; fi
    python bench.py --no-cpu-baseline --no-other-workloads --pipeline 4 --steps ${AB_STEPS:-8} --warmup 2 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=j['roofline']
print('%-20s %8.1f Mrays/s  %7.3f ms/frame  kernel %7.3f ms' % ('$name=$val', j['value'], j['ms_per_step'], r['kernel_ms_per_frame']))" | tee -a gpurun_out/ab_env_sweep.log
done
