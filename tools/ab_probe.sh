#!/bin/bash
# What does one more instruction per BVH node step cost the frame?  Builds libprt_hip.so with the sensitivity probes of
# dev_trace.h (trav_node_step) - PRT_PROBE_EXTRA_LOAD: a fifth divergent vector-memory instruction per step;
# PRT_PROBE_EXTRA_VALU=n: n more vector ALU instructions per step - and times them against the plain build in ONE gpurun call.
#   tools/ab_probe.sh build      (on the build side; writes variants/*.so)
#   tools/ab_probe.sh run        (on the GPU box: AB_STEPS=8 tools/ab_pool.sh over the variants, twice)
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -Wno-unused-function -Iinclude"
S="par_raytracer_amd/csrc/prt_api.hip par_raytracer_amd/csrc/bvh_build.cpp"
if [ "$1" = build ]; then
    mkdir -p variants
    hipcc $F -o variants/base.so $S &
    hipcc $F -DPRT_PROBE_EXTRA_LOAD -o variants/load5.so $S &
    hipcc $F -DPRT_PROBE_EXTRA_VALU=8 -o variants/valu8.so $S &
    hipcc $F -DPRT_PROBE_EXTRA_VALU=16 -o variants/valu16.so $S &
    wait
else
    cp par_raytracer_amd/libprt_hip.so /tmp/libprt_hip.keep.so
    V="variants/base.so variants/load5.so variants/valu8.so variants/valu16.so"
    AB_STEPS=8 bash tools/ab_pool.sh $V $V
    cp /tmp/libprt_hip.keep.so par_raytracer_amd/libprt_hip.so
fi
