#!/bin/bash
# A/B of libprt_hip.so builds (and environment switches) in adaptive mode, 10..50 spp on C4, on ONE GPU box
# (AB_MODE="8 0" AB_DEPTH=8: fixed 8 spp at bounce depth 8 instead).
#   tools/ab_adaptive.sh "<variant .so>[,ENV=VALUE...]" ...      one line per entry; list an entry twice to see the noise
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
cp par_raytracer_amd/libprt_hip.so /tmp/libprt_hip.keep.so
for e in "$@"; do
    so="${e%%,*}"; envs=""
    if [ "$e" != "$so" ]; then envs="$(echo "${e#*,}" | tr ',' ' ')"; fi
    cp "$so" par_raytracer_amd/libprt_hip.so
    echo "$e: $(env $envs python tools/one_mode.py ${AB_MODE:-10 50} ${AB_FRAMES:-6} ${AB_DEPTH:-2} 2>/dev/null | tail -1)" | tee -a gpurun_out/ab_adaptive.log
done
cp /tmp/libprt_hip.keep.so par_raytracer_amd/libprt_hip.so
