#!/bin/bash
# lane-utilisation counters (tools/util_probe.py) for several libprt_hip.so builds: tools/ab_util.sh <variant .so> ...
cd "$(dirname "$0")/.."
for v in "$@"; do
    cp "$v" par_raytracer_amd/libprt_hip.so
    echo "== $v"
    python tools/util_probe.py 2 4 2>&1 | grep -v "^\s*$"
done
