#!/bin/bash
# lane-utilisation counters (tools/util_probe.py) for several builds: tools/ab_util.sh <tag> ...   (libprt_hip_<tag>.so via PRT_HIP_LIB; "default" = the product library)
cd "$(dirname "$0")/.."
for v in "$@"; do
    lib=libprt_hip.so; [ "$v" != default ] && lib=libprt_hip_$v.so
    echo "== $v"
    PRT_HIP_LIB=$lib python tools/util_probe.py 2 4 2>&1 | grep -v "^\s*$"
done
