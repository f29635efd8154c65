#!/bin/bash
# A/B of several libprt_hip builds on ONE box WITHOUT touching the product library: each build is a file
# par_raytracer_amd/libprt_hip_<tag>.so selected with PRT_HIP_LIB (par_raytracer_amd/capi.py); "default" is libprt_hip.so.
#   tools/ab_libs.sh "<probe arguments>" default top21 top85 default      (probe = tools/shared_probe.py; list a build twice for the noise)
cd "$(dirname "$0")/.."
args="$1"; shift
for tag in "$@"; do
    lib=libprt_hip.so; [ "$tag" != default ] && lib=libprt_hip_$tag.so
    echo "==== $tag ($lib)"
    PRT_HIP_LIB=$lib timeout -k 10 400 python tools/shared_probe.py $args 2>&1 | grep -v amdgpu.ids
done
