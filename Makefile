# Top-level build.  Everything is built IN-TREE so the .so files travel to the GPU box.
#
#   make hip     par_raytracer_amd/libprt_hip.so   HIP kernels + C ABI (include/prt.h), gfx950
#   make host    par_raytracer_amd/libprt_host.so  C++ host mirror of the reference driver + prt_main
#   make oracle  oracle/libprt_oracle.so (+ oracle/_ref/ref_harness when /root/reference exists)
#
# -ffp-contract=off everywhere: parity with the reference's CPU arithmetic depends on mul and add NOT
# being fused; hipcc contracts by default.

ROOT    := $(abspath $(dir $(lastword $(MAKEFILE_LIST))))
PKG     := $(ROOT)/par_raytracer_amd
HIPCC   ?= /opt/rocm/bin/hipcc
CXX     ?= g++
ARCH    ?= gfx950

HIP_SRCS  := $(PKG)/csrc/prt_api.hip
HIP_DEPS  := $(wildcard $(PKG)/csrc/*.h) $(wildcard $(PKG)/csrc/*.hip) $(wildcard $(PKG)/csrc/*.cpp) $(ROOT)/include/prt.h $(ROOT)/include/prt_key.h
HIP_FLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math \
             -Wall -Wno-unused-function -I$(ROOT)/include

HOST_SRCS := $(PKG)/host/obj_loader.cpp $(PKG)/host/sphere_tree.cpp $(PKG)/host/host_scene.cpp \
             $(PKG)/host/image_out.cpp $(PKG)/host/image_in.cpp $(PKG)/host/image_jpeg.cpp $(PKG)/host/host_capi.cpp $(PKG)/host/render_host.cpp
HOST_DEPS := $(wildcard $(PKG)/host/*.h) $(ROOT)/include/prt.h $(ROOT)/include/prt_host.h
HOST_FLAGS := -O2 -std=c++14 -fPIC -ffp-contract=off -fno-strict-aliasing -Wall -Wno-unused-function -pthread -I$(ROOT)/include

.PHONY: all hip host oracle clean hip-experimental hip-bvh8
all: hip host oracle

hip: $(PKG)/libprt_hip.so
$(PKG)/libprt_hip.so: $(HIP_DEPS)
	$(HIPCC) $(HIP_FLAGS) $(HIP_EXTRA) -o $@ $(HIP_SRCS) $(PKG)/csrc/bvh_build.cpp

# Variant libraries (not built by default; git-ignored like every .so):
#   hip-experimental  the shipped library + the two experimental pipelines (MEGAKERNEL: the exact-association cross-check of
#                     round 1, PERSISTENT); tests/test_gpu_parity.py runs its four-pipeline comparisons when the file exists
#   hip-bvh8          the 8-wide compressed BVH of round 3 (80 B nodes, no per-step sort) instead of the 4-wide sorted one
hip-experimental: $(PKG)/libprt_hip_experimental.so
$(PKG)/libprt_hip_experimental.so: $(HIP_DEPS)
	$(HIPCC) $(HIP_FLAGS) -DPRT_EXPERIMENTAL -o $@ $(HIP_SRCS) $(PKG)/csrc/bvh_build.cpp
hip-bvh8: $(PKG)/libprt_hip_bvh8.so
$(PKG)/libprt_hip_bvh8.so: $(HIP_DEPS)
	$(HIPCC) $(HIP_FLAGS) -DPRT_BVH8 -o $@ $(HIP_SRCS) $(PKG)/csrc/bvh_build.cpp

host: $(PKG)/libprt_host.so $(PKG)/prt_main
$(PKG)/libprt_host.so: $(HOST_SRCS) $(HOST_DEPS) $(PKG)/libprt_hip.so
	$(CXX) $(HOST_FLAGS) -shared -o $@ $(HOST_SRCS) -L$(PKG) -lprt_hip -lz -Wl,-rpath,'$$ORIGIN'
$(PKG)/prt_main: $(PKG)/host/main.cpp $(PKG)/libprt_host.so
	$(CXX) $(HOST_FLAGS) -o $@ $(PKG)/host/main.cpp -L$(PKG) -lprt_host -lprt_hip -Wl,-rpath,'$$ORIGIN'

oracle:
	$(MAKE) -C $(ROOT)/oracle port ref

clean:
	rm -f $(PKG)/libprt_hip.so $(PKG)/libprt_hip_experimental.so $(PKG)/libprt_hip_bvh8.so $(PKG)/libprt_host.so $(PKG)/prt_main
	$(MAKE) -C $(ROOT)/oracle clean
