"""ctypes view of the C ABI (include/prt.h, include/prt_host.h).

Python is the test / bench harness language of this repository; the product is the two shared
libraries.  Nothing here computes: every call goes straight through the C ABI, and a missing
library is a hard error (there is no CPU or PyTorch fallback for the hot path).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(PKG_DIR)

c_float3 = C.c_float * 3
c_float4 = C.c_float * 4


class PrtMaterial(C.Structure):
    _fields_ = [("specular_intensity", C.c_float), ("index_of_refraction", C.c_float), ("alpha", C.c_float),
                ("ambient_color", c_float4), ("diffuse_color", c_float4), ("specular_color", c_float4),
                ("ambient_texture", C.c_int32), ("diffuse_texture", C.c_int32), ("specular_texture", C.c_int32),
                ("alpha_texture", C.c_int32), ("bump_texture", C.c_int32)]


class PrtLight(C.Structure):
    _fields_ = [("type", C.c_int32), ("color", c_float4), ("position", c_float3), ("facing", c_float3),
                ("falloff", C.c_float)]


class PrtGroup(C.Structure):
    _fields_ = [("first_index", C.c_uint32), ("index_count", C.c_uint32), ("material", C.c_int32)]


class PrtBSphere(C.Structure):
    _fields_ = [("center", c_float3), ("radius", C.c_float), ("c0", C.c_uint32), ("c1", C.c_uint32)]


class PrtTexture(C.Structure):
    _fields_ = [("size_x", C.c_uint32), ("size_y", C.c_uint32), ("channels", C.c_uint32),
                ("texels", C.POINTER(C.c_uint8))]


class PrtSceneDesc(C.Structure):
    _fields_ = [("positions", C.POINTER(C.c_float)), ("position_count", C.c_uint32),
                ("normals", C.POINTER(C.c_float)), ("normal_count", C.c_uint32),
                ("texcoords", C.POINTER(C.c_float)), ("texcoord_count", C.c_uint32),
                ("tangents", C.POINTER(C.c_float)),
                ("idx_positions", C.POINTER(C.c_uint32)), ("idx_texcoords", C.POINTER(C.c_uint32)),
                ("idx_normals", C.POINTER(C.c_uint32)), ("index_count", C.c_uint32),
                ("groups", C.POINTER(PrtGroup)), ("group_count", C.c_uint32),
                ("materials", C.POINTER(PrtMaterial)), ("material_count", C.c_uint32),
                ("textures", C.POINTER(PrtTexture)), ("texture_count", C.c_uint32),
                ("lights", C.POINTER(PrtLight)), ("light_count", C.c_uint32),
                ("spheres", C.POINTER(PrtBSphere)), ("sphere_group", C.POINTER(C.c_int32)),
                ("sphere_count", C.c_uint32)]


class PrtCamera(C.Structure):
    _fields_ = [("tan_a2", C.c_float), ("aspect", C.c_float), ("inv_width", C.c_float), ("inv_height", C.c_float),
                ("position", c_float3), ("forward", c_float3), ("right", c_float3), ("up", c_float3)]


class PrtParams(C.Structure):
    _fields_ = [("ray_bias", C.c_float), ("reflection_samples", C.c_uint32), ("spec_samples", C.c_uint32),
                ("bounce_depth", C.c_uint32), ("background_color", c_float4), ("spp", C.c_uint32),
                ("pipeline", C.c_uint32), ("seed", C.c_uint64), ("max_spp", C.c_uint32), ("variance_threshold", C.c_float)]


class PrtCounters(C.Structure):
    _fields_ = [("ray_count", C.c_uint64), ("node_visits", C.c_uint64), ("tri_tests", C.c_uint64),
                ("shaded_hits", C.c_uint64), ("render_ms", C.c_double), ("trace_kernel_ms", C.c_double),
                ("trace_kernel_launches", C.c_uint32), ("pipeline", C.c_uint32)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class PrtSceneInfo(C.Structure):
    _fields_ = [("triangle_count", C.c_uint32), ("bvh_node_count", C.c_uint32), ("bvh_max_depth", C.c_uint32),
                ("bvh_node_bytes", C.c_uint32), ("tri_record_bytes", C.c_uint32), ("shade_record_bytes", C.c_uint32),
                ("device_bytes", C.c_uint64), ("bvh_build_ms", C.c_double)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class PrtRenderStats(C.Structure):
    _fields_ = [("node_visits", C.c_uint64), ("tri_tests", C.c_uint64), ("wave_node_steps", C.c_uint64),
                ("wave_tri_steps", C.c_uint64), ("wave_leaf_visits", C.c_uint64), ("wave_refills", C.c_uint64),
                ("deepest_stack", C.c_uint64), ("phase_cycles", C.c_uint64 * 5), ("parked_rays", C.c_uint64),
                ("parked_shadow_rays", C.c_uint64), ("elided_shadow_rays", C.c_uint64), ("variance_close_calls", C.c_uint64), ("stack_lds_entries", C.c_uint32), ("stack_bound", C.c_uint32)]


PIPELINE_DEFAULT, PIPELINE_MEGAKERNEL, PIPELINE_WAVEFRONT, PIPELINE_PERSISTENT, PIPELINE_POOL = 0, 1, 2, 3, 4
FLAG_COUNT_VISITS = 0x100
FLAG_TRYOUT = 0x200
BUILD_EXPERIMENTAL, BUILD_BVH4 = 1, 2

# Every symbol include/prt.h declares; tests/test_capi_symbols.py checks the library exports them all.
PRT_SYMBOLS = ["prt_create", "prt_destroy", "prt_last_error", "prt_abi_version", "prt_set_option", "prt_build_flags", "prt_upload_scene", "prt_render",
               "prt_render_device", "prt_shard_rows", "prt_render_shard_device", "prt_render_shard", "prt_render_pixel_list", "prt_get_scene_info", "prt_get_render_stats", "prt_debug_check_bvh", "prt_debug_check_bvh_lbvh", "prt_debug_device_kat",
               "prt_multi_create", "prt_multi_destroy", "prt_multi_last_error", "prt_multi_device_count", "prt_multi_context",
               "prt_multi_upload_scene", "prt_multi_render", "prt_multi_depth", "prt_multi_submit", "prt_multi_wait", "prt_debug_throw"]
PRT_HOST_SYMBOLS = ["prt_host_load_obj", "prt_host_free_scene", "prt_host_scene_desc", "prt_host_scene_hierarchy_seconds",
                    "prt_host_scene_parse_seconds", "prt_host_last_error", "prt_host_make_camera",
                    "prt_host_default_params", "prt_host_render", "prt_host_render_error", "prt_host_write_image", "prt_host_tonemap",
                    "prt_host_load_texture", "prt_host_free_texture", "prt_host_debug_throw"]


class LibraryMissing(RuntimeError):
    pass


def _load(path: str, what: str) -> C.CDLL:
    if not os.path.exists(path):
        raise LibraryMissing("%s not built: %s is missing.  Run `make` (or __graft_entry__.build()) at the repo root; "
                             "there is no fallback implementation." % (what, path))
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


_hip: Optional[C.CDLL] = None
_host: Optional[C.CDLL] = None


def hip_lib() -> C.CDLL:
    """libprt_hip.so: the HIP kernels + C ABI."""
    global _hip
    if _hip is None:
        # PRT_HIP_LIB names a variant build of the same ABI inside the package directory (make hip-experimental, hip-bvh8;
        # tools/ab_*.sh); the product is libprt_hip.so
        lib = _load(os.path.join(PKG_DIR, os.path.basename(os.environ.get("PRT_HIP_LIB", "libprt_hip.so"))), "HIP extension")
        lib.prt_create.restype = C.c_void_p
        lib.prt_create.argtypes = [C.c_int]
        lib.prt_destroy.argtypes = [C.c_void_p]
        lib.prt_last_error.restype = C.c_char_p
        lib.prt_last_error.argtypes = [C.c_void_p]
        lib.prt_abi_version.restype = C.c_int
        lib.prt_build_flags.restype = C.c_int
        lib.prt_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        lib.prt_upload_scene.argtypes = [C.c_void_p, C.POINTER(PrtSceneDesc)]
        lib.prt_render.argtypes = [C.c_void_p, C.POINTER(PrtCamera), C.POINTER(PrtParams), C.c_uint32, C.c_uint32,
                                   C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(PrtCounters)]
        lib.prt_render_device.argtypes = [C.c_void_p, C.POINTER(PrtCamera), C.POINTER(PrtParams), C.c_uint32, C.c_uint32,
                                          C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(PrtCounters)]
        lib.prt_shard_rows.restype = C.c_uint32
        lib.prt_shard_rows.argtypes = [C.c_uint32] * 4
        lib.prt_render_shard_device.argtypes = [C.c_void_p, C.POINTER(PrtCamera), C.POINTER(PrtParams), C.c_uint32,
                                                C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p,
                                                C.POINTER(PrtCounters)]
        lib.prt_render_shard.argtypes = lib.prt_render_shard_device.argtypes
        lib.prt_render_pixel_list.argtypes = [C.c_void_p, C.POINTER(PrtCamera), C.POINTER(PrtParams), C.c_uint32, C.c_uint32,
                                              C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(PrtCounters)]
        lib.prt_get_scene_info.argtypes = [C.c_void_p, C.POINTER(PrtSceneInfo)]
        lib.prt_get_render_stats.argtypes = [C.c_void_p, C.POINTER(PrtRenderStats)]
        lib.prt_debug_check_bvh.argtypes = [C.POINTER(PrtSceneDesc), C.POINTER(C.c_uint64)]
        lib.prt_debug_check_bvh_lbvh.argtypes = [C.c_void_p, C.POINTER(PrtSceneDesc), C.POINTER(C.c_uint64)]
        lib.prt_debug_device_kat.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint32,
                                             C.POINTER(PrtCamera)]
        lib.prt_multi_create.restype = C.c_void_p
        lib.prt_multi_create.argtypes = [C.POINTER(C.c_int), C.c_int]
        lib.prt_multi_destroy.argtypes = [C.c_void_p]
        lib.prt_multi_last_error.restype = C.c_char_p
        lib.prt_multi_last_error.argtypes = [C.c_void_p]
        lib.prt_multi_device_count.argtypes = [C.c_void_p]
        lib.prt_multi_context.restype = C.c_void_p
        lib.prt_multi_context.argtypes = [C.c_void_p, C.c_int]
        lib.prt_multi_upload_scene.argtypes = [C.c_void_p, C.POINTER(PrtSceneDesc)]
        lib.prt_multi_render.argtypes = [C.c_void_p, C.POINTER(PrtCamera), C.POINTER(PrtParams), C.c_uint32, C.c_uint32,
                                         C.c_void_p, C.POINTER(PrtCounters)]
        lib.prt_multi_depth.argtypes = [C.c_void_p]
        lib.prt_multi_submit.argtypes = [C.c_void_p, C.POINTER(PrtCamera), C.POINTER(PrtParams), C.c_uint32, C.c_uint32,
                                         C.c_void_p, C.POINTER(C.c_uint64)]
        lib.prt_multi_wait.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(PrtCounters)]
        lib.prt_debug_throw.argtypes = [C.c_void_p, C.c_int]
        _hip = lib
    return _hip


def host_lib() -> C.CDLL:
    """libprt_host.so: the C++ host mirror of the reference driver (loader, hierarchy, tone map, Render)."""
    global _host
    if _host is None:
        hip_lib()   # dependency; loads first so the error names the right library
        lib = _load(os.path.join(PKG_DIR, "libprt_host.so"), "host library")
        lib.prt_host_load_obj.restype = C.c_void_p
        lib.prt_host_load_obj.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_float)]
        lib.prt_host_free_scene.argtypes = [C.c_void_p]
        lib.prt_host_scene_desc.restype = C.POINTER(PrtSceneDesc)
        lib.prt_host_scene_desc.argtypes = [C.c_void_p]
        lib.prt_host_scene_hierarchy_seconds.restype = C.c_double
        lib.prt_host_scene_hierarchy_seconds.argtypes = [C.c_void_p]
        lib.prt_host_scene_parse_seconds.restype = C.c_double
        lib.prt_host_scene_parse_seconds.argtypes = [C.c_void_p]
        lib.prt_host_last_error.restype = C.c_char_p
        lib.prt_host_make_camera.argtypes = [C.c_float, C.c_uint32, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                             C.POINTER(PrtCamera)]
        lib.prt_host_default_params.argtypes = [C.c_uint32, C.c_uint64, C.POINTER(PrtParams)]
        lib.prt_host_render.argtypes = [C.c_void_p, C.POINTER(PrtCamera), C.POINTER(PrtParams), C.c_uint32, C.c_uint32,
                                        C.c_int, C.c_void_p, C.POINTER(PrtCounters)]
        lib.prt_host_render_error.restype = C.c_char_p
        lib.prt_host_write_image.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_char_p]
        lib.prt_host_tonemap.restype = C.c_float
        lib.prt_host_tonemap.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        lib.prt_host_load_texture.restype = C.POINTER(C.c_uint8)
        lib.prt_host_load_texture.argtypes = [C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        lib.prt_host_free_texture.argtypes = [C.POINTER(C.c_uint8)]
        _host = lib
    return _host


def np_from_ptr(ptr, count: int, dtype) -> np.ndarray:
    """Copy `count` elements of `dtype` from a ctypes pointer into a numpy array."""
    if count == 0:
        return np.zeros(0, dtype=dtype)
    nbytes = count * np.dtype(dtype).itemsize
    buf = (C.c_char * nbytes).from_address(C.addressof(ptr.contents))
    return np.frombuffer(buf, dtype=dtype, count=count).copy()
