"""Thin Python objects over the C ABI: HostScene (reference-style host pipeline) and Renderer (prt_ctx).

Mirrors the reference driver's call sequence (main.cpp:537-612) so tests read like the reference's
own main(): load OBJ -> hierarchy -> scene -> camera -> Render -> image.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np

from . import capi
from .capi import PrtCamera, PrtCounters, PrtParams, PrtSceneDesc, PrtSceneInfo


class HostScene:
    """ParseOBJ + CalculateTangents + BuildHierarchy + InitScene + object list, flattened."""

    def __init__(self, directory: str, obj_name: str = "sponza.obj", light_mode: int = 0,
                 camera_position: Sequence[float] = (0.0, 0.0, 0.0)):
        lib = capi.host_lib()
        cp = (C.c_float * 3)(*[float(v) for v in camera_position])
        self._lib = lib
        self._h = lib.prt_host_load_obj(directory.encode(), obj_name.encode(), int(light_mode), cp)
        if not self._h:
            raise RuntimeError("prt_host_load_obj failed: %s" % lib.prt_host_last_error().decode())

    @property
    def handle(self):
        return self._h

    @property
    def desc(self) -> "C.POINTER(PrtSceneDesc)":
        return self._lib.prt_host_scene_desc(self._h)

    @property
    def hierarchy_seconds(self) -> float:
        return self._lib.prt_host_scene_hierarchy_seconds(self._h)

    @property
    def parse_seconds(self) -> float:
        return self._lib.prt_host_scene_parse_seconds(self._h)

    @property
    def n_tris(self) -> int:
        return self.desc.contents.index_count // 3

    def arrays(self) -> dict:
        """numpy copies of the flattened scene (for tests)."""
        d = self.desc.contents
        f = capi.np_from_ptr
        out = {
            "positions": f(d.positions, d.position_count * 3, np.float32).reshape(-1, 3),
            "normals": f(d.normals, d.normal_count * 3, np.float32).reshape(-1, 3),
            "texcoords": f(d.texcoords, d.texcoord_count * 2, np.float32).reshape(-1, 2),
            "idx_positions": f(d.idx_positions, d.index_count, np.uint32),
            "idx_texcoords": f(d.idx_texcoords, d.index_count, np.uint32),
            "idx_normals": f(d.idx_normals, d.index_count, np.uint32),
            "groups": np.array([(d.groups[i].first_index, d.groups[i].index_count, d.groups[i].material)
                                for i in range(d.group_count)], dtype=np.int64).reshape(-1, 3),
            "spheres": np.array([(tuple(d.spheres[i].center) + (d.spheres[i].radius,)) for i in range(d.sphere_count)],
                                dtype=np.float32).reshape(-1, 4),
            "sphere_children": np.array([(d.spheres[i].c0, d.spheres[i].c1) for i in range(d.sphere_count)],
                                        dtype=np.uint32).reshape(-1, 2),
            "sphere_group": f(d.sphere_group, d.sphere_count, np.int32),
            "materials": np.array([[m.specular_intensity, m.index_of_refraction, m.alpha] + list(m.ambient_color) +
                                   list(m.diffuse_color) + list(m.specular_color)
                                   for m in (d.materials[i] for i in range(d.material_count))], dtype=np.float32),
        }
        out["tangents"] = (f(d.tangents, d.normal_count * 3, np.float32).reshape(-1, 3) if d.tangents
                           else np.zeros((0, 3), dtype=np.float32))
        # texture slots per group, in the reference's slot order (ambient, diffuse, specular, alpha, bump)
        dims, blobs = [], []
        for g in range(d.group_count):
            m = d.materials[d.groups[g].material]
            for slot in (m.ambient_texture, m.diffuse_texture, m.specular_texture, m.alpha_texture, m.bump_texture):
                if slot < 0:
                    dims += [0, 0, 0]
                else:
                    t = d.textures[slot]
                    dims += [t.size_x, t.size_y, t.channels]
                    blobs.append(np.ctypeslib.as_array(t.texels, shape=(t.size_x * t.size_y * t.channels,)).copy())
        out["group_texture_dims"] = np.array(dims, dtype=np.uint32)
        out["group_texture_bytes"] = np.concatenate(blobs) if blobs else np.zeros(0, dtype=np.uint8)
        return out

    def close(self):
        if self._h:
            self._lib.prt_host_free_scene(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- prt_scene_desc <-> flat arrays (the level-1 drop-in's data: what FlattenReferenceScene hands to prt_upload_scene) ---

DESC_F32 = ("positions", "normals", "texcoords", "tangents")
DESC_U32 = ("idx_positions", "idx_texcoords", "idx_normals", "texture_dims")


def desc_arrays(desc_ptr) -> dict:
    """Every array a prt_scene_desc points to, as flat numpy copies: float / index arrays as such, the record arrays
    (prt_group 12 B, prt_material 80 B, prt_light 48 B, prt_bsphere 24 B) as raw bytes, textures as (size_x, size_y,
    channels) triples + the concatenated texel bytes - the layout of oracle/ref_harness --dump-desc."""
    d = desc_ptr.contents
    f = capi.np_from_ptr

    def raw(ptr, count, size):
        return f(C.cast(ptr, C.POINTER(C.c_uint8)), count * size, np.uint8) if count else np.zeros(0, dtype=np.uint8)

    out = {
        "positions": f(d.positions, d.position_count * 3, np.float32),
        "normals": f(d.normals, d.normal_count * 3, np.float32),
        "texcoords": f(d.texcoords, d.texcoord_count * 2, np.float32),
        "tangents": f(d.tangents, d.normal_count * 3, np.float32) if d.tangents else np.zeros(0, dtype=np.float32),
        "idx_positions": f(d.idx_positions, d.index_count, np.uint32),
        "idx_texcoords": f(d.idx_texcoords, d.index_count, np.uint32),
        "idx_normals": f(d.idx_normals, d.index_count, np.uint32),
        "groups": raw(d.groups, d.group_count, C.sizeof(capi.PrtGroup)),
        "materials": raw(d.materials, d.material_count, C.sizeof(capi.PrtMaterial)),
        "lights": raw(d.lights, d.light_count, C.sizeof(capi.PrtLight)),
        "spheres": raw(d.spheres, d.sphere_count, C.sizeof(capi.PrtBSphere)),
        "sphere_group": f(d.sphere_group, d.sphere_count, np.int32) if d.sphere_count else np.zeros(0, dtype=np.int32),
    }
    dims, blobs = [], []
    for i in range(d.texture_count):
        t = d.textures[i]
        dims += [t.size_x, t.size_y, t.channels]
        blobs.append(np.ctypeslib.as_array(t.texels, shape=(t.size_x * t.size_y * t.channels,)).copy())
    out["texture_dims"] = np.array(dims, dtype=np.uint32)
    out["texture_bytes"] = np.concatenate(blobs) if blobs else np.zeros(0, dtype=np.uint8)
    return out


class FlatDesc:
    """A prt_scene_desc built from the arrays of desc_arrays() / a tests/golden/desc_*.npz fixture.  Keeps the arrays
    alive; `.desc` is what Renderer.upload() and the oracle take."""

    def __init__(self, arrays: dict):
        a = {k: np.ascontiguousarray(np.asarray(v)) for k, v in arrays.items()}
        self._keep = a
        d = capi.PrtSceneDesc()

        def ptr(key, ctype):
            arr = a[key]
            return arr.ctypes.data_as(C.POINTER(ctype)) if arr.size else C.cast(None, C.POINTER(ctype))

        d.positions = ptr("positions", C.c_float); d.position_count = a["positions"].size // 3
        d.normals = ptr("normals", C.c_float); d.normal_count = a["normals"].size // 3
        d.texcoords = ptr("texcoords", C.c_float); d.texcoord_count = a["texcoords"].size // 2
        d.tangents = ptr("tangents", C.c_float)
        d.idx_positions = ptr("idx_positions", C.c_uint32)
        d.idx_texcoords = ptr("idx_texcoords", C.c_uint32)
        d.idx_normals = ptr("idx_normals", C.c_uint32)
        d.index_count = a["idx_positions"].size
        d.groups = ptr("groups", capi.PrtGroup); d.group_count = a["groups"].size // C.sizeof(capi.PrtGroup)
        d.materials = ptr("materials", capi.PrtMaterial); d.material_count = a["materials"].size // C.sizeof(capi.PrtMaterial)
        d.lights = ptr("lights", capi.PrtLight); d.light_count = a["lights"].size // C.sizeof(capi.PrtLight)
        d.spheres = ptr("spheres", capi.PrtBSphere); d.sphere_count = a["spheres"].size // C.sizeof(capi.PrtBSphere)
        d.sphere_group = ptr("sphere_group", C.c_int32)
        dims = a["texture_dims"].reshape(-1, 3)
        self._tex = (capi.PrtTexture * max(1, len(dims)))()
        off = 0
        base = a["texture_bytes"].ctypes.data
        for i, (sx, sy, ch) in enumerate(dims):
            self._tex[i].size_x, self._tex[i].size_y, self._tex[i].channels = int(sx), int(sy), int(ch)
            self._tex[i].texels = C.cast(base + off, C.POINTER(C.c_uint8))
            off += int(sx) * int(sy) * int(ch)
        d.textures = C.cast(self._tex, C.POINTER(capi.PrtTexture)) if len(dims) else C.cast(None, C.POINTER(capi.PrtTexture))
        d.texture_count = len(dims)
        self._desc = d

    @property
    def desc(self):
        return C.pointer(self._desc)


def make_camera(fov: float, width: int, height: int, position: Sequence[float], facing: Sequence[float]) -> PrtCamera:
    cam = PrtCamera()
    p = (C.c_float * 3)(*[float(v) for v in position])
    f = (C.c_float * 3)(*[float(v) for v in facing])
    capi.host_lib().prt_host_make_camera(float(fov), int(width), int(height), p, f, C.byref(cam))
    return cam


def default_params(spp: int, seed: int = 1234, bounce_depth: Optional[int] = None, reflection_samples: Optional[int] = None,
                   spec_samples: Optional[int] = None, pipeline: int = 0, max_spp: int = 0,
                   variance_threshold: float = 0.0) -> PrtParams:
    """`max_spp` > spp turns on the reference's adaptive loop (main.cpp:245-258): spp fixed samples, then up to max_spp."""
    p = PrtParams()
    capi.host_lib().prt_host_default_params(int(spp), int(seed), C.byref(p))
    if bounce_depth is not None:
        p.bounce_depth = int(bounce_depth)
    if reflection_samples is not None:
        p.reflection_samples = int(reflection_samples)
    if spec_samples is not None:
        p.spec_samples = int(spec_samples)
    p.pipeline = int(pipeline)
    p.max_spp = int(max_spp)
    p.variance_threshold = float(variance_threshold)
    return p


class Renderer:
    """prt_ctx: one HIP device, one uploaded scene."""

    def __init__(self, device_id: int = 0):
        lib = capi.hip_lib()
        self._lib = lib
        self._ctx = lib.prt_create(int(device_id))
        if not self._ctx:
            raise RuntimeError("prt_create(%d) failed: %s" % (device_id, lib.prt_last_error(None).decode()))
        self.device_id = device_id

    def _check(self, rc: int, what: str):
        if rc != 0:
            raise RuntimeError("%s failed (%d): %s" % (what, rc, self._lib.prt_last_error(self._ctx).decode()))

    def set_option(self, name: str, value=None) -> None:
        """prt_set_option: one entry of the context's option table (csrc/prt_options.h); value None restores the default.
        The environment (PRT_<NAME>) is only read when the context is created."""
        v = None if value is None else str(value).encode()
        self._check(self._lib.prt_set_option(self._ctx, name.encode(), v), "prt_set_option(%s)" % name)

    def upload(self, scene) -> PrtSceneInfo:
        desc = scene.desc if isinstance(scene, (HostScene, FlatDesc)) else scene
        self._check(self._lib.prt_upload_scene(self._ctx, desc), "prt_upload_scene")
        return self.scene_info()

    def scene_info(self) -> PrtSceneInfo:
        info = PrtSceneInfo()
        self._check(self._lib.prt_get_scene_info(self._ctx, C.byref(info)), "prt_get_scene_info")
        return info

    def render_stats(self) -> "capi.PrtRenderStats":
        """Diagnostics of the last render made with FLAG_COUNT_VISITS (lane utilisation, phase times, parked rays)."""
        st = capi.PrtRenderStats()
        self._check(self._lib.prt_get_render_stats(self._ctx, C.byref(st)), "prt_get_render_stats")
        return st

    def render(self, cam: PrtCamera, params: PrtParams, width: int, height: int, start_idx: int = 0,
               end_idx: Optional[int] = None) -> Tuple[np.ndarray, PrtCounters]:
        if end_idx is None:
            end_idx = width * height
        out = np.empty((end_idx - start_idx, 4), dtype=np.float32)
        counters = PrtCounters()
        self._check(self._lib.prt_render(self._ctx, C.byref(cam), C.byref(params), width, height, start_idx, end_idx,
                                         out.ctypes.data_as(C.c_void_p), C.byref(counters)), "prt_render")
        return out, counters

    def render_pixels(self, cam: PrtCamera, params: PrtParams, width: int, height: int, pixel_ids) -> Tuple[np.ndarray, PrtCounters]:
        ids = np.ascontiguousarray(pixel_ids, dtype=np.uint32).reshape(-1)
        out = np.empty((ids.size, 4), dtype=np.float32)
        counters = PrtCounters()
        self._check(self._lib.prt_render_pixel_list(self._ctx, C.byref(cam), C.byref(params), width, height,
                                                    ids.ctypes.data_as(C.c_void_p), ids.size,
                                                    out.ctypes.data_as(C.c_void_p), C.byref(counters)), "prt_render_pixel_list")
        return out, counters

    def render_lattice(self, cam, params, width: int, height: int, lattice: int) -> Tuple[np.ndarray, PrtCounters]:
        """Pixels (x % lattice == 0, y % lattice == 0), returned as [lh, lw, 4] like the oracle's lattice output."""
        xs = np.arange(0, width, lattice, dtype=np.uint32)
        ys = np.arange(0, height, lattice, dtype=np.uint32)
        ids = (ys[:, None] * np.uint32(width) + xs[None, :]).reshape(-1)
        out, ctr = self.render_pixels(cam, params, width, height, ids)
        return out.reshape(len(ys), len(xs), 4), ctr

    def render_device(self, cam, params, width, height, start_idx, end_idx, d_ptr: int,
                      want_counters: bool = True) -> Optional[PrtCounters]:
        counters = PrtCounters() if want_counters else None
        self._check(self._lib.prt_render_device(self._ctx, C.byref(cam), C.byref(params), width, height, start_idx,
                                                end_idx, C.c_void_p(d_ptr),
                                                C.byref(counters) if want_counters else None), "prt_render_device")
        return counters

    def shard_rows(self, height: int, block_rows: int, rank: int, nranks: int) -> int:
        return int(self._lib.prt_shard_rows(height, block_rows, rank, nranks))

    def render_shard_device(self, cam, params, width, height, block_rows, rank, nranks, d_ptr: int,
                            want_counters: bool = True) -> Optional[PrtCounters]:
        counters = PrtCounters() if want_counters else None
        self._check(self._lib.prt_render_shard_device(self._ctx, C.byref(cam), C.byref(params), width, height,
                                                      block_rows, rank, nranks, C.c_void_p(d_ptr),
                                                      C.byref(counters) if want_counters else None),
                    "prt_render_shard_device")
        return counters

    def render_shard(self, cam, params, width, height, block_rows, rank, nranks) -> Tuple[np.ndarray, PrtCounters]:
        rows = self.shard_rows(height, block_rows, rank, nranks)
        out = np.empty((rows, width, 4), dtype=np.float32)
        counters = PrtCounters()
        self._check(self._lib.prt_render_shard(self._ctx, C.byref(cam), C.byref(params), width, height, block_rows,
                                               rank, nranks, out.ctypes.data_as(C.c_void_p), C.byref(counters)),
                    "prt_render_shard")
        return out, counters

    def device_kat(self, kind: int, records: np.ndarray, out_shape, out_dtype, cam: Optional[PrtCamera] = None) -> np.ndarray:
        """Run one device function on `records` (see csrc/kernels_debug.h); test hook."""
        rec = np.ascontiguousarray(records)
        out = np.zeros(out_shape, dtype=out_dtype)
        self._check(self._lib.prt_debug_device_kat(self._ctx, int(kind), rec.ctypes.data_as(C.c_void_p), rec.nbytes,
                                                   out.ctypes.data_as(C.c_void_p), out.nbytes, rec.shape[0],
                                                   C.byref(cam) if cam is not None else None), "prt_debug_device_kat")
        return out

    def close(self):
        if self._ctx:
            self._lib.prt_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
