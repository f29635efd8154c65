"""ctypes binding of oracle/libprt_oracle.so (the CPU restatement) and a runner for oracle/_ref/ref_harness.

TEST INFRASTRUCTURE.  Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import subprocess
import tempfile
from typing import Dict, Optional, Sequence, Tuple

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PORT_LIB = os.path.join(HERE, "libprt_oracle.so")
REF_BIN = os.path.join(HERE, "_ref", "ref_harness")


class OracleCounters(C.Structure):
    _fields_ = [("ray_count", C.c_uint64), ("sphere_check_count", C.c_uint64), ("mesh_check_count", C.c_uint64),
                ("tri_tests", C.c_uint64), ("render_seconds", C.c_double), ("threads", C.c_uint32),
                ("reserved", C.c_uint32)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_ if n != "reserved"}


_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(PORT_LIB):
            raise RuntimeError("oracle/libprt_oracle.so not built: run `make -C oracle port`")
        l = C.CDLL(PORT_LIB)
        l.prt_oracle_render.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                        C.c_uint32, C.c_void_p, C.POINTER(OracleCounters)]
        l.prt_oracle_set_worker_cpus.argtypes = [C.POINTER(C.c_int), C.c_uint32]
        l.prt_oracle_rng_seed_state.argtypes = [C.c_uint64, C.c_void_p]
        l.prt_oracle_rng_next.argtypes = [C.c_uint64, C.c_uint32, C.c_void_p]
        l.prt_oracle_rng_float01.argtypes = [C.c_uint64, C.c_uint32, C.c_void_p]
        l.prt_oracle_rng_float11.argtypes = [C.c_uint64, C.c_uint32, C.c_void_p]
        l.prt_oracle_sample_key.restype = C.c_uint64
        l.prt_oracle_sample_key.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
        l.prt_oracle_hammersley.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p]
        l.prt_oracle_diffuse_dir.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        l.prt_oracle_specular_dir.argtypes = [C.c_void_p, C.c_float, C.c_uint32, C.c_uint32, C.c_void_p]
        l.prt_oracle_fresnel.restype = C.c_float
        l.prt_oracle_fresnel.argtypes = [C.c_float, C.c_void_p, C.c_void_p]
        l.prt_oracle_intersect_triangle.argtypes = [C.c_void_p, C.c_void_p]
        l.prt_oracle_intersect_sphere.argtypes = [C.c_void_p, C.c_void_p]
        l.prt_oracle_camera_ray.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        _lib = l
    return _lib


def set_worker_cpus(cpus) -> None:
    """Worker thread t of render() pins itself to cpus[t % len(cpus)]; an empty list removes the pinning."""
    arr = (C.c_int * max(1, len(cpus)))(*cpus)
    lib().prt_oracle_set_worker_cpus(arr, len(cpus))


def socket0_physical_cpus():
    """One logical CPU per physical core of socket 0, among the CPUs this process may run on (BASELINE.md section 3's
    protocol for the CPU baseline), and the CPU model name."""
    allowed = sorted(os.sched_getaffinity(0))
    per_core = {}
    for c in allowed:
        base = "/sys/devices/system/cpu/cpu%d/topology/" % c
        try:
            pkg = int(open(base + "physical_package_id").read())
            core = int(open(base + "core_id").read())
        except (OSError, ValueError):
            pkg, core = 0, c
        per_core.setdefault((pkg, core), c)
    pkgs = sorted({k[0] for k in per_core})
    first = pkgs[0] if pkgs else 0
    cpus = sorted(v for k, v in per_core.items() if k[0] == first)
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return cpus, model


def render(desc_ptr, cam, params, width: int, height: int, lattice: int = 1, threads: int = 1
           ) -> Tuple[np.ndarray, OracleCounters]:
    """CPU restatement render.  desc_ptr: POINTER(PrtSceneDesc) (e.g. HostScene.desc); cam/params: ctypes structs.
    Returns ([lh, lw, 4] float32, counters)."""
    lw, lh = (width + lattice - 1) // lattice, (height + lattice - 1) // lattice
    out = np.empty((lh, lw, 4), dtype=np.float32)
    ctr = OracleCounters()
    rc = lib().prt_oracle_render(C.cast(desc_ptr, C.c_void_p), C.cast(C.pointer(cam), C.c_void_p),
                                 C.cast(C.pointer(params), C.c_void_p), width, height, lattice, threads,
                                 out.ctypes.data_as(C.c_void_p), C.byref(ctr))
    if rc != 0:
        raise RuntimeError("prt_oracle_render failed: %d" % rc)
    return out, ctr


# ---------------------------------------------------------------------------------------------------
# compiled reference (build container only)
# ---------------------------------------------------------------------------------------------------

def have_reference() -> bool:
    return os.path.exists(REF_BIN) and os.access(REF_BIN, os.X_OK)


def read_sections(path: str) -> Dict[str, bytes]:
    """Parse the [u32 len][name][u64 nbytes][data] records ref_harness writes."""
    out = {}
    with open(path, "rb") as f:
        data = f.read()
    off = 0
    while off < len(data):
        n = int(np.frombuffer(data, dtype="<u4", count=1, offset=off)[0]); off += 4
        name = data[off:off + n].decode(); off += n
        nb = int(np.frombuffer(data, dtype="<u8", count=1, offset=off)[0]); off += 8
        out[name] = data[off:off + nb]; off += nb
    return out


def run_reference(directory: str, obj_name: str, width: int, height: int, spp: int, seed: int,
                  camera_position: Sequence[float], camera_facing: Sequence[float], fov: float = 60.0,
                  bounce_depth: int = 2, reflection_samples: int = 1, spec_samples: int = 1, lattice: int = 1,
                  light_mode: int = 0, render: bool = True, dump_scene: bool = False, kat: bool = False,
                  timeout: float = 3600.0, adaptive_max: int = 0, dump_desc: bool = False, write_png: bool = False) -> dict:
    """Run the compiled, unmodified reference through ref_harness.  Returns {'pixels', 'stats', 'scene', 'kat', 'desc',
    'png'}: 'desc' = the arrays of FlattenReferenceScene over the reference's scene graph, 'png' = the bytes of the file
    the reference's own WriteFramebufferImage wrote for the rendered frame (lattice 1 only)."""
    if not have_reference():
        raise RuntimeError("oracle/_ref/ref_harness is not built (needs /root/reference; `make -C oracle ref`)")
    with tempfile.TemporaryDirectory() as tmp:
        cmd = [REF_BIN, "-w", str(width), "-h", str(height), "--fov", repr(float(fov)),
               "--camera_position"] + [repr(float(v)) for v in camera_position] + \
              ["--camera_facing"] + [repr(float(v)) for v in camera_facing] + \
              ["--bounce_depth", str(bounce_depth), "--reflection_samples", str(reflection_samples),
               "--specular_samples", str(spec_samples),
               "-d", directory.rstrip("/") + "/", "--obj", obj_name, "--spp", str(spp), "--seed", str(seed),
               "--lattice", str(lattice), "--light-mode", str(light_mode), "--stats", os.path.join(tmp, "stats.json")]
        if adaptive_max:
            cmd += ["--adaptive", str(adaptive_max)]
        if render:
            cmd += ["--out", os.path.join(tmp, "out.f32")]
        if dump_scene:
            cmd += ["--dump-scene", os.path.join(tmp, "scene.bin")]
        if kat:
            cmd += ["--kat", os.path.join(tmp, "kat.bin")]
        if dump_desc:
            cmd += ["--dump-desc", os.path.join(tmp, "desc.bin")]
        if write_png and render and lattice == 1:
            cmd += ["--write-png", os.path.join(tmp, "out.png")]
        proc = subprocess.run(cmd, cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
        if proc.returncode != 0:
            raise RuntimeError("ref_harness failed (%d): %s" % (proc.returncode, proc.stderr.decode()[-2000:]))
        with open(os.path.join(tmp, "stats.json")) as f:
            stats = json.load(f)
        res = {"stats": stats, "pixels": None, "scene": None, "kat": None, "desc": None, "png": None}
        if dump_desc:
            res["desc"] = read_sections(os.path.join(tmp, "desc.bin"))
        if write_png and render and lattice == 1:
            with open(os.path.join(tmp, "out.png"), "rb") as f:
                res["png"] = f.read()
        if render:
            lw, lh = stats["lattice_width"], stats["lattice_height"]
            res["pixels"] = np.fromfile(os.path.join(tmp, "out.f32"), dtype=np.float32).reshape(lh, lw, 4)
        if dump_scene:
            res["scene"] = read_sections(os.path.join(tmp, "scene.bin"))
        if kat:
            res["kat"] = read_sections(os.path.join(tmp, "kat.bin"))
        return res


def decode_png_rgba8(data: bytes) -> np.ndarray:
    """Minimal PNG reader for what stbi_write_png(..., comp = 4, ...) writes (8-bit RGBA, not interlaced): [h, w, 4] u8.
    Test infrastructure: lets the golden fixtures hold the bytes of the reference's own output file without going
    through this repository's image decoders."""
    import struct
    import zlib
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    off, idat, w, h = 8, b"", 0, 0
    while off < len(data):
        n, kind = struct.unpack(">I4s", data[off:off + 8])
        body = data[off + 8:off + 8 + n]
        off += 12 + n
        if kind == b"IHDR":
            w, h, depth, ctype, _, _, interlace = struct.unpack(">IIBBBBB", body)
            assert (depth, ctype, interlace) == (8, 6, 0), "not an 8-bit RGBA non-interlaced PNG"
        elif kind == b"IDAT":
            idat += body
        elif kind == b"IEND":
            break
    raw = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(h, 1 + 4 * w)
    out = np.zeros((h, 4 * w), dtype=np.uint8)
    prev = np.zeros(4 * w, dtype=np.int32)
    for y in range(h):
        f, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        cur = np.zeros(4 * w, dtype=np.int32)
        if f == 0:
            cur = line
        elif f == 2:
            cur = (line + prev) & 255
        else:
            for i in range(4 * w):
                a = cur[i - 4] if i >= 4 else 0
                b = prev[i]
                c = prev[i - 4] if i >= 4 else 0
                if f == 1:
                    pred = a
                elif f == 3:
                    pred = (a + b) >> 1
                else:
                    p0 = a + b - c
                    pa, pb, pc = abs(p0 - a), abs(p0 - b), abs(p0 - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[i] = (line[i] + pred) & 255
        out[y] = cur.astype(np.uint8)
        prev = cur
    return out.reshape(h, w, 4)
