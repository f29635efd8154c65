"""The CPU oracle against the reference's per-function known-answer vectors (tests/golden/kat.npz).

kat.npz was produced by calling the UNMODIFIED reference's own functions through oracle/ref_harness.cpp
(oracle/gen_golden.py).  Everything here must match bit for bit: integer PRNG state, u64 -> f32 conversion,
Hammersley, bounce directions (host libm cosf/sinf/powf), Fresnel, ray/triangle, ray/sphere, camera rays.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import pytest

from conftest import load_golden

import oracle_py as orc


@pytest.fixture(scope="module")
def kat():
    return load_golden("kat")


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_survey_known_answers():
    """The vectors quoted in SURVEY.md §8c."""
    lib = orc.lib()
    st = np.zeros(16, dtype=np.uint64)
    lib.prt_oracle_rng_seed_state(0, st.ctypes.data_as(C.c_void_p))
    assert st[0] == 0x884b8826b976fba1 and st[15] == 0x9e3e03cf90310ba1
    nx = np.zeros(4, dtype=np.uint64)
    lib.prt_oracle_rng_next(0, 4, nx.ctypes.data_as(C.c_void_p))
    assert [int(v) for v in nx] == [0x027a83d081c116e8, 0x7135d42b7aa52974, 0x2778e6ae0fd5b174, 0x0572415b3092e1cd]
    lib.prt_oracle_rng_next(0x835fdd9143716fe3, 4, nx.ctypes.data_as(C.c_void_p))
    assert [int(v) for v in nx] == [0x21176fc04b12dd0b, 0x5277b6247f282e23, 0x04e006d26fd9bd2d, 0x31491f889f6ac196]
    f = np.zeros(2, dtype=np.float32)
    lib.prt_oracle_rng_float11(0x835fdd9143716fe3, 2, f.ctypes.data_as(C.c_void_p))
    assert f[0] == np.float32(float.fromhex("-0x1.7ba24p-1")) and f[1] == np.float32(float.fromhex("-0x1.6c425p-2"))


def test_rng_streams(kat):
    lib = orc.lib()
    seeds = kat["rng_seed"]
    for i, seed in enumerate(seeds):
        st = np.zeros(16, dtype=np.uint64)
        lib.prt_oracle_rng_seed_state(int(seed), st.ctypes.data_as(C.c_void_p))
        assert np.array_equal(st, kat["rng_state"][16 * i:16 * i + 16])
        nx = np.zeros(40, dtype=np.uint64)
        lib.prt_oracle_rng_next(int(seed), 40, nx.ctypes.data_as(C.c_void_p))
        assert np.array_equal(nx, kat["rng_next"][40 * i:40 * i + 40])      # 40 draws: wraps the 16-word state twice
        f = np.zeros(24, dtype=np.float32)
        lib.prt_oracle_rng_float01(int(seed), 24, f.ctypes.data_as(C.c_void_p))
        assert np.array_equal(_bits(f), _bits(kat["rng_f01"][24 * i:24 * i + 24]))
        lib.prt_oracle_rng_float11(int(seed), 24, f.ctypes.data_as(C.c_void_p))
        assert np.array_equal(_bits(f), _bits(kat["rng_f11"][24 * i:24 * i + 24]))


def test_sample_key(kat):
    lib = orc.lib()
    keys = [lib.prt_oracle_sample_key(1234, (p * 1000003) & 0xFFFFFFFF, s) for p in range(8) for s in range(4)]
    assert [int(k) for k in keys] == [int(k) for k in kat["key_1234"]]
    assert len(set(keys)) == len(keys)


def test_hammersley_and_diffuse_directions(kat):
    lib = orc.lib()
    xi = np.zeros((1024, 2), dtype=np.float32)
    for i in range(1024):
        lib.prt_oracle_hammersley(i, 1024, xi[i].ctypes.data_as(C.c_void_p))
    assert np.array_equal(_bits(xi.reshape(-1)), _bits(kat["hammersley_1024"]))
    normals = kat["diffuse_normals"].reshape(-1, 3)
    ref = kat["diffuse_dirs"].reshape(len(normals), 1024, 3)
    out = np.zeros(3, dtype=np.float32)
    for k, n in enumerate(normals):
        n = np.ascontiguousarray(n, dtype=np.float32)
        for i in range(0, 1024, 7):
            lib.prt_oracle_diffuse_dir(n.ctypes.data_as(C.c_void_p), i, 1024, out.ctypes.data_as(C.c_void_p))
            assert np.array_equal(_bits(out), _bits(ref[k, i])), (k, i)


def test_specular_directions(kat):
    lib = orc.lib()
    rec = kat["spec_in"].reshape(-1, 6)
    ref = kat["spec_dirs"].reshape(-1, 3)
    out = np.zeros(3, dtype=np.float32)
    for r, d in zip(rec, ref):
        n = np.ascontiguousarray(r[:3], dtype=np.float32)
        lib.prt_oracle_specular_dir(n.ctypes.data_as(C.c_void_p), float(r[3]), int(r[4]), int(r[5]), out.ctypes.data_as(C.c_void_p))
        assert np.array_equal(_bits(out), _bits(d))


def test_fresnel(kat):
    lib = orc.lib()
    rec = kat["fresnel_in"].reshape(-1, 7)
    ref = kat["fresnel_out"]
    got = np.zeros(len(rec), dtype=np.float32)
    for i, r in enumerate(rec):
        n = np.ascontiguousarray(r[1:4], dtype=np.float32)
        d = np.ascontiguousarray(r[4:7], dtype=np.float32)
        got[i] = lib.prt_oracle_fresnel(float(r[0]), n.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p))
    # NaN results (Ni = 0) must be NaN on both sides; everything else bit-equal
    both_nan = np.isnan(got) & np.isnan(ref)
    assert np.array_equal(_bits(got)[~both_nan], _bits(ref)[~both_nan])


def test_ray_triangle(kat):
    lib = orc.lib()
    rec = np.ascontiguousarray(kat["tri_in"].reshape(-1, 16))
    ref = kat["tri_out"].reshape(-1, 11)
    out = np.zeros(11, dtype=np.float32)
    hits = 0
    for r, e in zip(rec, ref):
        lib.prt_oracle_intersect_triangle(r.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        assert np.array_equal(_bits(out), _bits(e))
        hits += int(e[0])
    assert 200 < hits < len(rec) - 200                 # the vectors exercise both outcomes


def test_ray_sphere(kat):
    lib = orc.lib()
    rec = np.ascontiguousarray(kat["sphere_in"].reshape(-1, 10))
    ref = kat["sphere_out"].reshape(-1, 2)
    out = np.zeros(2, dtype=np.float32)
    for r, e in zip(rec, ref):
        lib.prt_oracle_intersect_sphere(r.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        assert np.array_equal(_bits(out), _bits(e))


def test_camera_rays(kat):
    from par_raytracer_amd import api, scenes
    lib = orc.lib()
    s = scenes.make_scene("sphere_plane")
    cam = api.make_camera(s.fov, 256, 256, s.camera_position, s.camera_facing)
    c = kat["camera"]
    mine = np.array([cam.tan_a2, cam.aspect, cam.inv_width, cam.inv_height] + list(cam.position) + list(cam.forward) +
                    list(cam.right) + list(cam.up), dtype=np.float32)
    assert np.array_equal(_bits(mine), _bits(c)), "MakeCamera differs from the reference"
    pts = kat["camray_in"].reshape(-1, 2)
    ref = kat["camray_out"].reshape(-1, 3)
    out = np.zeros(3, dtype=np.float32)
    for p, e in zip(pts, ref):
        lib.prt_oracle_camera_ray(C.byref(cam), float(p[0]), float(p[1]), out.ctypes.data_as(C.c_void_p))
        assert np.array_equal(_bits(out), _bits(e))
