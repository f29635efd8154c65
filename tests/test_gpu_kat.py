"""Device functions of the hot path against the reference's per-function known-answer vectors, bit for bit.

kat.npz was produced by calling the unmodified reference's own functions (oracle/gen_golden.py).  Each test
pushes the same inputs through ONE device function via prt_debug_device_kat (csrc/kernels_debug.h) and requires
identical bits: the PRNG (including the u64 -> f32 conversion), both RNG variants, the triangle test on
host-pre-differenced operands, the bounce-direction frame, Fresnel, camera rays.
"""
from __future__ import annotations

import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_golden, scene_dir

pytestmark = pytest.mark.gpu

KAT_RNG_NEXT, KAT_RNG_FLOAT01, KAT_TRIANGLE, KAT_DIFFUSE_DIR, KAT_CAMERA_RAY, KAT_FRESNEL, KAT_TTW, KAT_RNG_COMPACT = range(8)


@pytest.fixture(scope="module")
def kat():
    return load_golden("kat")


@pytest.fixture(scope="module")
def renderer(gpu_renderer_factory):
    return gpu_renderer_factory("sphere_plane", 0)


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_device_rng_matches_reference(renderer, kat):
    seeds = np.ascontiguousarray(kat["rng_seed"], dtype=np.uint64).reshape(-1, 1)
    n = len(seeds)
    full = renderer.device_kat(KAT_RNG_NEXT, seeds, (n, 40), np.uint64)
    assert np.array_equal(full, kat["rng_next"].reshape(n, 40)), "ring variant (wraps the 16-word state)"
    compact = renderer.device_kat(KAT_RNG_COMPACT, seeds, (n, 15), np.uint64)
    assert np.array_equal(compact, kat["rng_next"].reshape(n, 40)[:, :15]), "2-register variant, first 15 draws"
    f01 = renderer.device_kat(KAT_RNG_FLOAT01, seeds, (n, 24), np.float32)
    assert np.array_equal(_bits(f01), _bits(kat["rng_f01"].reshape(n, 24))), "u64 -> f32 conversion / clamp"


def test_device_triangle_test_matches_reference(renderer, kat):
    rec = kat["tri_in"].reshape(-1, 16).astype(np.float32)
    o, d, a, b, c, max_t = rec[:, 0:3], rec[:, 3:6], rec[:, 6:9], rec[:, 9:12], rec[:, 12:15], rec[:, 15:16]
    ab = (b - a).astype(np.float32)
    ac = (c - a).astype(np.float32)
    # Cross(ab, ac) with the reference's association, in float32 (what prt_upload_scene computes on the host)
    n = np.stack([ab[:, 1] * ac[:, 2] - ac[:, 1] * ab[:, 2],
                  ab[:, 2] * ac[:, 0] - ac[:, 2] * ab[:, 0],
                  ab[:, 0] * ac[:, 1] - ac[:, 0] * ab[:, 1]], axis=1).astype(np.float32)
    dev_in = np.concatenate([o, d, a, ab, ac, n, max_t], axis=1).astype(np.float32)
    got = renderer.device_kat(KAT_TRIANGLE, dev_in, (len(rec), 11), np.float32)
    ref = kat["tri_out"].reshape(-1, 11)
    assert np.array_equal(got[:, 0], ref[:, 0]), "hit / miss decisions differ"
    assert np.array_equal(_bits(got), _bits(ref))
    assert 200 < int(ref[:, 0].sum()) < len(ref) - 200


def test_device_bounce_directions_match_reference(renderer, kat):
    normals = kat["diffuse_normals"].reshape(-1, 3)
    ref = kat["diffuse_dirs"].reshape(len(normals), 1024, 3)
    rec = np.zeros((len(normals) * 1024, 4), dtype=np.float32)
    for k, nrm in enumerate(normals):
        rec[k * 1024:(k + 1) * 1024, :3] = nrm
        rec[k * 1024:(k + 1) * 1024, 3] = np.arange(1024, dtype=np.float32)
    got = renderer.device_kat(KAT_DIFFUSE_DIR, rec, (len(rec), 3), np.float32)
    # host table (glibc cosf/sinf, as the reference calls them) x device tangent frame == reference direction
    assert np.array_equal(_bits(got), _bits(ref.reshape(-1, 3)))


def test_device_fresnel_matches_reference(renderer, kat):
    rec = kat["fresnel_in"].reshape(-1, 7).astype(np.float32)
    got = renderer.device_kat(KAT_FRESNEL, rec, (len(rec),), np.float32)
    ref = kat["fresnel_out"]
    both_nan = np.isnan(got) & np.isnan(ref)
    assert np.array_equal(_bits(got)[~both_nan], _bits(ref)[~both_nan])


def test_device_camera_rays_match_reference(renderer, kat):
    from par_raytracer_amd import api, scenes
    s = scenes.make_scene("sphere_plane")
    cam = api.make_camera(s.fov, 256, 256, s.camera_position, s.camera_facing)
    pts = kat["camray_in"].reshape(-1, 2).astype(np.float32)
    got = renderer.device_kat(KAT_CAMERA_RAY, pts, (len(pts), 3), np.float32, cam)
    assert np.array_equal(_bits(got), _bits(kat["camray_out"].reshape(-1, 3)))


def test_driver_binary_end_to_end(tmp_path):
    """prt_main: the reference's main() call sequence on the HIP path; its ray count must equal the reference's."""
    exe = os.path.join(ROOT, "par_raytracer_amd", "prt_main")
    if not os.path.exists(exe):
        pytest.skip("prt_main not built")
    g = load_golden("c2_cornell_128")
    s, d = scene_dir("cornell_box")
    out = str(tmp_path / "out.png")
    cmd = [exe, "-d", d, "--obj", "scene.obj", "-w", "128", "-h", "128", "--spp", "4", "--seed", "1234", "--fov", "60",
           "--camera_position"] + [repr(float(v)) for v in s.camera_position] + ["--camera_facing"] + \
          [repr(float(v)) for v in s.camera_facing] + ["-o", out]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    text = p.stdout.decode()
    rays = [int(line.split(":")[1]) for line in text.splitlines() if line.startswith("Rays cast")]
    assert rays == [int(g["ray_count"])], text
    assert "Triangles: 12" in text
    data = open(out, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n" and len(data) > 1000


def test_driver_binary_adaptive_and_textured(tmp_path):
    """prt_main in the reference's adaptive mode (--spp = min_samples, --max_spp = max_samples) and on the textured scene."""
    exe = os.path.join(ROOT, "par_raytracer_amd", "prt_main")
    if not os.path.exists(exe):
        pytest.skip("prt_main not built")
    for fixture, scene in (("cornell_adaptive_4_16", "cornell_box"), ("gallery_160x120", "textured_gallery")):
        g = load_golden(fixture)
        s, d = scene_dir(scene)
        out = str(tmp_path / (fixture + ".png"))
        cmd = [exe, "-d", d, "--obj", "scene.obj", "-w", str(int(g["width"])), "-h", str(int(g["height"])), "--spp", str(int(g["spp"])),
               "--seed", str(int(g["seed"])), "--fov", repr(float(g["fov"])),
               "--camera_position"] + [repr(float(v)) for v in s.camera_position] + ["--camera_facing"] + \
              [repr(float(v)) for v in s.camera_facing] + ["-o", out]
        if "max_spp" in g and int(g["max_spp"]):
            cmd += ["--max_spp", str(int(g["max_spp"]))]
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        rays = [int(line.split(":")[1]) for line in p.stdout.decode().splitlines() if line.startswith("Rays cast")]
        assert rays == [int(g["ray_count"])], p.stdout.decode()
        assert open(out, "rb").read()[:8] == b"\x89PNG\r\n\x1a\n"
