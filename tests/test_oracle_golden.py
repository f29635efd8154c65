"""The CPU oracle (restatement) against the compiled reference's framebuffers.

tests/golden/*.npz hold the UNMODIFIED reference's float RGB and DebugCounters for every BASELINE config
(full frames for C1/C2, sparse lattices for C3-C5) plus path-coverage scenes (two lights, point light,
deep trees, multi-sample bounces).  The oracle must reproduce them BIT FOR BIT, counters included - that is
what pins it.  When oracle/_ref/ref_harness exists (build container) the same comparison is
also run live on freshly chosen parameters.
"""
from __future__ import annotations

import numpy as np
import pytest

from conftest import camera_and_params, host_scene, load_golden, scene_dir

import oracle_py as orc

FAST = ["c1_sphere_plane_256", "c2_cornell_128", "c2_cornell_512_l4", "cornell_point_light_d5",
        "icosphere_l3_two_lights", "terrain64_d3", "coincident_192x144_d3", "coincident_two_lights", "gallery_160x120", "gallery_two_lights_d4",
        "cornell_adaptive_4_16", "gallery_adaptive_10_50", "terrain64_adaptive_3_12_d4", "many_materials_two_lights",
        "jpeg_gallery_128x96", "png_gallery_128x96", "bmp_gallery_128x96", "tga_gallery_128x96", "gif_gallery_128x96", "psd_gallery_128x96", "hdr_gallery_128x96", "pic_gallery_128x96"]
SLOW = ["terrain192_d2", "c3_icosphere_1080p_l24", "c4_terrain1m_1080p_l40", "c4_terrain1m_adaptive_l60", "c5_terrain1m_4k_l120"]


def _check(name, threads=8):
    g = load_golden(name)
    hs = host_scene(str(g["scene"]), int(g["light_mode"]))
    cam, p = camera_and_params(g)
    img, ctr = orc.render(hs.desc, cam, p, int(g["width"]), int(g["height"]), int(g["lattice"]), threads)
    assert np.all(img[:, :, 3] == 1.0)
    same = np.array_equal(np.ascontiguousarray(img[:, :, :3]).view(np.uint32), g["rgb"].view(np.uint32))
    assert same, "oracle differs from the reference: max|d| = %g" % np.abs(img[:, :, :3] - g["rgb"]).max()
    assert ctr.ray_count == int(g["ray_count"])
    assert ctr.sphere_check_count == int(g["sphere_check_count"])
    assert ctr.mesh_check_count == int(g["mesh_check_count"])


@pytest.mark.parametrize("name", FAST)
def test_oracle_bit_identical_to_reference(name):
    _check(name)


@pytest.mark.slow
@pytest.mark.parametrize("name", SLOW)
def test_oracle_bit_identical_to_reference_large(name):
    _check(name)


def test_thread_count_does_not_change_pixels():
    g = load_golden("c2_cornell_128")
    hs = host_scene("cornell_box")
    cam, p = camera_and_params(g)
    a, ca = orc.render(hs.desc, cam, p, 128, 128, 1, 1)
    b, cb = orc.render(hs.desc, cam, p, 128, 128, 1, 5)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert ca.ray_count == cb.ray_count and ca.sphere_check_count == cb.sphere_check_count


def test_lattice_is_a_subset_of_the_full_frame():
    g = load_golden("c2_cornell_128")
    hs = host_scene("cornell_box")
    cam, p = camera_and_params(g)
    full, _ = orc.render(hs.desc, cam, p, 128, 128, 1, 4)
    lat, _ = orc.render(hs.desc, cam, p, 128, 128, 4, 4)
    assert np.array_equal(full[::4, ::4].view(np.uint32), lat.view(np.uint32))


@pytest.mark.skipif(not orc.have_reference(), reason="compiled reference only exists in the build container")
@pytest.mark.parametrize("scene,w,h,spp,depth,lm,rs,ss,seed", [
    ("cornell_box", 40, 30, 3, 4, 1, 1, 2, 99),
    ("sphere_plane", 48, 48, 2, 2, 2, 2, 1, 7),
    ("terrain_64", 32, 24, 2, 6, 0, 1, 1, 424242),
    ("textured_gallery", 56, 40, 3, 5, 2, 1, 2, 31337),          # textures + point light + deep alpha chains
    ("coincident", 64, 48, 3, 4, 1, 1, 1, 2025),                 # coplanar / ulp-offset / doubled faces: visit order decides
])
def test_live_against_compiled_reference(scene, w, h, spp, depth, lm, rs, ss, seed):
    from par_raytracer_amd import api
    s, d = scene_dir(scene)
    ref = orc.run_reference(d, "scene.obj", w, h, spp, seed, s.camera_position, s.camera_facing, s.fov, bounce_depth=depth,
                            reflection_samples=rs, spec_samples=ss, light_mode=lm)
    hs = host_scene(scene, lm)
    cam = api.make_camera(s.fov, w, h, s.camera_position, s.camera_facing)
    p = api.default_params(spp, seed, bounce_depth=depth, reflection_samples=rs, spec_samples=ss)
    img, ctr = orc.render(hs.desc, cam, p, w, h, 1, 4)
    assert np.array_equal(img.view(np.uint32), ref["pixels"].view(np.uint32))
    st = ref["stats"]
    assert (ctr.ray_count, ctr.sphere_check_count, ctr.mesh_check_count) == (
        st["ray_count"], st["sphere_check_count"], st["mesh_check_count"])
