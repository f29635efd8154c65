"""GPU edge cases of the RenderTask boundary (main.cpp:267-283): empty and one-pixel ranges, ranges that split a frame at
awkward places, degenerate image shapes, a camera that sees nothing, and arguments the C ABI must refuse instead of crashing.
All through the C ABI, compared with the CPU oracle (itself pinned to the compiled reference, tests/test_oracle_golden.py).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import pytest

from conftest import host_scene, scene_dir

pytestmark = pytest.mark.gpu

TOL = 1e-4
PIPELINES = {"wavefront": 2, "pool": 4}


def _setup(name, w, h, facing=None, light_mode=0):
    from par_raytracer_amd import api
    s, _ = scene_dir(name)
    hs = host_scene(name, light_mode)
    cam = api.make_camera(s.fov, w, h, s.camera_position, s.camera_facing if facing is None else facing)
    return s, hs, cam


@pytest.mark.parametrize("pipeline", sorted(PIPELINES))
def test_empty_and_single_pixel_ranges(gpu_renderer_factory, pipeline):
    """RenderTask with start_idx == end_idx renders nothing (and is not an error); a one-pixel range is that pixel of the
    full frame, wherever it lies."""
    from par_raytracer_amd import api
    w, h = 37, 23
    _, hs, cam = _setup("terrain_64", w, h)
    r = gpu_renderer_factory("terrain_64", 0)
    p = api.default_params(3, 77, pipeline=PIPELINES[pipeline])
    full, cf = r.render(cam, p, w, h)
    for k in (0, 5, w * h):
        out, c = r.render(cam, p, w, h, k, k)
        assert out.shape == (0, 4) and c.ray_count == 0
    total = 0
    for k in (0, 1, 63, 64, 65, w * h - 1):
        out, c = r.render(cam, p, w, h, k, k + 1)
        assert np.array_equal(out.view(np.uint32), full[k:k + 1].view(np.uint32)), "pixel %d" % k
        assert c.ray_count >= 3                    # at least the primary ray of every sample
        total += c.ray_count
    assert total <= cf.ray_count
    # an empty pixel list
    out, c = r.render_pixels(cam, p, w, h, np.zeros((0,), dtype=np.uint32))
    assert out.shape == (0, 4) and c.ray_count == 0


@pytest.mark.parametrize("pipeline", sorted(PIPELINES))
def test_ranges_concatenate_to_the_full_frame(gpu_renderer_factory, pipeline):
    """The reference splits a frame into contiguous index ranges per rank (main.cpp:313-317): any split, at any place,
    must give the same pixels and the same total ray count as one call."""
    from par_raytracer_amd import api
    import oracle_py as orc
    w, h = 61, 29
    _, hs, cam = _setup("cornell_box", w, h)
    r = gpu_renderer_factory("cornell_box", 0)
    p = api.default_params(2, 4242, bounce_depth=3, pipeline=PIPELINES[pipeline])
    full, cf = r.render(cam, p, w, h)
    ref, c_ref = orc.render(hs.desc, cam, p, w, h, 1, 8)
    assert cf.ray_count == c_ref.ray_count
    assert np.abs(full.reshape(h, w, 4)[:, :, :3] - ref[:, :, :3]).max() <= TOL
    cuts = [0, 1, 64, 129, 130, 700, 1001, w * h - 1, w * h]
    parts, rays = [], 0
    for a, b in zip(cuts[:-1], cuts[1:]):
        out, c = r.render(cam, p, w, h, a, b)
        parts.append(out)
        rays += c.ray_count
    got = np.concatenate(parts, axis=0)
    assert np.array_equal(got.view(np.uint32), full.view(np.uint32))
    assert rays == cf.ray_count


@pytest.mark.parametrize("shape", [(1, 1), (1, 257), (300, 1), (2, 3), (65, 1)])
def test_degenerate_image_shapes_against_the_oracle(gpu_renderer_factory, shape):
    """One-pixel, one-column and one-row images (aspect ratios far from 1) on both production pipelines."""
    from par_raytracer_amd import api
    import oracle_py as orc
    w, h = shape
    _, hs, cam = _setup("sphere_plane", w, h)
    r = gpu_renderer_factory("sphere_plane", 0)
    for name, pl in PIPELINES.items():
        p = api.default_params(4, 9, pipeline=pl)
        ref, c_ref = orc.render(hs.desc, cam, p, w, h, 1, 4)
        img, c = r.render(cam, p, w, h)
        assert c.ray_count == c_ref.ray_count, name
        assert np.abs(img.reshape(h, w, 4)[:, :, :3] - ref[:, :, :3]).max() <= TOL, name
        assert np.all(img[:, 3] == 1.0)


def test_camera_that_sees_nothing_returns_the_background(gpu_renderer_factory):
    """Every primary ray misses: each sample is background_color (raytracer.cpp:425 via TraceRayColor's miss branch), one
    ray per sample, nothing shaded."""
    from par_raytracer_amd import api
    import oracle_py as orc
    w, h = 48, 20
    _, hs, cam = _setup("sphere_plane", w, h, facing=(0.1, 0.7, 0.7))          # up and away from the sphere and the plane
    r = gpu_renderer_factory("sphere_plane", 0)
    for name, pl in PIPELINES.items():
        p = api.default_params(5, 31337, pipeline=pl)
        ref, c_ref = orc.render(hs.desc, cam, p, w, h, 1, 4)
        img, c = r.render(cam, p, w, h)
        assert c_ref.ray_count == w * h * 5 and c.ray_count == c_ref.ray_count, name
        assert c.shaded_hits == 0
        assert np.array_equal(img.reshape(h, w, 4).view(np.uint32), ref.view(np.uint32)), name   # no powf on this path: exact
        bg = np.array(p.background_color[:3], dtype=np.float32)
        assert np.allclose(img[:, :3], bg[None, :], rtol=1e-6)


def test_bad_arguments_are_errors_not_crashes(gpu_renderer_factory):
    """The reference asserts and carries on (brt.h:41); the C ABI returns a negative code and a message, and the context
    stays usable."""
    from par_raytracer_amd import api, capi
    lib = capi.hip_lib()
    w, h = 16, 8
    _, hs, cam = _setup("sphere_plane", w, h)
    r = gpu_renderer_factory("sphere_plane", 0)
    out = np.empty((w * h, 4), dtype=np.float32)
    ctr = capi.PrtCounters()

    def call(p, a=0, b=w * h, ww=w, hh=h, ptr=None, camera=cam):
        return lib.prt_render(r._ctx, C.byref(camera) if camera is not None else None, C.byref(p) if p is not None else None, ww, hh, a, b,
                              out.ctypes.data_as(C.c_void_p) if ptr is None else ptr, C.byref(ctr))

    good = api.default_params(2, 5)
    assert call(good) == 0
    first = out.copy()
    cases = {
        "end < start": lambda: call(good, 10, 5),
        "end past the image": lambda: call(good, 0, w * h + 1),
        "null output": lambda: call(good, ptr=C.c_void_p(0)),
        "null params": lambda: call(None),
        "null camera": lambda: call(good, camera=None),
        "zero width": lambda: call(good, 0, 0, ww=0),
        "spp 0": lambda: call(api.default_params(0, 5)),
        "spp 65536": lambda: call(api.default_params(65536, 5)),
        "bounce depth 17": lambda: call(api.default_params(2, 5, bounce_depth=17)),
        "unknown pipeline": lambda: call(api.default_params(2, 5, pipeline=9)),
        "adaptive on the wavefront pipeline": lambda: call(api.default_params(2, 5, pipeline=2, max_spp=4)),
        "max_spp 5000": lambda: call(api.default_params(2, 5, max_spp=5000)),
    }
    for what, f in cases.items():
        rc = f()
        assert rc < 0, what
        assert len(lib.prt_last_error(r._ctx)) > 0, what
    ids = np.array([0, w * h], dtype=np.uint32)                       # second id is outside the image
    rc = lib.prt_render_pixel_list(r._ctx, C.byref(cam), C.byref(good), w, h, ids.ctypes.data_as(C.c_void_p), 2,
                                   out.ctypes.data_as(C.c_void_p), C.byref(ctr))
    assert rc < 0
    for args in ((0, 0, 1), (8, 1, 1), (8, 2, 0)):                    # block_rows 0, rank >= nranks, nranks 0
        rc = lib.prt_render_shard(r._ctx, C.byref(cam), C.byref(good), w, h, args[0], args[1], args[2],
                                  out.ctypes.data_as(C.c_void_p), C.byref(ctr))
        assert rc < 0, args
    # ... and the context still renders the same frame
    assert call(good) == 0
    assert np.array_equal(out.view(np.uint32), first.view(np.uint32))


def test_more_ranks_than_row_blocks(gpu_renderer_factory):
    """A frame with fewer 8-row blocks than GPUs: the surplus ranks own zero rows (prt_shard_rows == 0) and their call is a
    no-op, the others still tile the frame."""
    from par_raytracer_amd import api
    w, h = 40, 20                                                       # 3 row blocks (8 + 8 + 4) for 8 ranks
    _, hs, cam = _setup("cornell_box", w, h)
    r = gpu_renderer_factory("cornell_box", 0)
    p = api.default_params(2, 11)
    full, cf = r.render(cam, p, w, h)
    full = full.reshape(h, w, 4)
    rays = 0
    rows_seen = 0
    for rank in range(8):
        rows = r.shard_rows(h, 8, rank, 8)
        img, c = r.render_shard(cam, p, w, h, 8, rank, 8)
        assert img.shape[0] == rows
        if rank >= 3:
            assert rows == 0 and c.ray_count == 0
            continue
        y0 = rank * 8
        assert np.array_equal(img.view(np.uint32), full[y0:y0 + rows].view(np.uint32))
        rays += c.ray_count
        rows_seen += rows
    assert rows_seen == h and rays == cf.ray_count


def test_host_render_forgets_a_freed_scene_and_runs_on_two_contexts(monkeypatch):
    """prt_host_render (the C++ Render() behind the C entry point, main.cpp:301-358) keeps its uploaded contexts between
    calls.  A scene that is freed and replaced by another one - quite possibly at the same address - must not be served from
    that cache; and the n_gpus > 1 path (interleaved row blocks on several contexts, scattered on the host) must give the
    single-context frame bit for bit.  Both "GPUs" are device 0 here (PRT_HOST_SHARE_DEVICE)."""
    from par_raytracer_amd import api, capi
    monkeypatch.setenv("PRT_HOST_SHARE_DEVICE", "1")
    lib = capi.host_lib()
    w, h = 96, 54
    p = api.default_params(2, 4321)

    def host_render(hs, cam, n_gpus):
        out = np.zeros((h * w, 4), dtype=np.float32)
        ctr = capi.PrtCounters()
        rc = lib.prt_host_render(hs.handle, C.byref(cam), C.byref(p), w, h, n_gpus, out.ctypes.data, C.byref(ctr))
        assert rc == 0, lib.prt_host_render_error().decode()
        return out, ctr

    def direct(hs, cam):
        r = api.Renderer(0)
        r.upload(hs)
        out, ctr = r.render(cam, p, w, h)
        r.close()
        return out, ctr

    frames = []
    for name in ("cornell_box", "terrain_64", "cornell_box"):
        s, d = scene_dir(name)
        hs = api.HostScene(d, "scene.obj", 0, s.camera_position)        # a fresh host scene each time, freed below
        cam = api.make_camera(s.fov, w, h, s.camera_position, s.camera_facing)
        want, wc = direct(hs, cam)
        for n_gpus in (1, 2):
            got, gc = host_render(hs, cam, n_gpus)
            assert gc.ray_count == wc.ray_count, (name, n_gpus)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (name, n_gpus)
        frames.append(want)
        hs.close()
    assert not np.array_equal(frames[0], frames[1])
    assert np.array_equal(frames[0], frames[2])


def test_reserved_compute_units_do_not_change_the_image(monkeypatch):
    """PRT_RESERVE_CUS=8 (what bench.py sets for N > 1 with RCCL): the context's streams are created with a CU mask that
    leaves eight compute units to the collective, and the persistent grids are sized for the rest.  Same pixels, same
    ray count, on both production pipelines."""
    from par_raytracer_amd import api
    w, h = 160, 90
    s, hs, cam = _setup("terrain_64", w, h)
    for pipeline in sorted(PIPELINES):
        p = api.default_params(3, 5, pipeline=PIPELINES[pipeline])
        frames = []
        for reserve in ("0", "8"):
            monkeypatch.setenv("PRT_RESERVE_CUS", reserve)
            r = api.Renderer(0)
            try:
                r.upload(hs)
                frames.append(r.render(cam, p, w, h))
            finally:
                r.close()
        monkeypatch.delenv("PRT_RESERVE_CUS")
        (a, ca), (b, cb) = frames
        assert ca.ray_count == cb.ray_count
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_render_stats_of_a_counting_render(gpu_renderer_factory):
    """prt_get_render_stats (include/prt.h): the diagnostics bench.py's roofline block is built from.  Lane-level counts equal
    the counters of the call; wave-level counts bound them (a wave step serves 1..64 lanes); nothing was parked on a scene
    without coincident geometry."""
    from par_raytracer_amd import api, capi
    w, h = 160, 90
    _, hs, cam = _setup("terrain_64", w, h)
    r = gpu_renderer_factory("terrain_64", 0)
    for name in sorted(PIPELINES):
        p = api.default_params(4, 11, pipeline=PIPELINES[name] | capi.FLAG_COUNT_VISITS)
        img, c = r.render(cam, p, w, h)
        st = r.render_stats()
        assert st.node_visits == c.node_visits > 0 and st.tri_tests == c.tri_tests > 0
        assert st.wave_node_steps * 64 >= st.node_visits >= st.wave_node_steps > 0
        assert st.wave_tri_steps * 64 >= st.tri_tests >= st.wave_tri_steps > 0
        assert 0 < st.deepest_stack <= st.stack_bound and st.stack_lds_entries > 0
        if name == "pool":
            assert st.phase_cycles[3] >= st.phase_cycles[0] + st.phase_cycles[1] + st.phase_cycles[2] > 0
            assert st.parked_rays <= 4 and st.parked_shadow_rays == 0
