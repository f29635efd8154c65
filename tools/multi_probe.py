"""The C++ product path for n GPUs (prt_multi_*), rehearsed on ONE GPU (every "device" is GPU 0): ms per C4 frame with one
frame at a time (prt_multi_render) and with two in flight (prt_multi_submit frame k + 1 before prt_multi_wait of frame k),
persistent worker threads either way.  Frames checked bit for bit against prt_render.
    python tools/multi_probe.py [frames]"""
import sys, os, tempfile, time, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from par_raytracer_amd import api, scenes, capi
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 12
lib = capi.hip_lib()
s = scenes.make_scene("terrain_1m"); d = tempfile.mkdtemp(); scenes.write_obj(s, d, "scene.obj")
hs = api.HostScene(d, "scene.obj", 0, s.camera_position)
w, h = 1920, 1080
cam = api.make_camera(s.fov, w, h, s.camera_position, s.camera_facing)
p = api.default_params(8, 1234)
r = api.Renderer(0); r.upload(hs)
want, wc = r.render(cam, p, w, h)
r.close()
for n in (1, 2, 4, 8):
    ids = (C.c_int * n)(*([0] * n))
    m = lib.prt_multi_create(ids, n)
    assert m, lib.prt_multi_last_error(None)
    assert lib.prt_multi_upload_scene(m, hs.desc) == 0, lib.prt_multi_last_error(m)
    outs = [np.zeros((h * w, 4), dtype=np.float32) for _ in range(2)]
    ctr = capi.PrtCounters()
    for _ in range(2):
        assert lib.prt_multi_render(m, C.byref(cam), C.byref(p), w, h, outs[0].ctypes.data, C.byref(ctr)) == 0, lib.prt_multi_last_error(m)
    assert np.array_equal(outs[0].view(np.uint32), want.view(np.uint32)) and ctr.ray_count == wc.ray_count
    t0 = time.perf_counter()
    for _ in range(frames):
        assert lib.prt_multi_render(m, C.byref(cam), C.byref(p), w, h, outs[0].ctypes.data, C.byref(ctr)) == 0
    one = (time.perf_counter() - t0) / frames * 1e3
    dev_ms = ctr.render_ms
    # two in flight
    t = [C.c_uint64(0), C.c_uint64(0)]
    t0 = time.perf_counter()
    for k in range(frames):
        assert lib.prt_multi_submit(m, C.byref(cam), C.byref(p), w, h, outs[k & 1].ctypes.data, C.byref(t[k & 1])) == 0, lib.prt_multi_last_error(m)
        if k:
            assert lib.prt_multi_wait(m, t[(k - 1) & 1], C.byref(ctr)) == 0, lib.prt_multi_last_error(m)
    assert lib.prt_multi_wait(m, t[(frames - 1) & 1], C.byref(ctr)) == 0
    two = (time.perf_counter() - t0) / frames * 1e3
    ok = all(np.array_equal(o.view(np.uint32), want.view(np.uint32)) for o in outs)
    print("n = %d \"devices\" on one GPU: one frame at a time %.3f ms per frame (host wall, download included; slowest shard %.3f ms on the device), "
          "two in flight %.3f ms per frame (%.1f %% less); frames bit-identical to prt_render: %s" % (n, one, dev_ms, two, 100.0 * (one - two) / one, ok), flush=True)
    lib.prt_multi_destroy(m)
