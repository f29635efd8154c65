"""Throughput of back-to-back frames when F contexts (own stream + workspace each) render concurrently from F host threads."""
import sys, os, tempfile, threading, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from par_raytracer_amd import api, scenes
s = scenes.make_scene("terrain_1m"); d = tempfile.mkdtemp(); scenes.write_obj(s, d, "scene.obj")
hs = api.HostScene(d, "scene.obj", 0, s.camera_position)
w, h = 1920, 1080
cam = api.make_camera(s.fov, w, h, s.camera_position, s.camera_facing)
p = api.default_params(8, 1234)
F_MAX = 3
rs = []
for _ in range(F_MAX):
    r = api.Renderer(0); r.upload(hs); rs.append(r)
bufs = [torch.zeros((h, w, 4), dtype=torch.float32, device="cuda") for _ in range(F_MAX)]
for nr in (1, 2, 4, 8):
    for F in (1, 2, 3):
        frames = 24
        for k in range(F): rs[k].render_shard_device(cam, p, w, h, 8, 0, nr, bufs[k].data_ptr(), False)
        def work(k):
            for _ in range(frames // F):
                rs[k].render_shard_device(cam, p, w, h, 8, 0, nr, bufs[k].data_ptr(), False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ts = [threading.Thread(target=work, args=(k,)) for k in range(F)]
        for t in ts: t.start()
        for t in ts: t.join()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / (frames // F * F)
        print("shard 1/%d, %d frames in flight: %.2f ms per frame" % (nr, F, dt * 1e3), flush=True)
