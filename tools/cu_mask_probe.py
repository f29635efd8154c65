"""Does torch work run beside the persistent render kernel?  (Advisor, round 2, finding 2.)  With PRT_RESERVE_CUS the context's
render stream is created by hipExtStreamCreateWithCUMask - a BLOCKING stream, ordered against the legacy null stream.  One
thread renders C4 frames back to back (k_pool, ~12 ms each); the main thread times a small torch kernel + stream synchronise,
issued (a) on torch's default (null) stream, (b) on a torch stream of its own (non-blocking) - what bench.py uses for N > 1.
usage: cu_mask_probe.py [reserve_cus]      (0 = plain non-blocking render stream, no mask)"""
import sys, os, tempfile, threading, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
reserve = int(sys.argv[1]) if len(sys.argv) > 1 else 8
if reserve: os.environ["PRT_RESERVE_CUS"] = str(reserve)
else: os.environ.pop("PRT_RESERVE_CUS", None)
import numpy as np, torch
from par_raytracer_amd import api, scenes
s = scenes.make_scene("terrain_1m"); d = tempfile.mkdtemp(); scenes.write_obj(s, d, "scene.obj")
hs = api.HostScene(d, "scene.obj", 0, s.camera_position)
r = api.Renderer(0); r.upload(hs)
w, h = 1920, 1080
cam = api.make_camera(s.fov, w, h, s.camera_position, s.camera_facing)
p = api.default_params(8, 1234)
buf = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
x = torch.ones(1 << 20, device="cuda"); torch.cuda.synchronize()
r.render_device(cam, p, w, h, 0, w * h, buf.data_ptr(), True)
stop = False
frames = []
def render_loop():
    while not stop:
        c = r.render_device(cam, p, w, h, 0, w * h, buf.data_ptr(), True)
        frames.append(c.render_ms)
def probe(stream, n=60):
    lat = []
    for _ in range(n):
        time.sleep(0.004)
        t0 = time.perf_counter()
        with torch.cuda.stream(stream):
            y = x * 2.0
        stream.synchronize()
        lat.append((time.perf_counter() - t0) * 1e3)
    lat.sort()
    return lat[len(lat) // 2], lat[int(len(lat) * 0.9)]
side = torch.cuda.Stream()
idle = probe(torch.cuda.default_stream()), probe(side)
t = threading.Thread(target=render_loop); t.start(); time.sleep(0.2)
n0 = len(frames); busy_null = probe(torch.cuda.default_stream()); n1 = len(frames)
ms_null = sum(frames[n0:n1]) / max(1, n1 - n0)
busy_side = probe(side); n2 = len(frames)
ms_side = sum(frames[n1:n2]) / max(1, n2 - n1)
stop = True; t.join()
print("reserve_cus %d: small torch kernel + synchronise, median / p90 ms - GPU idle: null stream %.3f / %.3f, own stream %.3f / %.3f; "
      "while k_pool frames run back to back: null stream %.3f / %.3f (frames %.2f ms), own stream %.3f / %.3f (frames %.2f ms)" % (
      reserve, idle[0][0], idle[0][1], idle[1][0], idle[1][1], busy_null[0], busy_null[1], ms_null, busy_side[0], busy_side[1], ms_side), flush=True)
