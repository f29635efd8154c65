import sys, time, os, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from par_raytracer_amd import api, scenes
s = scenes.make_scene("terrain_1m"); d = tempfile.mkdtemp(); scenes.write_obj(s, d, "scene.obj")
hs = api.HostScene(d, "scene.obj", 0, s.camera_position)
r = api.Renderer(0); r.upload(hs)
w, h = 1920, 1080
cam = api.make_camera(s.fov, w, h, s.camera_position, s.camera_facing)
p = api.default_params(8, 1234)
for n in (1, 2, 4, 8):
    rows = max(r.shard_rows(h, 8, k, n) for k in range(n))
    buf = torch.zeros((rows, w, 4), dtype=torch.float32, device="cuda")
    res = []
    for k in range(n):
        r.render_shard_device(cam, p, w, h, 8, k, n, buf.data_ptr())
        t0 = time.perf_counter()
        for _ in range(5):
            c = r.render_shard_device(cam, p, w, h, 8, k, n, buf.data_ptr())
        torch.cuda.synchronize()
        res.append(((time.perf_counter() - t0) / 5 * 1e3, c.render_ms, c.ray_count))
    print("nranks %d: wall ms per shard %s  device ms %s  rays %s" % (n, ["%.2f" % a for a, _, _ in res], ["%.2f" % b for _, b, _ in res], [c for _, _, c in res]), flush=True)
