"""POOL_FAIR sweep on C4 shards (pool capacity = largest top-up = k/8 of a fair share): python tools/fair_probe.py"""
import sys, os, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from par_raytracer_amd import api, scenes, capi
s = scenes.make_scene("terrain_1m"); d = tempfile.mkdtemp(); scenes.write_obj(s, d, "scene.obj")
hs = api.HostScene(d, "scene.obj", 0, s.camera_position)
r = api.Renderer(0); r.upload(hs)
w, h = 1920, 1080
cam = api.make_camera(s.fov, w, h, s.camera_position, s.camera_facing)
buf = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda"); torch.cuda.synchronize()
p = api.default_params(8, 1234, pipeline=capi.PIPELINE_POOL)
def run(nr, n=8):
    r.render_shard_device(cam, p, w, h, 8, 0, nr, buf.data_ptr(), True)
    ms = sorted(r.render_shard_device(cam, p, w, h, 8, 0, nr, buf.data_ptr(), True).render_ms for _ in range(n))
    return ms[0], ms[len(ms) // 2]
for shared in (1, 0):
    r.set_option("POOL_SHARED", shared)
    for fair in (None, 2, 3, 4, 5, 6, 8, 10):
        r.set_option("POOL_FAIR", fair)
        res = [run(nr) for nr in (8, 4, 16)]
        print("shared %d fair %-5s  1/8 %.3f (med %.3f)  1/4 %.3f (%.3f)  1/16 %.3f (%.3f)" % ((shared, fair) + tuple(x for a in res for x in a)), flush=True)
