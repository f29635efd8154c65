"""Wavefront vs pool pipeline by shard size (C4, rank 0 of nranks = 1, 2, 4, 8, 16)."""
import sys, os, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from par_raytracer_amd import api, scenes
s = scenes.make_scene("terrain_1m"); d = tempfile.mkdtemp(); scenes.write_obj(s, d, "scene.obj")
hs = api.HostScene(d, "scene.obj", 0, s.camera_position)
r = api.Renderer(0); r.upload(hs)
w, h = 1920, 1080
cam = api.make_camera(s.fov, w, h, s.camera_position, s.camera_facing)
buf = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
def run(pipeline, nranks, reps=5):
    p = api.default_params(8, 1234, pipeline=pipeline)
    r.render_shard_device(cam, p, w, h, 8, 0, nranks, buf.data_ptr())
    return min(r.render_shard_device(cam, p, w, h, 8, 0, nranks, buf.data_ptr()).render_ms for _ in range(reps))
caps = [int(a) for a in sys.argv[1:]] or [128, 256, 512]
for nr in (1, 2, 4, 8, 16):
    os.environ.pop("PRT_POOL_CAP", None)
    line = "nranks %2d: wavefront %.2f  pool(auto) %.2f" % (nr, run(2, nr), run(4, nr))
    for c in caps:
        os.environ["PRT_POOL_CAP"] = str(c)
        line += "  pool(%d) %.2f" % (c, run(4, nr))
    print(line, flush=True)
