"""Lane-utilisation counters (PRT_DEBUG_UTIL) of one C4 frame for the given pipelines."""
import sys, os, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["PRT_DEBUG_UTIL"] = "1"
import numpy as np, torch
from par_raytracer_amd import api, scenes, capi
s = scenes.make_scene("terrain_1m"); d = tempfile.mkdtemp(); scenes.write_obj(s, d, "scene.obj")
hs = api.HostScene(d, "scene.obj", 0, s.camera_position)
r = api.Renderer(0); r.upload(hs)
w, h = 1920, 1080
cam = api.make_camera(s.fov, w, h, s.camera_position, s.camera_facing)
buf = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
for pl in [int(a) for a in sys.argv[1:]] or [2, 4]:
    for nr in (1, 8):
        p = api.default_params(8, 1234, pipeline=pl | capi.FLAG_COUNT_VISITS)
        sys.stderr.write("pipeline %d nranks %d\n" % (pl, nr)); sys.stderr.flush()
        c = r.render_shard_device(cam, p, w, h, 8, 0, nr, buf.data_ptr(), True)
        sys.stderr.write("  render_ms %.2f rays %d\n" % (c.render_ms, c.ray_count)); sys.stderr.flush()
