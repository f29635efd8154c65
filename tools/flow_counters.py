"""DEBUG_UTIL counters of the round-free kernel on C4 (full frame, 1/8 shard): python tools/flow_counters.py"""
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
r.set_option("DEBUG_UTIL", 1); r.set_option("POOL_FLOW", 1)
for nr in (1, 8):
    p = api.default_params(8, 1234, pipeline=capi.PIPELINE_POOL | capi.FLAG_COUNT_VISITS)
    sys.stderr.write("== no rounds, nranks %d\n" % nr); sys.stderr.flush()
    c = r.render_shard_device(cam, p, w, h, 8, 0, nr, buf.data_ptr(), True)
    sys.stderr.write("   %.2f ms\n" % c.render_ms)
