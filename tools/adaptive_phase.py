"""Phase split of the pool kernel in adaptive mode (10..50 spp) on C4: PRT_DEBUG_UTIL counters of a counting build."""
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
for mn, mx in ((10, 50), (50, 0)):
    p = api.default_params(mn, 1234, max_spp=mx, pipeline=capi.PIPELINE_POOL | capi.FLAG_COUNT_VISITS)
    sys.stderr.write("spp %d max_spp %d\n" % (mn, mx)); sys.stderr.flush()
    c = r.render_device(cam, p, w, h, 0, w * h, buf.data_ptr(), True)
    sys.stderr.write("  render_ms %.2f rays %d\n" % (c.render_ms, c.ray_count)); sys.stderr.flush()
