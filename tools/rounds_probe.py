import sys, os, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from par_raytracer_amd import api, scenes
s = scenes.make_scene("terrain_1m"); d = tempfile.mkdtemp(); scenes.write_obj(s, d, "scene.obj")
hs = api.HostScene(d, "scene.obj", 0, s.camera_position)
r = api.Renderer(0); r.upload(hs)
w, h = 1920, 1080
cam = api.make_camera(s.fov, w, h, s.camera_position, s.camera_facing)
buf = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
p = api.default_params(8, 1234, pipeline=2)
os.environ["PRT_CHAINS"] = "1"
r.render_device(cam, p, w, h, 0, w * h, buf.data_ptr(), True)
os.environ["PRT_DEBUG_ROUNDS"] = "1"
c = r.render_device(cam, p, w, h, 0, w * h, buf.data_ptr(), True)
print("frame %.2f ms" % c.render_ms)
