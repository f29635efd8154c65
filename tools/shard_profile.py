import sys, os, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from par_raytracer_amd import api, scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
s = scenes.make_scene("terrain_1m"); d = tempfile.mkdtemp(); scenes.write_obj(s, d, "scene.obj")
hs = api.HostScene(d, "scene.obj", 0, s.camera_position)
r = api.Renderer(0); r.upload(hs)
w, h = 1920, 1080
cam = api.make_camera(s.fov, w, h, s.camera_position, s.camera_facing)
p = api.default_params(8, 1234)
rows = r.shard_rows(h, 8, 0, n)
buf = torch.zeros((rows, w, 4), dtype=torch.float32, device="cuda")
for _ in range(4):
    r.render_shard_device(cam, p, w, h, 8, 0, n, buf.data_ptr())
torch.cuda.synchronize()
