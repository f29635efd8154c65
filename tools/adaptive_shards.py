import sys, os, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from par_raytracer_amd import api, scenes
s = scenes.make_scene("terrain_1m"); d = tempfile.mkdtemp(); scenes.write_obj(s, d, "scene.obj")
hs = api.HostScene(d, "scene.obj", 0, s.camera_position)
r = api.Renderer(0); r.upload(hs)
w, h = 1920, 1080
cam = api.make_camera(s.fov, w, h, s.camera_position, s.camera_facing)
buf = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda"); torch.cuda.synchronize()
p = api.default_params(10, 1234, max_spp=50)
for nr in (1, 2, 4, 8):
    r.render_shard_device(cam, p, w, h, 8, 0, nr, buf.data_ptr(), True)
    c = r.render_shard_device(cam, p, w, h, 8, 0, nr, buf.data_ptr(), True)
    print("adaptive 10..50, shard 1/%d: %.1f ms, %d rays, %.0f Mrays/s" % (nr, c.render_ms, c.ray_count, c.ray_count / c.render_ms / 1e3), flush=True)
