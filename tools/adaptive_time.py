"""C4 in the reference's default adaptive mode (10..50 spp): device time and rays per frame."""
import sys, os, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from par_raytracer_amd import api, scenes
name = sys.argv[1] if len(sys.argv) > 1 else "terrain_1m"
s = scenes.make_scene(name); d = tempfile.mkdtemp(); scenes.write_obj(s, d, "scene.obj")
hs = api.HostScene(d, "scene.obj", 0, s.camera_position)
r = api.Renderer(0); r.upload(hs)
w, h = 1920, 1080
cam = api.make_camera(s.fov, w, h, s.camera_position, s.camera_facing)
buf = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
for mn, mx in ((10, 50), (8, 0), (10, 0), (50, 0)):
    p = api.default_params(mn, 1234, max_spp=mx)
    r.render_device(cam, p, w, h, 0, w * h, buf.data_ptr(), True)
    cs = [r.render_device(cam, p, w, h, 0, w * h, buf.data_ptr(), True) for _ in range(3)]
    ms = min(c.render_ms for c in cs)
    print("%s spp %d max_spp %d: %.2f ms, %d rays (%.1f per pixel), %.0f Mrays/s, pipeline %d" % (
        name, mn, mx, ms, cs[0].ray_count, cs[0].ray_count / (w * h), cs[0].ray_count / ms / 1e3, cs[0].pipeline), flush=True)
