"""C4's terrain with a diffuse map and a bump map on every material: cost of the texture path at scale."""
import sys, os, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
from par_raytracer_amd import api, scenes
import texture_fixtures
s = scenes.make_scene("terrain_1m")
rng = np.random.default_rng(1)
yy, xx = np.mgrid[0:1024, 0:1024]
kd = np.stack([(128 + 100 * np.sin(xx / 37.0) * np.cos(yy / 53.0)), (140 + 80 * np.sin(xx / 11.0)), (120 + 60 * np.cos(yy / 19.0))], axis=2).clip(0, 255).astype(np.uint8)
bump = (128 + 100 * np.sin(xx / 5.0) * np.sin(yy / 7.0)).clip(0, 255).astype(np.uint8)
s.textures = {"kd.png": (kd, "png"), "bump.tga": (bump, "tga")}
s.texture_writer = texture_fixtures.write_texture
for m in s.materials:
    m.map_Kd = "kd.png"; m.map_bump = "bump.tga"
d = tempfile.mkdtemp(); scenes.write_obj(s, d, "scene.obj")
hs = api.HostScene(d, "scene.obj", 0, s.camera_position)
r = api.Renderer(0); info = r.upload(hs)
w, h = 1920, 1080
cam = api.make_camera(s.fov, w, h, s.camera_position, s.camera_facing)
buf = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda"); torch.cuda.synchronize()
for pl in (2, 4):
    p = api.default_params(8, 1234, pipeline=pl)
    r.render_device(cam, p, w, h, 0, w * h, buf.data_ptr(), True)
    cs = [r.render_device(cam, p, w, h, 0, w * h, buf.data_ptr(), True) for _ in range(3)]
    ms = min(c.render_ms for c in cs)
    print("textured terrain (%d materials, %.0f MB resident), pipeline %d: %.2f ms, %d rays, %.0f Mrays/s" % (len(s.materials), info.device_bytes / 1e6, pl, ms, cs[0].ray_count, cs[0].ray_count / ms / 1e3), flush=True)
