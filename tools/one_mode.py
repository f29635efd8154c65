"""Render C4 a few times in one sampling mode (for A/B and rocprofv3 --pmc runs):  python tools/one_mode.py <spp> <max_spp> [frames [bounce_depth]]"""
import sys, os, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from par_raytracer_amd import api, scenes, capi
spp, mx = int(sys.argv[1]), int(sys.argv[2]); frames = int(sys.argv[3]) if len(sys.argv) > 3 else 2
s = scenes.make_scene("terrain_1m"); d = tempfile.mkdtemp(); scenes.write_obj(s, d, "scene.obj")
hs = api.HostScene(d, "scene.obj", 0, s.camera_position)
r = api.Renderer(0); r.upload(hs)
w, h = 1920, 1080
cam = api.make_camera(s.fov, w, h, s.camera_position, s.camera_facing)
buf = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda"); torch.cuda.synchronize()
p = api.default_params(spp, 1234, max_spp=mx, pipeline=capi.PIPELINE_POOL)
if len(sys.argv) > 4: p.bounce_depth = int(sys.argv[4])
ms = []
for _ in range(frames):
    c = r.render_device(cam, p, w, h, 0, w * h, buf.data_ptr(), True)
    ms.append(c.render_ms)
ms = sorted(ms[1:]) if len(ms) > 1 else ms
print("spp %d max %d depth %d: min %.2f median %.2f ms over %d frames, %d rays" % (spp, mx, p.bounce_depth, ms[0], ms[len(ms) // 2], len(ms), c.ray_count))
